"""CPU ORACLE — test infrastructure only.

A numpy/scipy restatement of the reference's per-timestep hot path (FEniCS/dolfin 2019.1.0
assembly of the forms in ``src/flowcontrol/nsforms.py`` + the factor-once / solve-many
direct solve of ``src/flowcontrol/flowsolver.py:665-799``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module;
the product path (``flowcontrol_amd``) never does — it must fail loudly when the HIP
library is missing.

The arithmetic lives in the un-vendored third-party stack ``fenics=2019.1.0``
(``environment.yml:7``: dolfin/FFC/UFL/FIAT → PETSc → MUMPS, versions of the latter two
un-pinned), which cannot be imported or built in this image; what is restated here is its
published algorithm: exact (degree-5, 7-point) quadrature of the UFL forms on affine P2/P1
triangles, ``SystemAssembler``'s symmetric Dirichlet elimination, and an LU solve whose
factorisation is re-used while the operator is unchanged.

Parity pin: the known-answer constants of the reference's own slow tests
(``tests/integration/test_cylinder.py:66-74``, ``test_operatorgetter.py:23-26`` …), checked in
``tests/test_oracle_reference_constants.py``.

Each function cites the reference lines it follows.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

# ── reference element (7-point, degree-5 Radon rule; P2/P1 Lagrange) ──────────────────────
_S15 = np.sqrt(15.0)
_A1, _A2 = (6.0 - _S15) / 21.0, (6.0 + _S15) / 21.0
_QB = np.array(
    [[1 / 3, 1 / 3, 1 / 3]]
    + [np.roll([1 - 2 * _A1, _A1, _A1], k).tolist() for k in range(3)]
    + [np.roll([1 - 2 * _A2, _A2, _A2], k).tolist() for k in range(3)]
)
_QW = np.array([9 / 40] + [(155 - _S15) / 1200] * 3 + [(155 + _S15) / 1200] * 3)
_DL = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])
_EV = ((1, 2), (2, 0), (0, 1))


def _p2(lam):
    lam = np.asarray(lam, float)
    out = np.empty(lam.shape[:-1] + (6,))
    for i in range(3):
        out[..., i] = lam[..., i] * (2 * lam[..., i] - 1)
    for k, (i, j) in enumerate(_EV):
        out[..., 3 + k] = 4 * lam[..., i] * lam[..., j]
    return out


def _dp2(lam):
    lam = np.asarray(lam, float)
    out = np.empty(lam.shape[:-1] + (6, 2))
    for i in range(3):
        out[..., i, :] = (4 * lam[..., i, None] - 1) * _DL[i]
    for k, (i, j) in enumerate(_EV):
        out[..., 3 + k, :] = 4 * (lam[..., i, None] * _DL[j] + lam[..., j, None] * _DL[i])
    return out


_PHI2, _DPHI2, _PHI1 = _p2(_QB), _dp2(_QB), _QB.copy()


@dataclass
class Disc:
    """Plain-array description of the discretisation (numbering supplied by the caller).

    coords (nv,2); cells (nc,3) CCW vertex ids; cell_nodes (nc,6) P2 scalar node ids
    (vertices then opposite-edge midpoints); nn = number of P2 scalar nodes.
    W layout: [ux(nn), uy(nn), p(nv)].
    """

    coords: np.ndarray
    cells: np.ndarray
    cell_nodes: np.ndarray
    nn: int

    def __post_init__(self):
        self.nv = self.coords.shape[0]
        self.nc = self.cells.shape[0]
        self.N = 2 * self.nn + self.nv
        p = self.coords[self.cells]
        J = np.stack([p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]], axis=2)  # (nc, 2, 2) columns
        self.detJ = J[:, 0, 0] * J[:, 1, 1] - J[:, 0, 1] * J[:, 1, 0]
        assert np.all(self.detJ > 0), "oracle expects CCW cells"
        self.Jinv = np.linalg.inv(J)
        # physical gradients of basis functions at quadrature points
        self.G2 = np.einsum("qar,crd->cqad", _DPHI2, self.Jinv)  # (nc, 7, 6, 2)
        self.G1 = np.einsum("mr,crd->cmd", _DL, self.Jinv)  # (nc, 3, 2)
        self.w = _QW[None, :] * (0.5 * self.detJ)[:, None]  # (nc, 7)
        cn = self.cell_nodes.astype(np.int64)
        self.cell_dofs = np.hstack([cn, cn + self.nn, self.cells.astype(np.int64) + 2 * self.nn])

    @classmethod
    def from_taylor_hood(cls, th) -> "Disc":
        return cls(th.mesh.coords.copy(), th.mesh.cells.copy(), th.cell_nodes.copy(), th.nn)

    @classmethod
    def from_mesh_arrays(cls, coords, cells) -> "Disc":
        """The oracle's OWN numbering, built from the raw mesh arrays alone (nothing of the product's discretisation):
        cells re-oriented counter-clockwise, P2 edge nodes numbered by the sorted (lower vertex, upper vertex) key of their
        edge — dolfin's P2 space puts one dof at every vertex and every edge midpoint, the mixed space
        ``VectorElement(P2) x P1`` (``flowsolver.py:300-306``) then has 2(nv + ne) + nv dofs whatever their order.
        ``node_coords()`` gives the points of the scalar P2 nodes, by which a test matches this numbering with another."""
        coords = np.asarray(coords, dtype=np.float64)
        cells = np.asarray(cells, dtype=np.int64).copy()
        p = coords[cells]
        det = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 1, 1] - p[:, 0, 1]) * (p[:, 2, 0] - p[:, 0, 0])
        cells[det < 0] = cells[det < 0][:, [0, 2, 1]]
        nv, nc = coords.shape[0], cells.shape[0]
        lo = np.stack([np.minimum(cells[:, i], cells[:, j]) for i, j in _EV], axis=1)  # edge k is opposite to vertex k
        hi = np.stack([np.maximum(cells[:, i], cells[:, j]) for i, j in _EV], axis=1)
        uniq, inv = np.unique((lo * nv + hi).reshape(-1), return_inverse=True)
        d = cls(coords, cells, np.hstack([cells, nv + inv.reshape(nc, 3)]), nv + uniq.size)
        d._edge_lo, d._edge_hi = uniq // nv, uniq % nv
        return d

    def node_coords(self) -> np.ndarray:
        """Points of the scalar P2 nodes (vertices, then edge midpoints) of :meth:`from_mesh_arrays`."""
        return np.vstack([self.coords, 0.5 * (self.coords[self._edge_lo] + self.coords[self._edge_hi])])

    # field helpers
    def vel_at_q(self, u: np.ndarray):
        """u (2nn,) → values (nc,7,2) and gradients ∂_i u_j (nc,7,2,2) at quadrature points."""
        ue = np.stack([u[self.cell_nodes], u[self.nn + self.cell_nodes]], axis=2)  # (nc,6,2)
        val = np.einsum("qa,caj->cqj", _PHI2, ue)
        grad = np.einsum("cqai,caj->cqij", self.G2, ue)
        return val, grad


# ── bilinear form ─────────────────────────────────────────────────────────────────────────
def assemble_matrix(
    d: Disc,
    mass: float = 0.0,
    nu: float = 0.0,
    adv: np.ndarray | None = None,
    lin: np.ndarray | None = None,
    adv_scale: float = 1.0,
    lin_scale: float = 1.0,
    pressure: float = -1.0,
    divergence: float = -1.0,
) -> sp.csr_matrix:
    """Global matrix of
        mass·(u,v) + adv_scale·((adv·∇)u, v) + lin_scale·((u·∇)lin, v) + nu·(∇u,∇v)
        + pressure·(p, div v) + divergence·(q, div u)
    i.e. the ``lhs`` of ``NSForms._order1/_order2`` (``nsforms.py:238-305``; mass = 1/dt or
    3/(2dt) minus shift), of ``NSForms.picard`` (``nsforms.py:179-185``) and the Jacobian of
    ``NSForms.steady`` (``nsforms.py:141-147``).  ``nabla_grad(u)[i,j] = ∂_i u_j``.
    No boundary conditions are applied here.
    """
    nc = d.nc
    Ae = np.zeros((nc, 15, 15))
    w = d.w
    Mab = np.einsum("cq,qa,qb->cab", w, _PHI2, _PHI2)
    Kab = np.einsum("cq,cqad,cqbd->cab", w, d.G2, d.G2)
    blk = mass * Mab + nu * Kab
    if adv is not None:
        Uq, _ = d.vel_at_q(adv)
        UdG = np.einsum("cqi,cqbi->cqb", Uq, d.G2)  # (U·∇)φ_b
        blk = blk + adv_scale * np.einsum("cq,qa,cqb->cab", w, _PHI2, UdG)
    Ae[:, 0:6, 0:6] = blk
    Ae[:, 6:12, 6:12] = blk
    if lin is not None:
        _, GU = d.vel_at_q(lin)  # GU[c,q,k,j] = ∂_k U_j
        for j in range(2):
            for k in range(2):
                Ae[:, 6 * j : 6 * j + 6, 6 * k : 6 * k + 6] += lin_scale * np.einsum(
                    "cq,qa,qb,cq->cab", w, _PHI2, _PHI2, GU[:, :, k, j]
                )
    for j in range(2):
        Bj = np.einsum("cq,qm,cqa->cam", w, _PHI1, d.G2[:, :, :, j])  # ∫ ψ_m ∂_j φ_a
        Ae[:, 6 * j : 6 * j + 6, 12:15] = pressure * Bj
        Ae[:, 12:15, 6 * j : 6 * j + 6] = divergence * np.transpose(Bj, (0, 2, 1))
    rows = np.repeat(d.cell_dofs, 15, axis=1).reshape(-1)
    cols = np.tile(d.cell_dofs, (1, 15)).reshape(-1)
    A = sp.coo_matrix((Ae.reshape(-1), (rows, cols)), shape=(d.N, d.N)).tocsr()
    A.sum_duplicates()
    return A


def velocity_mass(d: Disc) -> sp.csr_matrix:
    """(u, v) on V (2nn × 2nn): the matrix behind ``dolfin.norm(u, "L2")`` (``flowsolver.py:829``)."""
    Mab = np.einsum("cq,qa,qb->cab", d.w, _PHI2, _PHI2)
    cn = d.cell_nodes.astype(np.int64)
    rows = np.repeat(cn, 6, axis=1).reshape(-1)
    cols = np.tile(cn, (1, 6)).reshape(-1)
    M = sp.coo_matrix((Mab.reshape(-1), (rows, cols)), shape=(d.nn, d.nn)).tocsr()
    return sp.block_diag([M, M]).tocsr()


# ── linear forms ──────────────────────────────────────────────────────────────────────────
def _load(d: Disc, gq: np.ndarray) -> np.ndarray:
    """∫ g·v for g given at quadrature points (nc,7,2) → W-vector (pressure rows zero)."""
    Le = np.einsum("cq,qa,cqj->caj", d.w, _PHI2, gq)
    b = np.zeros(d.N)
    np.add.at(b, d.cell_nodes.reshape(-1), Le[:, :, 0].reshape(-1))
    np.add.at(b, d.nn + d.cell_nodes.reshape(-1), Le[:, :, 1].reshape(-1))
    return b


def convective(d: Disc, u: np.ndarray) -> np.ndarray:
    """(u·∇)u at quadrature points, (nc,7,2)."""
    val, grad = d.vel_at_q(u)
    return np.einsum("cqi,cqij->cqj", val, grad)


def rhs_transient(
    d: Disc,
    order: int,
    dt: float,
    u_n: np.ndarray,
    u_nn: np.ndarray | None,
    f_nodal: np.ndarray | None = None,
    nonlinear: bool = True,
) -> np.ndarray:
    """``dolfin.rhs`` of ``NSForms._order1`` / ``_order2`` (``nsforms.py:238-305``) without BCs.

    BDF1: ∫ u_n/dt·v − b0 (u_n·∇)u_n·v + f·v,           b0 = 1
    BDF2: ∫ (4u_n − u_nn)/(2dt)·v − 2(u_n·∇)u_n·v + (u_nn·∇)u_nn·v + f·v
    ``f_nodal`` (2nn,) is the P2 interpolant of the body force (Appendix A: an
    ``Expression(element=V.ufl_element())`` is interpolated cell-wise before integration).
    """
    vn, _ = d.vel_at_q(u_n)
    nl = 1.0 if nonlinear else 0.0
    if order == 1:
        g = vn / dt - nl * convective(d, u_n)
    elif order == 2:
        vnn, _ = d.vel_at_q(u_nn)
        g = (4.0 * vn - vnn) / (2.0 * dt) - nl * 2.0 * convective(d, u_n) + nl * convective(d, u_nn)
    else:
        raise ValueError("order must be 1 or 2")
    if f_nodal is not None:
        fq, _ = d.vel_at_q(f_nodal)
        g = g + fq
    return _load(d, g)


def steady_residual(d: Disc, nu: float, up: np.ndarray, f_nodal: np.ndarray | None = None) -> np.ndarray:
    """Residual vector of ``NSForms.steady`` (``nsforms.py:141-147``), no BCs."""
    nn = d.nn
    u, p = up[: 2 * nn], up[2 * nn :]
    val, grad = d.vel_at_q(u)
    conv = np.einsum("cqi,cqij->cqj", val, grad)
    if f_nodal is not None:
        fq, _ = d.vel_at_q(f_nodal)
        conv = conv - fq
    F = _load(d, conv)
    # ν ∇U:∇v − P div v
    pq = np.einsum("qm,cm->cq", _PHI1, p[d.cells])
    Le = nu * np.einsum("cq,cqai,cqij->caj", d.w, d.G2, grad) - np.einsum("cq,cq,cqaj->caj", d.w, pq, d.G2)
    np.add.at(F, d.cell_nodes.reshape(-1), Le[:, :, 0].reshape(-1))
    np.add.at(F, nn + d.cell_nodes.reshape(-1), Le[:, :, 1].reshape(-1))
    # − q div U
    divu = grad[:, :, 0, 0] + grad[:, :, 1, 1]
    Lp = -np.einsum("cq,qm,cq->cm", d.w, _PHI1, divu)
    np.add.at(F, 2 * nn + d.cells.reshape(-1), Lp.reshape(-1))
    return F


# ── Dirichlet conditions ──────────────────────────────────────────────────────────────────
def apply_bc_symmetric(A: sp.csr_matrix, b: np.ndarray | None, dofs: np.ndarray, vals: np.ndarray):
    """``SystemAssembler`` semantics: lift, zero BC rows *and* columns, unit diagonal, b[D]=g."""
    N = A.shape[0]
    keep = np.ones(N)
    keep[dofs] = 0.0
    if b is not None:
        g = np.zeros(N)
        g[dofs] = vals
        b = b - A @ g
        b[dofs] = vals
    Dk = sp.diags(keep)
    Abc = (Dk @ A @ Dk + sp.diags(1.0 - keep)).tocsr()
    return Abc, b


def apply_bc_rows(A: sp.csr_matrix, b: np.ndarray | None, dofs: np.ndarray, vals: np.ndarray):
    """``bc.apply(A, b)`` semantics (``steadystate.py:143``, ``operatorgetter.py:81``): zero the
    rows, 1 on the diagonal, columns untouched, b[row] = g."""
    N = A.shape[0]
    keep = np.ones(N)
    keep[dofs] = 0.0
    Abc = (sp.diags(keep) @ A + sp.diags(1.0 - keep)).tocsr()
    if b is not None:
        b = b.copy()
        b[dofs] = vals
    return Abc, b


class _PermutedLU:
    """SuperLU on P A Pᵀ with the caller's fill-reducing ordering (NATURAL column order)."""

    def __init__(self, A, perm):
        self.perm = np.asarray(perm)
        self.lu = spla.splu(A[self.perm][:, self.perm].tocsc(), permc_spec="NATURAL", diag_pivot_thresh=0.01)

    def solve(self, b):
        x = np.empty_like(b)
        x[self.perm] = self.lu.solve(b[self.perm])
        return x


def _lu(A, perm=None):
    """Sparse LU (SuperLU).  ``perm``: optional fill-reducing symmetric ordering (e.g. nested
    dissection); without it SuperLU's own MMD(AᵀA+A) ordering is used."""
    if perm is not None:
        return _PermutedLU(A, perm)
    return spla.splu(A.tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.1, options=dict(SymmetricMode=True))


# ── steady state (setup) ──────────────────────────────────────────────────────────────────
def picard(d: Disc, nu: float, up0: np.ndarray, bc_dofs, bc_vals, max_iter=10, tol=1e-8, f_nodal=None, log=None, perm=None):
    """``SteadyStateSolver.picard`` (``steadystate.py:98-159``)."""
    up0 = up0.copy()
    bp = np.zeros(d.N) if f_nodal is None else _load(d, d.vel_at_q(f_nodal)[0])
    up1 = up0
    for i in range(max_iter):
        A = assemble_matrix(d, nu=nu, adv=up0[: 2 * d.nn])
        Ab, b = apply_bc_rows(A, bp, bc_dofs, bc_vals)
        up1 = _lu(Ab, perm).solve(b)
        rel = np.linalg.norm(up1 - up0) / (np.linalg.norm(up0) + 1e-14)
        up0 = up1.copy()
        if log:
            log(f"Picard {i + 1}/{max_iter} rel_err={rel:.3e}")
        if rel < tol:
            break
    return up1


def newton(d: Disc, nu: float, up0: np.ndarray, bc_dofs, bc_vals, max_iter=25, f_nodal=None, rtol=1e-9, atol=1e-10, log=None, perm=None):
    """``SteadyStateSolver.newton`` → ``dolfin.solve(F == 0, UP0, bcs)`` (``steadystate.py:60-96``)
    with dolfin's NewtonSolver defaults (residual criterion, rel 1e-9 / abs 1e-10)."""
    up = up0.copy()
    up[bc_dofs] = bc_vals
    r0 = None
    for it in range(max_iter + 1):
        F = steady_residual(d, nu, up, f_nodal)
        F[bc_dofs] = 0.0
        r = np.linalg.norm(F)
        r0 = r if r0 is None else r0
        if log:
            log(f"Newton {it}: |F|={r:.3e}")
        if r < atol or r < rtol * r0:
            return up
        if it == max_iter:
            break
        u = up[: 2 * d.nn]
        Jm = assemble_matrix(d, nu=nu, adv=u, lin=u)
        Jb, _ = apply_bc_rows(Jm, None, bc_dofs, bc_vals)
        up = up - _lu(Jb, perm).solve(F)
    raise RuntimeError("Newton solver did not converge")


def steady_jacobian_A(d: Disc, nu: float, up0: np.ndarray, bc_dofs) -> sp.csr_matrix:
    """``OperatorGetter.get_A`` (``operatorgetter.py:25-83``): A = −dF/dUP0 with ``bc.apply(Jac)``
    (BC rows → identity rows)."""
    u = up0[: 2 * d.nn]
    Jm = assemble_matrix(d, nu=nu, adv=u, lin=u)
    A, _ = apply_bc_rows(-Jm, None, bc_dofs, np.zeros(len(bc_dofs)))
    return A


# ── time stepping ─────────────────────────────────────────────────────────────────────────
class TimeStepper:
    """``FlowSolver._prepare_systems`` + the solve part of ``FlowSolver.step``
    (``flowsolver.py:665-751``): constant LHS per order, factor once, two triangular solves
    per step; RHS re-assembled every step with BC values re-evaluated and lifted.

    ``bc_profiles`` (n_bc_dofs, n_act): Dirichlet value = bc_profiles @ u_ctrl (every actuator
    expression is linear in ``u_ctrl``).  ``force_profiles`` (2nn, n_act): nodal
    body force per unit u_ctrl (FORCE-type actuators).
    """

    def __init__(self, d: Disc, Re: float, dt: float, U0: np.ndarray, bc_dofs, bc_profiles,
                 force_profiles=None, nonlinear=True, shift=0.0, orders=(1, 2), perm=None):
        self.d, self.dt, self.nonlinear = d, dt, nonlinear
        self.perm = perm
        self.bc_dofs = np.asarray(bc_dofs, dtype=np.int64)
        self.bc_profiles = np.asarray(bc_profiles, dtype=np.float64).reshape(len(self.bc_dofs), -1)
        self.force_profiles = force_profiles
        self.A_full, self.A_bc, self.lu = {}, {}, {}
        for order in orders:
            alpha = (1.0 if order == 1 else 1.5) / dt - shift
            A = assemble_matrix(d, mass=alpha, nu=1.0 / Re, adv=U0, lin=U0)
            self.A_full[order] = A
            self.A_bc[order], _ = apply_bc_symmetric(A, None, self.bc_dofs, np.zeros(len(self.bc_dofs)))
            self.lu[order] = None

    def rhs(self, order, u_n, u_nn, u_ctrl) -> np.ndarray:
        u_ctrl = np.atleast_1d(np.asarray(u_ctrl, dtype=np.float64))
        f = None
        if self.force_profiles is not None:
            f = self.force_profiles @ u_ctrl
        b = rhs_transient(self.d, order, self.dt, u_n, u_nn, f, self.nonlinear)
        g = np.zeros(self.d.N)
        g[self.bc_dofs] = self.bc_profiles @ u_ctrl
        b = b - self.A_full[order] @ g
        b[self.bc_dofs] = g[self.bc_dofs]
        return b

    def solve(self, order, b) -> np.ndarray:
        if self.lu[order] is None:
            self.lu[order] = _lu(self.A_bc[order], self.perm)
        return self.lu[order].solve(b)

    def step(self, order, u_n, u_nn, u_ctrl) -> np.ndarray:
        return self.solve(order, self.rhs(order, u_n, u_nn, u_ctrl))


class TimeStepperCN:
    """Crank–Nicolson variant (``NSForms._cn``, ``nsforms.py:191-236``; ``flowsolver.py:681-686,755-758``):
    θ=½ on the linear terms, explicit perturbation advection, fully implicit pressure, body force
    ½(f^{n+1} + f^n) with f^n the force of the previous step (zero before the first step)."""

    def __init__(self, d: Disc, Re: float, dt: float, U0: np.ndarray, bc_dofs, bc_profiles, force_profiles=None,
                 nonlinear=True, shift=0.0, perm=None):
        self.d, self.dt, self.nonlinear, self.perm = d, dt, nonlinear, perm
        self.bc_dofs = np.asarray(bc_dofs, dtype=np.int64)
        self.bc_profiles = np.asarray(bc_profiles, dtype=np.float64).reshape(len(self.bc_dofs), -1)
        self.force_profiles = force_profiles
        nu = 1.0 / Re
        self.A_full = assemble_matrix(d, mass=1.0 / dt - shift, nu=0.5 * nu, adv=U0, lin=U0, adv_scale=0.5, lin_scale=0.5)
        self.C = assemble_matrix(d, mass=0.0, nu=0.5 * nu, adv=U0, lin=U0, adv_scale=0.5, lin_scale=0.5, pressure=0.0, divergence=0.0)
        self.A_bc, _ = apply_bc_symmetric(self.A_full, None, self.bc_dofs, np.zeros(len(self.bc_dofs)))
        self.lu = None
        self.u_ctrl_prev = None

    def rhs(self, u_n, u_ctrl, u_ctrl_prev=None) -> np.ndarray:
        u_ctrl = np.atleast_1d(np.asarray(u_ctrl, dtype=np.float64))
        f = None
        if self.force_profiles is not None:
            prev = np.zeros_like(u_ctrl) if u_ctrl_prev is None else np.atleast_1d(u_ctrl_prev)
            f = self.force_profiles @ (0.5 * (u_ctrl + prev))
        b = rhs_transient(self.d, 1, self.dt, u_n, None, f, self.nonlinear)
        b = b - self.C @ np.r_[u_n, np.zeros(self.d.nv)]
        g = np.zeros(self.d.N)
        g[self.bc_dofs] = self.bc_profiles @ u_ctrl
        b = b - self.A_full @ g
        b[self.bc_dofs] = g[self.bc_dofs]
        return b

    def step(self, u_n, u_ctrl) -> np.ndarray:
        if self.lu is None:
            self.lu = _lu(self.A_bc, self.perm)
        b = self.rhs(u_n, u_ctrl, self.u_ctrl_prev)
        self.u_ctrl_prev = np.atleast_1d(np.asarray(u_ctrl, dtype=np.float64)).copy()
        return self.lu.solve(b)


def div0_gaussian_nodal(x: np.ndarray, xloc: float, yloc: float, size: float) -> np.ndarray:
    """``get_div0_u`` (``utils/physics.py:32-56``): ψ = 0.25·exp(−r²/(2 s²)),
    u = (∂ψ/∂y, −∂ψ/∂x) evaluated at the P2 nodes ``x`` (n,2) → (n,2)."""
    dx, dy = x[:, 0] - xloc, x[:, 1] - yloc
    psi = 0.25 * np.exp(-0.5 * (dx * dx + dy * dy) / size**2)
    return np.stack([-dy / size**2 * psi, dx / size**2 * psi], axis=1)


def zoh_discretize(A, B, C, D, dt):
    """``control.c2d(sys, dt, method="zoh")`` (``controller.py:121-134``): exact ZOH through the
    augmented-matrix exponential."""
    from scipy.linalg import expm

    A, B = np.atleast_2d(A), np.atleast_2d(B)
    n, m = A.shape[0], B.shape[1]
    Mx = np.zeros((n + m, n + m))
    Mx[:n, :n], Mx[:n, n:] = A * dt, B * dt
    E = expm(Mx)
    return E[:n, :n], E[:n, n:], np.atleast_2d(C), np.atleast_2d(D)
