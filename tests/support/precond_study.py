"""CPU study behind the factorisation-free preconditioner (``fc_setup_krylov``): which block preconditioner of the cylinder O1
BDF2 operator makes GMRES converge without any factorisation?  Test infrastructure (it uses the oracle's matrix and golden
right-hand side); numbers recorded in profiles/EXPERIMENTS.md III.

    python tests/support/precond_study.py            # ~4 min on one core

Variants (GMRES(200), rtol 1e-10, zero initial guess):
  * additive Vanka: one patch per pressure vertex (its P1 dof + the velocity dofs of the incident cells, 37 dofs on average, exact
    dense patch inverses, partition-of-unity weights)        -> ~750 iterations of GMRES(200); GMRES(100) x 10 stops at 4e-9
  * block-triangular / SIMPLE with S = B diag(F)^-1 Bt solved exactly, F by k damped-Jacobi sweeps
  * the same with one smoothed-aggregation AMG V(1,1)-cycle in place of S^-1           -> what the library builds
"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver  # noqa: E402
from oracle import ns_oracle as O  # noqa: E402


def problem():
    fs = CylinderFlowSolver.make_default(path_out=tempfile.mkdtemp())
    th = fs.th
    d = O.Disc.from_taylor_hood(th)
    g = np.load(ROOT / "tests" / "golden" / "cylinder_O1.npz")
    dofs, prof = fs._bc_tables()
    ts = O.TimeStepper(d, fs.params_flow.Re, fs.params_time.dt, g["UP0"][: 2 * th.nn], dofs, prof, orders=(2,))
    return th, d, ts.A_bc[2].tocsr(), g["rhs2"], g["sol2"], np.asarray(dofs)


def aggregate(Am, theta):
    n = Am.shape[0]
    dg = Am.diagonal()
    C = Am.tocoo()
    strong = (np.abs(C.data) >= theta * np.sqrt(np.abs(dg[C.row] * dg[C.col]))) & (C.row != C.col)
    S = sp.csr_matrix((np.ones(strong.sum()), (C.row[strong], C.col[strong])), shape=(n, n))
    agg, na = -np.ones(n, int), 0
    for i in range(n):
        nb = S.indices[S.indptr[i] : S.indptr[i + 1]]
        if agg[i] < 0 and np.all(agg[nb] < 0):
            agg[i] = agg[nb] = na
            na += 1
    first = agg.copy()
    for i in range(n):
        if first[i] < 0:
            cand = first[S.indices[S.indptr[i] : S.indptr[i + 1]]]
            cand = cand[cand >= 0]
            if cand.size:
                agg[i] = cand[0]
    for i in range(n):
        if agg[i] < 0:
            agg[i] = na
            na += 1
    return agg, na


def sa_hierarchy(A0, theta=0.08, nmin=256):
    levels, Am = [], A0.tocsr()
    while Am.shape[0] > nmin:
        agg, na = aggregate(Am, theta)
        n = Am.shape[0]
        T = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, na))
        T = T @ sp.diags(1 / np.sqrt(np.asarray(T.sum(axis=0)).ravel()))
        Dm = Am.diagonal()
        DA = sp.diags(1 / Dm) @ Am
        rho = abs(spla.eigs(DA, k=1, which="LM", return_eigenvectors=False)[0])
        P = (T - (4 / (3 * rho)) * (DA @ T)).tocsr()
        levels.append(dict(A=Am, P=P, R=P.T.tocsr(), D=Dm, rho=rho))
        Am = (P.T @ Am @ P).tocsr()
    levels.append(dict(A=Am, inv=np.linalg.inv(Am.toarray())))
    return levels


def vcycle(levels, r, lvl=0):
    L = levels[lvl]
    if "inv" in L:
        return L["inv"] @ r
    w = 4 / (3 * L["rho"])
    x = w * r / L["D"]
    x += L["P"] @ vcycle(levels, L["R"] @ (r - L["A"] @ x), lvl + 1)
    return x + w * (r - L["A"] @ x) / L["D"]


def main():
    th, d, A, b, xref, bc = problem()
    N, nn, nv = d.N, th.nn, th.nv
    isbc = np.zeros(N, bool)
    isbc[bc] = True
    iu, ip = np.arange(2 * nn), np.arange(2 * nn, N)
    F, Bt, B = A[iu][:, iu].tocsr(), A[iu][:, ip].tocsr(), A[ip][:, iu].tocsr()
    dF = F.diagonal()
    Shat = (B @ sp.diags(1 / dF) @ Bt).tocsr()
    Shlu = spla.splu(Shat.tocsc())
    lv = sa_hierarchy(Shat)

    def run(M, name, maxiter=2):
        its = [0]

        def cb(_):
            its[0] += 1

        t0 = time.time()
        x, _ = spla.gmres(A, b, M=spla.LinearOperator((N, N), matvec=M), rtol=1e-10, restart=200, maxiter=maxiter, callback=cb, callback_type="pr_norm")
        print(f"{name:55s} iterations {its[0]:5d}  error {np.linalg.norm(x - xref) / np.linalg.norm(xref):.1e}  "
              f"residual {np.linalg.norm(b - A @ x) / np.linalg.norm(b):.1e}  ({time.time() - t0:.0f} s)", flush=True)

    def jac(k, om=0.7):
        def f(r):
            x = (om if k > 1 else 1.0) * r / dF
            for _ in range(k - 1):
                x = x + om * (r - F @ x) / dF
            return x
        return f

    def simple(Sinv, F1):
        def M(r):
            u1 = F1(r[iu])
            zp = -Sinv(r[ip] - B @ u1)
            return np.r_[u1 - (Bt @ zp) / dF, zp]
        return M

    def block_triangular(Sinv, F1):
        def M(r):
            zp = -Sinv(r[ip])
            return np.r_[F1(r[iu] - Bt @ zp), zp]
        return M

    for k in (1, 2, 3):
        run(block_triangular(Shlu.solve, jac(k)), f"block-triangular, Jacobi x{k}, exact S^-1")
        run(simple(Shlu.solve, jac(k)), f"SIMPLE, Jacobi x{k}, exact S^-1")
        run(simple(lambda r: vcycle(lv, r), jac(k)), f"SIMPLE, Jacobi x{k}, one AMG V(1,1)-cycle   [the library]")
    # additive Vanka: vertex patches
    inc = [[] for _ in range(nv)]
    for c, cell in enumerate(d.cells):
        for v in cell:
            inc[v].append(c)
    patches = []
    for v in range(nv):
        nodes = np.unique(d.cell_nodes[inc[v]].reshape(-1))
        dofs = np.r_[nodes, nodes + nn, 2 * nn + v]
        patches.append(dofs[~isbc[dofs]])
    invs = [np.linalg.inv(A[p][:, p].toarray()) for p in patches]
    count = np.zeros(N)
    for p in patches:
        count[p] += 1
    w = 1.0 / np.maximum(count, 1)
    print(f"Vanka: {len(patches)} patches, mean {np.mean([len(p) for p in patches]):.1f} dofs, {8 * sum(i.size for i in invs) / 1e6:.0f} MB of dense inverses")

    def vanka(r):
        z = np.zeros(N)
        for p, Ai in zip(patches, invs):
            z[p] += Ai @ r[p]
        z *= w
        z[isbc] = r[isbc]
        return z

    run(lambda r: simple(Shlu.solve, jac(1))(r) + vanka(r - A @ simple(Shlu.solve, jac(1))(r)), "SIMPLE (Jacobi x1, exact S^-1) then additive Vanka")
    run(vanka, "additive Vanka alone (GMRES(200), up to 5 cycles)", maxiter=5)


if __name__ == "__main__":
    main()
