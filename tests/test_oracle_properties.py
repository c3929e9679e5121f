"""Structural identities of the oracle (the reference's property-style tests restated):
finite-difference check of the steady Jacobian (tests/integration/test_operatorgetter.py:106-130),
Newton ≡ Picard on a Re=1 problem (tests/test_steadystate.py:80-97), linearity of the RHS in u_ctrl,
second-order convergence of BDF2 and Crank–Nicolson (no reference constants exist for CN: the
CN stepper is pinned only by these properties and by agreeing with BDF2 as dt → 0)."""
import numpy as np
import pytest

from flowcontrol_amd.fem.mesh import Mesh
from flowcontrol_amd.fem.spaces import TaylorHood
from oracle import ns_oracle as O


@pytest.fixture(scope="module")
def small():
    th = TaylorHood(Mesh.unit_square(6, 6))
    d = O.Disc.from_taylor_hood(th)
    m = th.mesh
    be = m.boundary_edges()
    be = be[m.edge_midpoints()[be, 0] < 1 - 1e-9]  # x = 1 stays open (outflow)
    nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
    dofs = np.sort(np.r_[nodes, nodes + th.nn])
    return th, d, dofs


def test_jacobian_matches_finite_differences(small):
    th, d, dofs = small
    rng = np.random.default_rng(1)
    up = 0.3 * rng.standard_normal(th.N)
    x = rng.standard_normal(th.N)
    x[dofs] = 0.0
    h = 1e-6
    nu = 0.1
    fd = (O.steady_residual(d, nu, up + h * x) - O.steady_residual(d, nu, up)) / h
    u = up[: 2 * th.nn]
    Jx = O.assemble_matrix(d, nu=nu, adv=u, lin=u) @ x
    interior = np.setdiff1d(np.arange(th.N), dofs)
    assert np.linalg.norm(Jx[interior] - fd[interior]) / np.linalg.norm(Jx[interior]) < 1e-4


def test_newton_equals_picard_lid_cavity_re1():
    th = TaylorHood(Mesh.unit_square(6, 6))
    d = O.Disc.from_taylor_hood(th)
    m = th.mesh
    be = m.boundary_edges()
    nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
    dofs = np.r_[nodes, nodes + th.nn]
    vals = np.r_[(np.abs(th.node_coords[nodes, 1] - 1.0) < 1e-12).astype(float), np.zeros(len(nodes))]
    # pin the pressure level through one extra "Dirichlet" row (all-Dirichlet velocity ⇒ singular p)
    dofs = np.r_[dofs, 2 * th.nn]
    vals = np.r_[vals, 0.0]
    order = np.argsort(dofs)
    dofs, vals = dofs[order], vals[order]
    up0 = np.zeros(th.N)
    up_p = O.picard(d, 1.0, up0, dofs, vals, max_iter=40, tol=1e-12)
    up_n = O.newton(d, 1.0, up0, dofs, vals, max_iter=25)
    u = slice(0, 2 * th.nn)
    assert np.linalg.norm(up_p[u] - up_n[u]) / np.linalg.norm(up_n[u]) < 1e-3


def test_rhs_is_affine_in_u_ctrl_and_bdf_orders(small):
    th, d, dofs = small
    x = th.node_coords
    U0 = np.r_[1 + 0.2 * np.sin(x[:, 0]), 0.1 * np.cos(x[:, 1])]
    prof = np.stack([np.sin(np.arange(len(dofs))), np.cos(0.3 * np.arange(len(dofs)))], axis=1)
    ts = O.TimeStepper(d, 50.0, 0.01, U0, dofs, prof)
    rng = np.random.default_rng(0)
    u_n, u_nn = 0.1 * rng.standard_normal(2 * th.nn), 0.1 * rng.standard_normal(2 * th.nn)
    b0 = ts.rhs(2, u_n, u_nn, [0.0, 0.0])
    b1 = ts.rhs(2, u_n, u_nn, [1.0, 0.0])
    b2 = ts.rhs(2, u_n, u_nn, [0.0, 1.0])
    b = ts.rhs(2, u_n, u_nn, [0.7, -0.4])
    assert np.allclose(b, b0 + 0.7 * (b1 - b0) - 0.4 * (b2 - b0), rtol=1e-12, atol=1e-13)
    # symmetric elimination keeps the Dirichlet values exactly
    up = ts.step(2, u_n, u_nn, [0.7, -0.4])
    assert np.allclose(up[dofs], prof @ [0.7, -0.4], atol=1e-13)


def _advance(d, stepper_factory, dt, T, u0):
    n = int(round(T / dt))
    ts = stepper_factory(dt)
    u_n, u_nn = u0.copy(), u0.copy()
    for k in range(n):
        if isinstance(ts, O.TimeStepperCN):
            up = ts.step(u_n, [0.0])
        else:
            up = ts.step(1 if k == 0 else 2, u_n, u_nn, [0.0])
        u_nn, u_n = u_n, up[: u0.size]
    return u_n


def test_bdf2_and_crank_nicolson_orders_and_agreement(small):
    """Linearised equations (is_eq_nonlinear=False): CN and BDF2 are both O(dt²) and converge to the
    same trajectory.  (With the nonlinear term the reference's CN form treats (u_n·∇)u_n by explicit
    Euler, nsforms.py:217-218, so the full scheme is formally first order — not asserted here.)"""
    th, d, dofs = small
    x = th.node_coords
    U0 = np.r_[1 + 0.2 * np.sin(x[:, 0]), 0.1 * np.cos(x[:, 1])]
    prof = np.zeros((len(dofs), 1))
    u0 = O.div0_gaussian_nodal(th.node_coords, 0.5, 0.5, 0.12)
    u0 = np.r_[u0[:, 0], u0[:, 1]]
    u0[dofs] = 0.0
    T, Re = 0.08, 50.0
    cn = lambda dt: O.TimeStepperCN(d, Re, dt, U0, dofs, prof, nonlinear=False)  # noqa: E731
    bdf = lambda dt: O.TimeStepper(d, Re, dt, U0, dofs, prof, nonlinear=False)  # noqa: E731
    ref = _advance(d, bdf, T / 128, T, u0)
    e_cn = [np.linalg.norm(_advance(d, cn, T / n, T, u0) - ref) for n in (8, 16)]
    e_bdf = [np.linalg.norm(_advance(d, bdf, T / n, T, u0) - ref) for n in (8, 16)]
    assert 3.0 < e_cn[0] / e_cn[1] < 5.5, e_cn  # O(dt²)
    assert 2.5 < e_bdf[0] / e_bdf[1] < 5.5, e_bdf  # BDF1 start-up step, then O(dt²)
    assert e_cn[1] / np.linalg.norm(ref) < 2e-2
