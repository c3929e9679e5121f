"""The C ABI on its own: a solver is created, set up and stepped through ``libfc_hip.so`` with ctypes and numpy only —
nothing of the Python package but the loader (``flowcontrol_amd._lib``: path, build check, argtypes).  This is the call
sequence INTEGRATION.md §B gives a maintainer of the reference; every piece of solver setup (elimination tree,
factor layout, sweep tables, numeric factorisation) happens behind ``fc_setup_solver``.

Checked against the CPU oracle on the same inputs (sparse LU of the same matrices).
"""
import ctypes as C

import numpy as np
import pytest

from flowcontrol_amd import _lib
from oracle import ns_oracle as O

pytestmark = pytest.mark.gpu


def _square_mesh(n):
    """n x n squares cut into CCW triangles; edges numbered in order of first appearance."""
    xs = np.linspace(0.0, 1.0, n + 1)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel()], axis=1)
    vid = lambda i, j: i * (n + 1) + j  # noqa: E731
    cells = []
    for i in range(n):
        for j in range(n):
            a, b, c, d = vid(i, j), vid(i + 1, j), vid(i + 1, j + 1), vid(i, j + 1)
            cells += [(a, b, c), (a, c, d)]
    cells = np.array(cells, dtype=np.int32)
    edge_id, edges = {}, []
    cell_edges = np.empty_like(cells)
    for c, tri in enumerate(cells):
        for k in range(3):  # local edge k is opposite local vertex k
            key = tuple(sorted((int(tri[(k + 1) % 3]), int(tri[(k + 2) % 3]))))
            if key not in edge_id:
                edge_id[key] = len(edges)
                edges.append(key)
            cell_edges[c, k] = edge_id[key]
    return coords, cells, cell_edges, np.array(edges, dtype=np.int32)


def _check(lib, rc):
    assert rc == 0, lib.fc_last_error().decode()


@pytest.mark.parametrize("mode", ["factors", "krylov"])
def test_setup_and_step_through_the_c_abi_only(mode):
    """mode "krylov": ``fc_setup_krylov`` in place of ``fc_setup_solver`` — nothing is factorised, the same ``fc_step`` calls run
    the device GMRES with the SIMPLE / AMG preconditioner (on this 12 x 12 mesh the pressure Schur complement has 169 rows: the
    hierarchy is its dense inverse alone)."""
    lib = _lib.load()
    n = 12
    coords, cells, cell_edges, edges = _square_mesh(n)
    nv, ne, nc = len(coords), len(edges), len(cells)
    nn, N = nv + ne, 2 * (nv + ne) + nv
    node_xy = np.vstack([coords, 0.5 * (coords[edges[:, 0]] + coords[edges[:, 1]])])
    # Dirichlet data: no-slip on bottom / left, a tangential "lid" on top driven by actuator 0, a normal jet on the left
    # wall's middle third driven by actuator 1; the right side is an outflow
    x, y = node_xy[:, 0], node_xy[:, 1]
    wall = (y < 1e-12) | (y > 1 - 1e-12) | (x < 1e-12)
    nodes = np.flatnonzero(wall)
    bc_dofs = np.r_[nodes, nodes + nn].astype(np.int32)
    prof = np.zeros((bc_dofs.size, 2))
    lid = y[nodes] > 1 - 1e-12
    prof[: nodes.size][lid, 0] = (x[nodes][lid] * (1 - x[nodes][lid])) * 4
    jet = (x[nodes] < 1e-12) & (np.abs(y[nodes] - 0.5) < 1.0 / 6)
    prof[: nodes.size][jet, 1] = 1 - (6 * (y[nodes][jet] - 0.5)) ** 2
    U0 = np.r_[0.4 * y * (2 - y), 0.05 * np.sin(np.pi * x) * y * (1 - y)]
    Re, dt = 80.0, 0.01

    h = C.c_void_p()
    _check(lib, lib.fc_create(C.byref(h), 0, nv, ne, nc, np.ascontiguousarray(coords), cells, cell_edges))
    try:
        sizes = [C.c_int64() for _ in range(3)]
        _check(lib, lib.fc_get_sizes(h, *[C.byref(s) for s in sizes]))
        assert (sizes[0].value, sizes[2].value) == (N, nn)
        _check(lib, lib.fc_set_bc(h, bc_dofs.size, bc_dofs.ctypes.data_as(C.c_void_p), 2, prof.ctypes.data_as(C.c_void_p)))
        _check(lib, lib.fc_set_time_scheme(h, dt, 1))
        # sensors: two point-like functionals (a few dofs each)
        rp = np.array([0, 3, 5], dtype=np.int32)
        sidx = np.array([5, nn + 7, 2 * nn + 3, 40, nn + 41], dtype=np.int32)
        sw = np.array([0.5, 0.25, 0.25, 1.0, -1.0])
        _check(lib, lib.fc_set_sensors(h, 2, rp.ctypes.data_as(C.c_void_p), sidx.ctypes.data_as(C.c_void_p), sw.ctypes.data_as(C.c_void_p)))
        for slot, alpha in ((_lib.SLOT_BDF1, 1.0 / dt), (_lib.SLOT_BDF2, 1.5 / dt)):
            _check(lib, lib.fc_assemble_matrix(h, slot, alpha, 1.0 / Re, U0.ctypes.data_as(C.c_void_p), 1.0, U0.ctypes.data_as(C.c_void_p), 1.0, -1.0, -1.0))
            _check(lib, lib.fc_apply_bc(h, slot))
            if mode == "factors":
                _check(lib, lib.fc_setup_solver(h, slot, 0, 2, 0, 0, 1))
            else:
                _check(lib, lib.fc_setup_krylov(h, slot, 2, _lib.METHOD_GMRES, 300, 1e-12, 1))
        info = np.zeros(10, dtype=np.int64)
        _check(lib, lib.fc_get_solver_info(h, _lib.SLOT_BDF2, info))
        if mode == "factors":
            assert info[0] > 0 and info[3] == 2 * info[4] + 1  # factor values; stages = up-sweeps + down-sweeps of the tree
        else:
            kinfo = np.zeros(8, dtype=np.int64)
            _check(lib, lib.fc_get_krylov_info(h, _lib.SLOT_BDF2, kinfo, None))
            assert info[0] == 0 and info[3] == 0 and kinfo[0] > 0 and kinfo[2] == nv  # no factor values, no sweep stages; pressure dofs
        perm = np.empty(N, dtype=np.int32)
        _check(lib, lib.fc_get_permutation(h, perm))
        assert np.array_equal(np.sort(perm), np.arange(N))

        rng = np.random.default_rng(3)
        u0 = 0.1 * rng.standard_normal(2 * nn)
        u0[bc_dofs] = 0.0
        _check(lib, lib.fc_set_state(h, u0, u0, None))

        d = O.Disc(coords, cells, np.hstack([cells, cell_edges + nv]), nn)
        ts = O.TimeStepper(d, Re, dt, U0, bc_dofs, prof)
        M = O.velocity_mass(d)
        u_n, u_nn = u0.copy(), u0.copy()
        y_out, dE, inf = np.zeros(2), C.c_double(), np.zeros(4)
        for k in range(6):
            u = np.array([0.3 * np.sin(0.5 * k), 0.1])
            slot, order = (_lib.SLOT_BDF1, 1) if k == 0 else (_lib.SLOT_BDF2, 2)
            if k % 2 == 0:
                _check(lib, lib.fc_step(h, slot, u.ctypes.data_as(C.c_void_p), None, y_out.ctypes.data_as(C.c_void_p), C.byref(dE), 1, inf.ctypes.data_as(C.c_void_p)))
            else:  # the same step in two halves: enqueue, (host work), collect
                _check(lib, lib.fc_step_begin(h, slot, u.ctypes.data_as(C.c_void_p), None, 1))
                assert lib.fc_step_begin(h, slot, u.ctypes.data_as(C.c_void_p), None, 1) != 0  # one step in flight at a time
                _check(lib, lib.fc_step_end(h, y_out.ctypes.data_as(C.c_void_p), C.byref(dE), inf.ctypes.data_as(C.c_void_p)))
            up = ts.step(order, u_n, u_nn, u)
            u_nn, u_n = u_n, up[: 2 * nn]
            y_ref = np.array([sw[a:b] @ up[sidx[a:b]] for a, b in zip(rp[:-1], rp[1:])])
            assert np.allclose(y_out, y_ref, rtol=1e-9, atol=1e-12)
            assert abs(dE.value - 0.5 * u_n @ (M @ u_n)) < 1e-10 * abs(dE.value)
            assert inf[1] < 1e-10  # residual monitor of the step's solve
            assert (inf[0] > 1) == (mode == "krylov")  # Krylov iterations of the step / refinement sweeps (none)
        sol = np.empty(N)
        _check(lib, lib.fc_get_solution(h, sol))
        assert np.linalg.norm(sol - up) < 1e-10 * np.linalg.norm(up)
    finally:
        lib.fc_destroy(h)


def test_base_flow_iterations_through_the_c_abi_only():
    """fc_set_baseflow_bc / fc_picard_step / fc_newton_step: a host program with nothing but ctypes drives the reference's
    base-flow recipe (Picard, then Newton with dolfin's residual criterion; steadystate.py:60-159) on a channel-like square
    with a parabolic inflow; every iteration's assembly / elimination / factorisation / solve runs on the device.  Checked
    against the oracle's Picard and Newton on the same mesh and data."""
    lib = _lib.load()
    n = 10
    coords, cells, cell_edges, edges = _square_mesh(n)
    nv, ne, nc = len(coords), len(edges), len(cells)
    nn, N = nv + ne, 2 * (nv + ne) + nv
    node_xy = np.vstack([coords, 0.5 * (coords[edges[:, 0]] + coords[edges[:, 1]])])
    x, y = node_xy[:, 0], node_xy[:, 1]
    wall = (y < 1e-12) | (y > 1 - 1e-12) | (x < 1e-12)  # no-slip bottom / top, inflow on the left, outflow on the right
    nodes = np.flatnonzero(wall)
    bc_dofs = np.r_[nodes, nodes + nn].astype(np.int32)
    bc_vals = np.r_[np.where(x[nodes] < 1e-12, 4.0 * y[nodes] * (1 - y[nodes]), 0.0), np.zeros(nodes.size)]
    nu = 1.0 / 50.0
    d = O.Disc(coords, cells, np.hstack([cells, cell_edges + nv]), nn)
    up0 = np.zeros(N)
    up0[:nn] = 1.0
    ref = O.picard(d, nu, up0.copy(), bc_dofs, bc_vals, max_iter=4, tol=1e-12)
    ref = O.newton(d, nu, ref, bc_dofs, bc_vals, max_iter=10)

    h = C.c_void_p()
    _check(lib, lib.fc_create(C.byref(h), 0, nv, ne, nc, np.ascontiguousarray(coords), cells, cell_edges))
    try:
        _check(lib, lib.fc_set_baseflow_bc(h, bc_dofs.size, bc_dofs.ctypes.data_as(C.c_void_p), bc_vals.ctypes.data_as(C.c_void_p)))
        up = up0.copy()
        rel = C.c_double()
        for _ in range(4):
            _check(lib, lib.fc_picard_step(h, nu, up, None, C.byref(rel)))
        assert rel.value < 0.1
        res, r0 = C.c_double(), None
        for it in range(11):
            _check(lib, lib.fc_newton_step(h, nu, up, None, C.byref(res), 0))  # residual only
            r0 = res.value if r0 is None else r0
            if res.value < 1e-10 or res.value < 1e-9 * r0:
                break
            _check(lib, lib.fc_newton_step(h, nu, up, None, C.byref(res), 1))
        assert it < 10, "Newton did not converge"
        assert np.allclose(up[bc_dofs], bc_vals)
        assert np.linalg.norm(up[: 2 * nn] - ref[: 2 * nn]) < 1e-9 * np.linalg.norm(ref[: 2 * nn])
        assert np.linalg.norm(up[2 * nn :] - ref[2 * nn :]) < 1e-8 * np.linalg.norm(ref[2 * nn :])
    finally:
        lib.fc_destroy(h)


def test_batched_steps_through_the_c_abi_only():
    """fc_set_batch / fc_set_state_batch / fc_step_batch with ctypes + numpy only: four simulations with different states
    and controls on one handle against the oracle's time stepper, one simulation at a time."""
    lib = _lib.load()
    n = 10
    coords, cells, cell_edges, edges = _square_mesh(n)
    nv, ne, nc = len(coords), len(edges), len(cells)
    nn, N = nv + ne, 2 * (nv + ne) + nv
    node_xy = np.vstack([coords, 0.5 * (coords[edges[:, 0]] + coords[edges[:, 1]])])
    x, y = node_xy[:, 0], node_xy[:, 1]
    wall = (y < 1e-12) | (y > 1 - 1e-12) | (x < 1e-12)
    nodes = np.flatnonzero(wall)
    bc_dofs = np.r_[nodes, nodes + nn].astype(np.int32)
    prof = np.zeros((bc_dofs.size, 1))
    lid = y[nodes] > 1 - 1e-12
    prof[: nodes.size][lid, 0] = (x[nodes][lid] * (1 - x[nodes][lid])) * 4
    U0 = np.r_[0.4 * y * (2 - y), np.zeros(nn)]
    Re, dt, k = 60.0, 0.01, 4
    h = C.c_void_p()
    _check(lib, lib.fc_create(C.byref(h), 0, nv, ne, nc, np.ascontiguousarray(coords), cells, cell_edges))
    try:
        _check(lib, lib.fc_set_bc(h, bc_dofs.size, bc_dofs.ctypes.data_as(C.c_void_p), 1, prof.ctypes.data_as(C.c_void_p)))
        _check(lib, lib.fc_set_time_scheme(h, dt, 1))
        rp = np.array([0, 2], dtype=np.int32)
        sidx = np.array([7, nn + 9], dtype=np.int32)
        sw = np.array([1.0, -0.5])
        _check(lib, lib.fc_set_sensors(h, 1, rp.ctypes.data_as(C.c_void_p), sidx.ctypes.data_as(C.c_void_p), sw.ctypes.data_as(C.c_void_p)))
        for slot, alpha in ((_lib.SLOT_BDF1, 1.0 / dt), (_lib.SLOT_BDF2, 1.5 / dt)):
            _check(lib, lib.fc_assemble_matrix(h, slot, alpha, 1.0 / Re, U0.ctypes.data_as(C.c_void_p), 1.0, U0.ctypes.data_as(C.c_void_p), 1.0, -1.0, -1.0))
            _check(lib, lib.fc_apply_bc(h, slot))
            _check(lib, lib.fc_setup_solver(h, slot, 0, 2, 0, 0, 1))
        _check(lib, lib.fc_set_batch(h, k))
        rng = np.random.default_rng(11)
        u0 = 0.1 * rng.standard_normal((k, 2 * nn))
        u0[:, bc_dofs] = 0.0
        _check(lib, lib.fc_set_state_batch(h, k, u0, u0, None))
        d = O.Disc(coords, cells, np.hstack([cells, cell_edges + nv]), nn)
        ts = O.TimeStepper(d, Re, dt, U0, bc_dofs, prof)
        M = O.velocity_mass(d)
        u_n, u_nn = u0.copy(), u0.copy()
        y_out, dE, inf = np.zeros((k, 1)), np.zeros(k), np.zeros((k, 4))
        for step in range(4):
            u = np.array([[0.2 * (s + 1) * np.cos(0.4 * step)] for s in range(k)])
            slot, order = (_lib.SLOT_BDF1, 1) if step == 0 else (_lib.SLOT_BDF2, 2)
            _check(lib, lib.fc_step_batch(h, slot, k, u.ctypes.data_as(C.c_void_p), None, y_out.ctypes.data_as(C.c_void_p), dE.ctypes.data_as(C.c_void_p), 1,
                                          inf.ctypes.data_as(C.c_void_p)))
            for s in range(k):
                up = ts.step(order, u_n[s], u_nn[s], u[s])
                u_nn[s], u_n[s] = u_n[s], up[: 2 * nn]
                assert np.isclose(y_out[s, 0], sw @ up[sidx], rtol=1e-9, atol=1e-12)
                assert abs(dE[s] - 0.5 * u_n[s] @ (M @ u_n[s])) < 1e-10 * abs(dE[s])
            assert np.all(inf[:, 1] < 1e-10)
        got = np.empty((k, 2 * nn))
        _check(lib, lib.fc_get_state_batch(h, k, got.ctypes.data_as(C.c_void_p), None, None))
        assert np.linalg.norm(got - u_n) < 1e-10 * np.linalg.norm(u_n)
    finally:
        lib.fc_destroy(h)
