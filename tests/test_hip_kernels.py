"""GPU parity of the individual HIP kernels against the CPU oracle (through the C ABI).

Tolerances: everything is fp64; the device sums in a different (but fixed) order than numpy, so
vectors agree to a few ulps relative to their norm — 1e-12 relative is asserted.
"""
import numpy as np
import pytest

from flowcontrol_amd.fem.mesh import Mesh, read_xdmf_mesh
from flowcontrol_amd.fem.spaces import TaylorHood
from flowcontrol_amd.examples.data import mesh_file  # noqa: E402

pytestmark = pytest.mark.gpu

RTOL = 1e-12


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _smooth_velocity(th, k=1.0):
    x = th.node_coords
    return np.r_[1.0 + 0.3 * np.sin(k * x[:, 0]) * np.cos(0.7 * k * x[:, 1]), 0.2 * np.cos(0.5 * k * x[:, 0] + 0.1) * np.sin(k * x[:, 1])]


@pytest.fixture(scope="module", params=["square8", "O1"])
def setup(request, golden_dir):
    from flowcontrol_amd.device import DeviceSolver
    from oracle import ns_oracle as O

    if request.param == "square8":
        mesh = Mesh.unit_square(8, 8)
    else:
        mesh = read_xdmf_mesh(mesh_file("O1"))
    th = TaylorHood(mesh)
    dev = DeviceSolver(th)
    d = O.Disc.from_taylor_hood(th)
    yield th, dev, d, O
    dev.close()


def test_pattern_matches_oracle(setup):
    th, dev, d, O = setup
    A = O.assemble_matrix(d, mass=1.0, nu=1.0, adv=_smooth_velocity(th), lin=_smooth_velocity(th))
    import scipy.sparse as sp

    P = sp.csr_matrix((np.ones(dev.nnz), dev.colidx, dev.rowptr), shape=(dev.N, dev.N))
    # every oracle nonzero lies inside the device pattern
    A.eliminate_zeros()
    assert (abs(A) > 0).multiply(P).nnz == (abs(A) > 0).nnz


@pytest.mark.parametrize("case", ["bdf2", "picard", "jacobian", "mass"])
def test_matrix_assembly(setup, case):
    th, dev, d, O = setup
    from flowcontrol_amd.device import SLOT_SCRATCH

    U = _smooth_velocity(th)
    kw = dict(
        bdf2=dict(mass=300.0, nu=0.01, adv=U, lin=U),
        picard=dict(mass=0.0, nu=0.01, adv=U, lin=None),
        jacobian=dict(mass=0.0, nu=0.02, adv=U, lin=0.5 * U),
        mass=dict(mass=1.0, nu=0.0, adv=None, lin=None, pressure=0.0, divergence=0.0),
    )[case]
    dev.assemble_matrix(SLOT_SCRATCH, **kw)
    A_dev = dev.matrix(SLOT_SCRATCH)
    A_ref = O.assemble_matrix(d, **kw)
    diff = (A_dev - A_ref).tocoo()
    assert np.abs(diff.data).max() <= RTOL * np.abs(A_ref.data).max()


def test_spmv(setup):
    th, dev, d, O = setup
    from flowcontrol_amd.device import SLOT_SCRATCH

    U = _smooth_velocity(th)
    dev.assemble_matrix(SLOT_SCRATCH, mass=300.0, nu=0.01, adv=U, lin=U)
    A = dev.matrix(SLOT_SCRATCH)
    x = np.random.default_rng(0).standard_normal(dev.N)
    assert _rel(dev.spmv(SLOT_SCRATCH, x), A @ x) < RTOL


def _bc_setup(th):
    """Dirichlet on every boundary facet except the x = xmax side (outflow keeps the pressure
    level unique); two 'actuators' with smooth profiles on them."""
    m = th.mesh
    be = m.boundary_edges()
    be = be[m.edge_midpoints()[be, 0] < m.coords[:, 0].max() - 1e-9]
    nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
    dofs = np.r_[nodes, nodes + th.nn]
    x = th.node_coords[nodes]
    p0 = np.r_[np.sin(x[:, 0] + 2 * x[:, 1]), 0 * x[:, 0]]
    p1 = np.r_[0 * x[:, 0], np.cos(3 * x[:, 0] - x[:, 1])]
    order = np.argsort(dofs)
    return dofs[order], np.stack([p0, p1], axis=1)[order]


@pytest.mark.parametrize("order", [1, 2])
def test_rhs_solve_step(setup, order):
    th, dev, d, O = setup
    from flowcontrol_amd.device import SLOT_BDF1, SLOT_BDF2, SLOT_MASS

    dt, Re = 0.005, 100.0
    U0 = _smooth_velocity(th)
    dofs, prof = _bc_setup(th)
    dev.set_bc(dofs, prof)
    dev.set_time_scheme(dt, True)
    ts = O.TimeStepper(d, Re, dt, U0, dofs, prof, orders=(order,))
    slot = SLOT_BDF1 if order == 1 else SLOT_BDF2
    alpha = (1.0 if order == 1 else 1.5) / dt
    dev.assemble_matrix(slot, mass=alpha, nu=1.0 / Re, adv=U0, lin=U0)
    dev.apply_bc(slot)
    A_bc = dev.matrix(slot)
    dA = (A_bc - ts.A_bc[order]).tocoo()
    assert np.abs(dA.data).max() <= RTOL * np.abs(ts.A_bc[order].data).max()
    dev.assemble_matrix(SLOT_MASS, mass=1.0, nu=0.0, pressure=0.0, divergence=0.0)
    dev.setup_solver(slot, refine=1)
    rng = np.random.default_rng(order)
    u_n = 0.1 * _smooth_velocity(th, 2.0) + 0.01 * rng.standard_normal(2 * th.nn)
    u_nn = 0.1 * _smooth_velocity(th, 1.5)
    dev.set_state(u_n, u_nn, np.zeros(th.nv))
    uc = np.array([0.3, -0.2])
    # RHS (element loop + gather + lifting)
    b_dev = dev.assemble_rhs(slot, uc)
    b_ref = ts.rhs(order, u_n, u_nn, uc)
    assert _rel(b_dev, b_ref) < RTOL
    # solve
    x_dev, info = dev.solve(slot, b_ref)
    x_ref = ts.solve(order, b_ref)
    assert _rel(x_dev, x_ref) < 1e-10
    assert np.linalg.norm(ts.A_bc[order] @ x_dev - b_ref) / np.linalg.norm(b_ref) < 1e-12
    # full step: shift, sensors, energy
    rows = [th.point_eval_row(p, c) for p, c in [((0.31, 0.42), 1), ((0.5, 0.5), 0), ((0.77, 0.13), 2)]] if th.nc < 1000 else [
        th.point_eval_row(p, 1) for p in [(3.0, 0.0), (3.1, 1.0), (3.1, -1.0)]
    ]
    dev.set_sensors(rows)
    y, dE, info = dev.step(slot, uc)
    up_ref = ts.step(order, u_n, u_nn, uc)
    up_dev = dev.get_solution()
    assert _rel(up_dev, up_ref) < 1e-10
    y_ref = np.array([w @ up_ref[i] for i, w in rows])
    assert np.allclose(y, y_ref, rtol=1e-9, atol=1e-12)
    M = O.velocity_mass(d)
    u_new = up_ref[: 2 * th.nn]
    assert np.isclose(dE, 0.5 * u_new @ (M @ u_new), rtol=1e-10)
    g_un, g_unn, g_pn = dev.get_state()
    assert _rel(g_un, u_new) < 1e-10 and np.array_equal(g_unn, u_n)
    assert info[1] < 1e-9  # relative residual before refinement


def test_crank_nicolson_step(setup):
    """CN = BDF1 element rhs − C u_n + ½(f^{n+1}+f^n), LHS with θ=½ on the linear terms
    (reference NSForms._cn, nsforms.py:191-236) — device vs oracle, two consecutive steps."""
    th, dev, d, O = setup
    from flowcontrol_amd.device import SLOT_BDF1, SLOT_SCRATCH

    dt, Re = 0.01, 50.0
    U0 = _smooth_velocity(th)
    dofs, prof = _bc_setup(th)
    x = th.node_coords
    fprof = np.stack([np.r_[np.exp(-((x[:, 0] - 0.4) ** 2 + (x[:, 1] - 0.5) ** 2) / 0.02), 0 * x[:, 0]],
                      np.r_[0 * x[:, 0], np.sin(x[:, 0]) * np.cos(x[:, 1])]])  # (n_act=2, 2nn)
    dev.set_bc(dofs, prof)
    dev.set_force(fprof)
    dev.set_time_scheme(dt, True)
    dev.assemble_matrix(SLOT_BDF1, mass=1.0 / dt, nu=0.5 / Re, adv=U0, lin=U0, adv_scale=0.5, lin_scale=0.5)
    dev.apply_bc(SLOT_BDF1)
    dev.setup_solver(SLOT_BDF1)
    dev.assemble_matrix(SLOT_SCRATCH, mass=0.0, nu=0.5 / Re, adv=U0, lin=U0, adv_scale=0.5, lin_scale=0.5, pressure=0.0, divergence=0.0)
    dev.set_rhs_operator(SLOT_BDF1, dev.matrix(SLOT_SCRATCH)[:, : 2 * th.nn])
    dev.set_sensors([])
    ts = O.TimeStepperCN(d, Re, dt, U0, dofs, prof, force_profiles=fprof.T)
    u_n = 0.1 * _smooth_velocity(th, 2.0)
    dev.set_state(u_n, u_n, np.zeros(th.nv))
    prev = None
    for uc in (np.array([0.3, -0.2]), np.array([-0.1, 0.4])):
        uf = 0.5 * (uc + (0 if prev is None else prev))
        dev.step(SLOT_BDF1, uc, compute_energy=True, u_force=uf)
        up_ref = ts.step(u_n, uc)
        assert _rel(dev.get_solution(), up_ref) < 1e-10
        u_n = up_ref[: 2 * th.nn]
        prev = uc
    dev.set_rhs_operator(SLOT_BDF1, None)
    dev.set_force(None)


def test_block_and_segment_down_sweeps_agree(setup, monkeypatch):
    """The LDS-tiled block kernel (forced on for every down stage, through all of its (lanes per row, rows per slot)
    instantiations: FC_BLOCK_TARGET 1 / 64 / huge = 32 / intermediate / fewest rows per workgroup) and the segment kernel
    (FC_BLOCK_KERNEL=0) produce the same solve to round-off.  The knobs are read when a handle lays out its sweep tables, so
    every variant gets a handle of its own."""
    th, dev0, d, O = setup
    from flowcontrol_amd.device import SLOT_BDF2, DeviceSolver

    dt, Re = 0.005, 100.0
    U0 = _smooth_velocity(th)
    dofs, prof = _bc_setup(th)
    b = np.random.default_rng(3).standard_normal(dev0.N)

    def solve():
        dev = DeviceSolver(th, dev0.device_index)
        try:
            dev.set_bc(dofs, prof)
            dev.set_time_scheme(dt, True)
            dev.assemble_matrix(SLOT_BDF2, mass=1.5 / dt, nu=1.0 / Re, adv=U0, lin=U0)
            dev.apply_bc(SLOT_BDF2)
            dev.setup_solver(SLOT_BDF2)
            return dev.solve(SLOT_BDF2, b)
        finally:
            dev.close()

    monkeypatch.setenv("FC_BLOCK_KERNEL", "0")
    x_seg, _ = solve()
    monkeypatch.delenv("FC_BLOCK_KERNEL")
    monkeypatch.setenv("FC_BLOCK_MIN", "1")
    monkeypatch.setenv("FC_FLAT_ROW", "0")  # the row-lane block kernel on every level ...
    for target in (1, 64, 1 << 30):
        monkeypatch.setenv("FC_BLOCK_TARGET", str(target))
        x_blk, info = solve()
        assert _rel(x_blk, x_seg) < 1e-12
        assert info[1] < 1e-9
    # ... and the flat kernel (fc_nd_flat_block: tiles read as one contiguous stream, row sums from LDS) on the levels of small nodes, through
    # its four loads-per-thread instantiations (tiles of <= 1024 / 2048 / 3072 / 4096 values), row form and column form of the up-sweep
    for up_form in ("row", "column"):
        monkeypatch.setenv("FC_UP_FORM", up_form)
        for flat_row, tile in ((256, 2048), (512, 1024), (512, 3072), (160, 4096)):
            monkeypatch.setenv("FC_FLAT_ROW", str(flat_row))
            monkeypatch.setenv("FC_FLAT_TILE", str(tile))
            x_flat, info = solve()
            assert _rel(x_flat, x_seg) < 1e-12, (up_form, flat_row, tile)
            assert info[1] < 1e-9


@pytest.mark.parametrize("wide", [False, True, "huge"])
def test_device_factorisation_matches_host_multifrontal(setup, wide, monkeypatch):
    """(wide = True: FC_FE_WIDE_NF=64 sends every level of this small mesh through the 64-column kernels — pivot / panels /
    update<64> with the look-ahead inversion in the 64 x 65 LDS overlay — which cavity_fine alone reaches otherwise.
    wide = "huge": FC_FE_HUGE_NF=256 sends every level whose largest front has order >= 256 — the top four of O1 — through
    the 128-column kernels, fc_fe_pivot_huge / panels_huge / update_huge, which only the pinball mesh and the cavity
    meshes reach otherwise.)
    fc_refactor (scatter, extend-add, blocked Gauss-Jordan front elimination on the fp64 matrix cores)
    against the numpy multifrontal of tests/support/nd_numeric.py on the same matrix and tree: factor
    values to round-off, then again after the matrix changed (numeric phase only)."""
    th, dev, d, O = setup
    from flowcontrol_amd.device import SLOT_BDF2
    from tests.support import nd_numeric, ndsolver

    if wide == "huge":
        monkeypatch.setenv("FC_FE_HUGE_NF", "256")
    elif wide:
        monkeypatch.setenv("FC_FE_WIDE_NF", "64")
        monkeypatch.setenv("FC_FE_HUGE_NF", "1000000")
        monkeypatch.setenv("FC_FE_HUGE_MB", "1e9")
    dt, Re = 0.005, 100.0
    dofs, prof = _bc_setup(th)
    dev.set_bc(dofs, prof)
    dev.set_time_scheme(dt, True)
    for scale in (1.0, 1.7):
        U0 = scale * _smooth_velocity(th)
        dev.assemble_matrix(SLOT_BDF2, mass=1.5 / dt, nu=1.0 / Re, adv=U0, lin=U0)
        dev.apply_bc(SLOT_BDF2)
        dev.setup_solver(SLOT_BDF2)  # second round: numeric phase only
        assert SLOT_BDF2 in dev._structured and dev.refactor_ms[SLOT_BDF2] > 0
        A = dev.matrix(SLOT_BDF2)
        host = nd_numeric.factorize_blocks(A, ndsolver.tree_of(dev))
        got = dev.factor_values(SLOT_BDF2)
        assert got.shape == host.vals.shape
        assert np.abs(got - host.vals).max() <= 1e-10 * np.abs(host.vals).max()
        b = np.random.default_rng(5).standard_normal(dev.N)
        x, info = dev.solve(SLOT_BDF2, b)
        assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-12
        assert info[1] < 1e-12


def test_pivoting_inside_the_pivot_block(setup):
    """The pivot blocks are inverted with (threshold) partial pivoting among their own rows.  On the assembled operators the
    diagonal always passes the threshold, so the exchange path would never run: here the ux- and uy-rows of a few P2 nodes are
    swapped (both rows have the same column pattern; P A x = P b has the same solution), which puts O(1) convection
    couplings on the diagonal and the O(mass / dt) entries next to them — without the row exchange the elimination would
    divide by those small entries (or by zero).  Nodes are taken from leaves of at most 32 pivot rows, so that both rows sit
    in one pivot block."""
    th, dev, d, O = setup
    from flowcontrol_amd.device import SLOT_BDF2

    dt, Re = 0.005, 100.0
    dofs, prof = _bc_setup(th)
    dev.set_bc(dofs, prof)
    dev.set_time_scheme(dt, True)
    U0 = _smooth_velocity(th)
    dev.assemble_matrix(SLOT_BDF2, mass=1.5 / dt, nu=1.0 / Re, adv=U0, lin=U0)
    dev.apply_bc(SLOT_BDF2)
    dev.setup_solver(SLOT_BDF2)
    A = dev.matrix(SLOT_BDF2).tocsr()
    from tests.support import ndsolver

    t = ndsolver.tree_of(dev)
    is_bc = np.zeros(dev.N, dtype=bool)
    is_bc[dofs] = True
    K = t.depth
    swaps = []
    for n in range(t.nnodes(K)):
        i0, i1 = int(t.node_ptr[K][n]), int(t.node_ptr[K][n + 1])
        if not 2 <= i1 - i0 <= 32:
            continue
        own = set(t.perm[i0:i1].tolist())
        for q in sorted(own):
            if q < th.nn and q + th.nn in own and not is_bc[q] and not is_bc[q + th.nn]:
                ra, rb = A.indices[A.indptr[q] : A.indptr[q + 1]], A.indices[A.indptr[q + th.nn] : A.indptr[q + th.nn + 1]]
                if ra.size == rb.size and np.array_equal(ra, rb):
                    swaps.append((q, q + th.nn))
                    break
        if len(swaps) >= 12:
            break
    assert len(swaps) >= 3, "no leaf holds both velocity dofs of a free node"
    vals = A.data.copy()
    for a, b_ in swaps:
        sa, sb = slice(A.indptr[a], A.indptr[a + 1]), slice(A.indptr[b_], A.indptr[b_ + 1])
        vals[sa], vals[sb] = A.data[sb].copy(), A.data[sa].copy()
        # the swapped diagonal must be far below the column's largest candidate: the exchange is not optional
        Ad = A[[a, b_]][:, [a, b_]].toarray()
        assert abs(Ad[1, 0]) < 1e-2 * abs(Ad[0, 0]) and abs(Ad[0, 1]) < 1e-2 * abs(Ad[1, 1])
    try:
        dev.set_matrix_values(SLOT_BDF2, vals)
        dev.refactor(SLOT_BDF2)
        b = np.random.default_rng(11).standard_normal(dev.N)
        x, info = dev.solve(SLOT_BDF2, b)
        Ap = dev.matrix(SLOT_BDF2)
        assert np.linalg.norm(Ap @ x - b) / np.linalg.norm(b) < 1e-11
        pb = b.copy()
        for a, b_ in swaps:
            pb[a], pb[b_] = b[b_], b[a]
        assert np.linalg.norm(A @ x - pb) / np.linalg.norm(b) < 1e-11  # the solution of the un-swapped system for the swapped-back right-hand side
    finally:
        dev.set_matrix_values(SLOT_BDF2, A.data)
        dev.refactor(SLOT_BDF2)


def test_bicgstab_with_exact_and_lagged_factors(setup):
    """Device BiCGStab (fc_solve, FC_METHOD_BICGSTAB), right-preconditioned by the factor sweeps: one
    iteration with the operator's own factors; a handful with the factors of a different (earlier)
    operator — fc_update_operator — and the same solution as a fresh factorisation."""
    th, dev, d, O = setup
    from flowcontrol_amd._lib import FcError
    from flowcontrol_amd.device import SLOT_BDF2

    dt, Re = 0.005, 100.0
    dofs, prof = _bc_setup(th)
    dev.set_bc(dofs, prof)
    dev.set_time_scheme(dt, True)
    U0 = _smooth_velocity(th)
    dev.assemble_matrix(SLOT_BDF2, mass=1.5 / dt, nu=1.0 / Re, adv=U0, lin=U0)
    dev.apply_bc(SLOT_BDF2)
    dev.setup_solver(SLOT_BDF2)
    b = np.random.default_rng(21).standard_normal(dev.N)
    x_direct, _ = dev.solve(SLOT_BDF2, b)
    try:
        dev.set_solver_options(refine=20, method="bicgstab", rtol=1e-12)
        x_k, info = dev.solve(SLOT_BDF2, b)
        assert info[0] == 1 and info[1] < 1e-12  # exact preconditioner: one iteration
        assert _rel(x_k, x_direct) < 1e-11
        # a different operator (stronger, rotated advection; dt halved) behind the OLD factors
        U1 = 1.6 * _smooth_velocity(th, 1.3)
        dev.assemble_matrix(SLOT_BDF2, mass=3.0 / dt, nu=1.0 / Re, adv=U1, lin=U1)
        dev.apply_bc(SLOT_BDF2)
        dev.update_operator(SLOT_BDF2)
        A1 = dev.matrix(SLOT_BDF2)
        x_lag, info = dev.solve(SLOT_BDF2, b)
        assert 1 < info[0] <= 20 and info[1] < 1e-11
        assert np.linalg.norm(A1 @ x_lag - b) / np.linalg.norm(b) < 1e-11
        # iteration cap too small -> loud failure
        dev.set_solver_options(refine=1, method="bicgstab", rtol=1e-14)
        with pytest.raises(FcError):
            dev.solve(SLOT_BDF2, b)
    finally:
        dev.set_solver_options(0, True)
    dev.setup_solver(SLOT_BDF2)  # fresh factors of the new operator: direct solve agrees with the Krylov one
    x_new, _ = dev.solve(SLOT_BDF2, b)
    assert _rel(x_lag, x_new) < 1e-10


def test_error_paths_of_the_factorisation_and_krylov_entry_points(golden_dir):
    """Call-order and argument errors come back as status codes with a message, never as a crash."""
    import ctypes as C

    from flowcontrol_amd import _lib
    from flowcontrol_amd._lib import FC_ERR_INVALID, FcError, check
    from flowcontrol_amd.device import SLOT_BDF2, DeviceSolver

    th = TaylorHood(Mesh.unit_square(6, 6))
    dev = DeviceSolver(th, 0)
    lib, h = dev.lib, dev._h
    FC_ERR_NOT_READY = -5
    assert lib.fc_refactor(h, SLOT_BDF2, None) == FC_ERR_NOT_READY  # no plan yet
    assert lib.fc_update_operator(h, SLOT_BDF2) == FC_ERR_NOT_READY
    assert lib.fc_set_front_shifts(h, 1, np.zeros(1, np.int64), np.ones(1)) == FC_ERR_NOT_READY
    assert b"fc_factor_plan" in lib.fc_last_error()
    assert lib.fc_set_solver_options(h, _lib.METHOD_BICGSTAB, 0, 1e-10, 1) == FC_ERR_INVALID  # needs >= 1 iteration
    assert lib.fc_set_solver_options(h, _lib.METHOD_GMRES, 0, 1e-10, 1) == FC_ERR_INVALID  # needs >= 1 iteration
    assert lib.fc_set_solver_options(h, 9, 5, 1e-10, 1) == FC_ERR_INVALID  # unknown method
    assert lib.fc_refactor(h, 7, None) == FC_ERR_INVALID
    # a working system, then misuse
    dofs, prof = _bc_setup(th)
    dev.set_bc(dofs, prof)
    dev.set_time_scheme(0.01, True)
    dev.assemble_matrix(SLOT_BDF2, mass=150.0, nu=0.01)
    dev.apply_bc(SLOT_BDF2)
    dev.setup_solver(SLOT_BDF2)
    assert lib.fc_set_front_shifts(h, 1, np.array([1 << 40], dtype=np.int64), np.ones(1)) == FC_ERR_INVALID
    with pytest.raises(ValueError):
        dev.set_pressure_pin(3)  # a velocity dof
    dev.set_state(np.zeros(2 * th.nn), np.zeros(2 * th.nn), np.zeros(th.nv))
    dev.set_sensors([th.point_eval_row((0.3, 0.4), 0)])
    dev.set_solver_options(refine=5, method="bicgstab")  # a Krylov method inside the step: one iteration with exact factors
    y_k, _, info_k = dev.step(SLOT_BDF2, np.zeros(prof.shape[1]))
    assert np.isfinite(y_k).all() and info_k[0] <= 1  # (zero state, zero actuation: b = 0, no iteration needed)
    dev.set_solver_options(0, True)
    y, dE, info = dev.step(SLOT_BDF2, np.zeros(prof.shape[1]))
    assert np.isfinite(y).all() and info[1] < 1e-10
    assert lib.fc_set_stage_diag(h, 5, np.ones(dev.N)) == FC_ERR_INVALID
    n = C.c_int64(dev._n_factor_values + 1)
    assert lib.fc_get_factor_values(h, SLOT_BDF2, n, np.empty(n.value)) == FC_ERR_INVALID
    check(lib.fc_refactor(h, SLOT_BDF2, None))
    dev.close()
