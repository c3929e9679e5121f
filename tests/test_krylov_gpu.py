"""Krylov drivers of ``fc_solve`` (north_star: "HIP BiCGStab/GMRES"; plug-in point flowsolver.py:812-814) against the
oracle's sparse direct solve: device-resident BiCGStab and restarted GMRES(30), right-preconditioned by the selected-
inverse factors — exact factors (one iteration), factors of an earlier operator (a few: what the Newton / Picard
iterations of the base-flow solvers do between two refactorisations)."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from flowcontrol_amd.fem.mesh import read_xdmf_mesh
from flowcontrol_amd.fem.spaces import TaylorHood
from flowcontrol_amd.examples.data import mesh_file  # noqa: E402

pytestmark = pytest.mark.gpu


def _bc(th):
    m = th.mesh
    be = m.boundary_edges()
    be = be[m.edge_midpoints()[be, 0] < m.coords[:, 0].max() - 1e-9]
    nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
    return np.sort(np.r_[nodes, nodes + th.nn])


@pytest.mark.parametrize("mesh", ["O1", "cavity_coarse"])
def test_krylov_methods_match_the_direct_solve(mesh, golden_dir):
    from flowcontrol_amd.device import SLOT_BDF2, DeviceSolver
    from oracle import ns_oracle as O

    th = TaylorHood(read_xdmf_mesh(mesh_file(mesh)))
    dev = DeviceSolver(th)
    d = O.Disc.from_taylor_hood(th)
    x = th.node_coords
    U0 = np.r_[1.0 + 0.3 * np.sin(x[:, 0]) * np.cos(0.7 * x[:, 1]), 0.2 * np.cos(0.5 * x[:, 0] + 0.1) * np.sin(x[:, 1])]
    U1 = U0 + 0.4 * np.r_[np.cos(2.0 * x[:, 1]), np.sin(1.5 * x[:, 0])]  # the operator the old factors must still precondition
    dofs = _bc(th)
    dev.set_bc(dofs, np.zeros((dofs.size, 1)))
    dt, Re = 0.005, 100.0
    dev.assemble_matrix(SLOT_BDF2, mass=1.5 / dt, nu=1.0 / Re, adv=U0, lin=U0)
    dev.apply_bc(SLOT_BDF2)
    dev.setup_solver(SLOT_BDF2)
    A0 = dev.matrix(SLOT_BDF2).tocsc()
    rng = np.random.default_rng(3)
    b = rng.standard_normal(dev.N)
    b[dofs] = 0.0
    x0 = spla.splu(A0).solve(b)
    report = {}
    for method in ("bicgstab", "gmres"):
        dev.set_solver_options(refine=60, method=method, rtol=1e-12)
        xs, info = dev.solve(SLOT_BDF2, b)
        assert np.linalg.norm(xs - x0) <= 1e-10 * np.linalg.norm(x0)
        assert info[0] <= 2 and info[1] < 1e-11  # exact preconditioner: one iteration
        report[method + "_exact"] = int(info[0])
    # new operator, old factors
    dev.assemble_matrix(SLOT_BDF2, mass=1.5 / dt, nu=1.0 / Re, adv=U1, lin=U1)
    dev.apply_bc(SLOT_BDF2)
    dev.update_operator(SLOT_BDF2)
    A1 = dev.matrix(SLOT_BDF2).tocsc()
    x1 = spla.splu(A1).solve(b)
    assert np.linalg.norm(x1 - x0) > 1e-5 * np.linalg.norm(x0)  # the operators do differ
    for method in ("bicgstab", "gmres"):
        dev.set_solver_options(refine=60, method=method, rtol=1e-12)
        xs, info = dev.solve(SLOT_BDF2, b)
        assert np.linalg.norm(xs - x1) <= 1e-10 * np.linalg.norm(x1), (method, info)
        assert 1 < info[0] <= 40 and info[1] < 1e-11
        report[method + "_lagged"] = int(info[0])
    print(f"[{mesh}] Krylov iterations: {report}")
    dev.close()


@pytest.mark.parametrize("method", ["gmres", "bicgstab"])
def test_time_steps_with_truncated_factors(method, tmp_path_factory, golden_dir):
    """Memory-lean mode: only the tree levels >= 1 are factorised (the root's pivot block, the largest front, is never
    formed), the root's Schur complement is replaced by a diagonal estimate and every step is solved by GMRES / BiCGStab
    preconditioned with those truncated factors.  The trajectory must still be the oracle's."""
    from flowcontrol_amd._lib import SLOT_BDF2
    from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver
    from flowcontrol_amd.fem.spaces import Function
    from flowcontrol_amd.flowsolverparameters import ParamIC

    g = np.load(golden_dir / "cylinder_O1.npz")
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp_path_factory.mktemp(f"trunc_{method}"), num_steps=6)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    fs.nd_truncate, fs.krylov_method, fs.krylov_max_iter = 1, method, 800
    U0, P0 = Function(fs.W, g["UP0"]).split()
    fs._assign_steady_state(U0, P0)
    fs.initialize_time_stepping(ic=None)
    its = []
    for _ in range(6):
        fs.step([0.0, 0.0])
        assert fs.solve_info[1] < 1e-10  # the tail's residual monitor checks the Krylov result
        its.append(int(fs.solve_info[0]))
    assert 1 < max(its) <= 800
    dev = fs.th.device()
    full = 22279303  # factor values of the full selected inverse on this mesh
    assert dev._n_factor_values < 0.93 * full
    ts = fs.timeseries
    y = ts[["y_meas_1", "y_meas_2", "y_meas_3"]].to_numpy()
    assert np.linalg.norm(y - g["ol_y"][:7]) <= 1e-8 * np.linalg.norm(g["ol_y"][:7])
    assert np.linalg.norm(ts["dE"].to_numpy() - g["ol_dE"][:7]) <= 1e-8 * np.linalg.norm(g["ol_dE"][:7])
    print(f"[truncated factors, {method}] stored factor values {dev._n_factor_values} of {full}; iterations per step {its}")
    fs.th.release_device()


@pytest.mark.parametrize("case,bits", [("cylinder", 16), ("cylinder", 32), ("cavity", 16)])
def test_time_steps_with_compressed_factors(case, bits, tmp_path_factory, golden_dir):
    """The Krylov mode that works without the full fp64 factorisation: the selected inverse is stored in bfloat16 (25 % of
    the factor memory) or fp32 (50 %) — the fp64 values never exist on the device — and every step is solved by the device
    GMRES right-preconditioned with those compressed factors.  O1 (cylinder) and cavity_coarse must follow the oracle's
    series to 1e-8 at a handful of iterations per step (bar: <= 40)."""
    from flowcontrol_amd._lib import SLOT_BDF2
    from flowcontrol_amd.fem.spaces import Function
    from flowcontrol_amd.flowsolverparameters import ParamIC

    if case == "cylinder":
        from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver as Case

        g = np.load(golden_dir / "cylinder_O1.npz")
        fs = Case.make_default(Re=100, path_out=tmp_path_factory.mktemp(f"lp_{case}_{bits}"), num_steps=10)
        fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
        y_ref, dE_ref, u = g["ol_y"][:11], g["ol_dE"][:11], [0.0, 0.0]
    else:
        from flowcontrol_amd.examples.cavity.cavityflowsolver import CavityFlowSolver as Case

        g = np.load(golden_dir / "cavity_coarse.npz")
        fs = Case.make_default(Re=7500, path_out=tmp_path_factory.mktemp(f"lp_{case}_{bits}"), num_steps=10)
        y_ref, dE_ref, u = g["y"][:11], g["dE"][:11], [0.0]
    fs.factor_bits, fs.krylov_method, fs.krylov_max_iter, fs.krylov_rtol = bits, "gmres", 60, 1e-12
    U0, P0 = Function(fs.W, g["UP0"]).split()
    fs._assign_steady_state(U0, P0)
    fs.initialize_time_stepping(ic=None)
    its = []
    for _ in range(10):
        fs.step(u)
        assert fs.solve_info[1] < 1e-10  # the tail's residual monitor checks the Krylov result against the fp64 operator
        its.append(int(fs.solve_info[0]))
    assert 1 <= max(its) <= 40, its
    dev = fs.th.device()
    got_bits, nbytes = dev.factor_storage(SLOT_BDF2)
    values = dev._n_factor_values
    assert got_bits == bits and nbytes <= (bits / 64.0) * 8.0 * values + 1024  # 25 % / 50 % of the fp64 factor bytes
    ts = fs.timeseries
    y = ts[[c for c in ts.columns if c.startswith("y_meas_")]].to_numpy()
    assert np.linalg.norm(y - y_ref) <= 1e-8 * np.linalg.norm(y_ref)
    assert np.linalg.norm(ts["dE"].to_numpy() - dE_ref) <= 1e-8 * np.linalg.norm(dE_ref)
    print(f"[compressed factors, {case}, {bits} bit] {nbytes / 1e6:.1f} MB of factor values ({values} values); GMRES iterations per step {its}")
    fs.th.release_device()


def test_factors_that_miss_the_acceptance_residual_precondition_gmres(golden_dir):
    """Steady Oseen operators far beyond the mesh's resolution (Re ~ 10^4 on O1): pivoting confined to the pivot blocks loses digits
    on some of them.  ``fc_accept_factors`` then keeps the factors as a GMRES preconditioner instead of refusing them; the solve is as
    good as the oracle's sparse LU (same normwise backward error), and the batched API -- which applies factors directly -- says no."""
    from flowcontrol_amd._lib import FcError
    from flowcontrol_amd.device import SLOT_BDF2, DeviceSolver

    th = TaylorHood(read_xdmf_mesh(mesh_file("O1")))
    dev = DeviceSolver(th)
    x = th.node_coords
    dofs = _bc(th)
    dev.set_bc(dofs, np.zeros((dofs.size, 1)))
    rng = np.random.default_rng(2026)
    seen = {False: 0, True: 0}
    for i in range(16):
        amp, kx, ky, ph = rng.uniform(0.2, 3.0), rng.uniform(0.2, 2.0), rng.uniform(0.2, 2.0), rng.uniform(0, 6.28)
        U0 = amp * np.r_[1.0 + 0.5 * np.sin(kx * x[:, 0] + ph) * np.cos(ky * x[:, 1]), 0.4 * np.cos(ky * x[:, 0]) * np.sin(kx * x[:, 1] + ph)]
        dev.assemble_matrix(SLOT_BDF2, mass=0.0, nu=1.3e-4, adv=U0, lin=U0)
        dev.apply_bc(SLOT_BDF2)
        if i == 0:
            dev.setup_solver(SLOT_BDF2)
        else:
            dev.refactor(SLOT_BDF2)
        inexact = dev.factors_inexact[SLOT_BDF2]
        seen[inexact] += 1
        if seen[inexact] > 2:
            continue  # two of each kind are checked in full
        A = dev.matrix(SLOT_BDF2).tocsc()
        b = rng.standard_normal(dev.N)
        b[dofs] = 0.0
        xs, info = dev.solve(SLOT_BDF2, b)
        x_lu = spla.splu(A).solve(b)
        norm_a = np.sqrt((A.data**2).sum())

        def backward(v):
            return np.linalg.norm(A @ v - b) / (norm_a * np.linalg.norm(v) + np.linalg.norm(b))

        assert backward(xs) <= max(10.0 * backward(x_lu), 1e-13), (i, inexact, backward(xs), backward(x_lu))
        assert np.linalg.norm(A @ xs - b) <= 1e-8 * np.linalg.norm(b)
        assert np.linalg.norm(xs - x_lu) <= 1e-6 * np.linalg.norm(x_lu)
        if inexact:
            assert 1 <= info[0] <= 40  # GMRES iterations
            dev.set_batch(2)
            with pytest.raises(FcError, match="inexact"):
                dev.solve_batch(SLOT_BDF2, np.stack([b, b]))
            dev.set_batch(0)
    print("inexact / exact operators:", seen)
    assert seen[True] >= 1, "no operator of this family tripped the acceptance residual: the test lost its subject"
    # a well-conditioned operator afterwards clears the flag
    dev.assemble_matrix(SLOT_BDF2, mass=300.0, nu=1e-2, adv=U0, lin=U0)
    dev.apply_bc(SLOT_BDF2)
    dev.refactor(SLOT_BDF2)
    assert not dev.factors_inexact[SLOT_BDF2]
    dev.close()


@pytest.mark.parametrize("mesh", ["O1", "cavity_coarse"])
def test_the_two_forms_of_the_up_sweep_solve_alike(mesh, monkeypatch):
    """FC_UP_FORM=row: one segment-row launch per tree level (the default while the factors stay in the Infinity Cache);
    FC_UP_FORM=column: per level the nodes' dense -L blocks through the LDS-tiled block kernel + one fold launch (the default for
    factors that stream from HBM).  Same values, same sums in a different association: solutions agree to round-off and both
    meet the direct solve; the column form runs two launches per level."""
    from flowcontrol_amd.device import SLOT_BDF2, DeviceSolver

    th = TaylorHood(read_xdmf_mesh(mesh_file(mesh)))
    x = th.node_coords
    U0 = np.r_[1.0 + 0.3 * np.sin(x[:, 0]) * np.cos(0.7 * x[:, 1]), 0.2 * np.cos(0.5 * x[:, 0] + 0.1) * np.sin(x[:, 1])]
    dofs = _bc(th)
    rng = np.random.default_rng(11)
    b = rng.standard_normal(th.N)
    b[dofs] = 0.0
    sol, launches = {}, {}
    for form in ("row", "column"):
        monkeypatch.setenv("FC_UP_FORM", form)  # read when the handle is created
        dev = DeviceSolver(th)
        dev.set_bc(dofs, np.zeros((dofs.size, 1)))
        dev.assemble_matrix(SLOT_BDF2, mass=300.0, nu=0.01, adv=U0, lin=U0)
        dev.apply_bc(SLOT_BDF2)
        dev.setup_solver(SLOT_BDF2)
        xs, info = dev.solve(SLOT_BDF2, b)
        assert info[1] < 1e-12
        if form == "row":
            x0 = spla.splu(dev.matrix(SLOT_BDF2).tocsc()).solve(b)
            assert np.linalg.norm(xs - x0) <= 1e-10 * np.linalg.norm(x0)
        sol[form] = xs
        launches[form] = dev.bench_sweeps(SLOT_BDF2, reps=3)[1]
        dev.close()
    assert np.linalg.norm(sol["row"] - sol["column"]) <= 1e-13 * np.linalg.norm(sol["row"])
    n_up = (launches["row"] - 1) // 2  # row form: depth up launches + root + depth down launches
    assert launches["column"] == launches["row"] + n_up, launches
