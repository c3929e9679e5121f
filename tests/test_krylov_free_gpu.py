"""Factorisation-free Krylov mode (``fc_setup_krylov``; north_star: "HIP BiCGStab/GMRES with CSR SpMV and block-Jacobi/ILU(0)
preconditioning"; plug-in point ``flowsolver.py:812-814``): NOTHING is factorised — the device GMRES / BiCGStab run on the
permuted system matrix, right-preconditioned by the SIMPLE / AMG block preconditioner (damped-Jacobi sweeps on the velocity
block, one smoothed-aggregation V-cycle on the pressure Schur complement ``B diag(F)^-1 Bt``).  Against the oracle's sparse
direct solve and golden time series."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from flowcontrol_amd.examples.data import mesh_file
from flowcontrol_amd.fem.mesh import read_xdmf_mesh
from flowcontrol_amd.fem.spaces import TaylorHood

pytestmark = pytest.mark.gpu


def _bc(th):
    m = th.mesh
    be = m.boundary_edges()
    be = be[m.edge_midpoints()[be, 0] < m.coords[:, 0].max() - 1e-9]
    nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
    return np.sort(np.r_[nodes, nodes + th.nn])


@pytest.mark.parametrize("mesh,dt,Re", [("O1", 0.005, 100.0), ("cavity_coarse", 4e-4, 7500.0)])
def test_factor_free_solve_matches_the_direct_solve(mesh, dt, Re):
    from flowcontrol_amd.device import SLOT_BDF2, DeviceSolver
    from flowcontrol_amd._lib import FcError

    th = TaylorHood(read_xdmf_mesh(mesh_file(mesh)))
    dev = DeviceSolver(th)
    x = th.node_coords
    U0 = np.r_[1.0 + 0.3 * np.sin(x[:, 0]) * np.cos(0.7 * x[:, 1]), 0.2 * np.cos(0.5 * x[:, 0] + 0.1) * np.sin(x[:, 1])]
    dofs = _bc(th)
    dev.set_bc(dofs, np.zeros((dofs.size, 1)))
    dev.assemble_matrix(SLOT_BDF2, mass=1.5 / dt, nu=1.0 / Re, adv=U0, lin=U0)
    dev.apply_bc(SLOT_BDF2)
    A = dev.matrix(SLOT_BDF2).tocsc()
    rng = np.random.default_rng(3)
    b = rng.standard_normal(dev.N)
    b[dofs] = 0.0
    x0 = spla.splu(A).solve(b)
    report = {}
    for method in ("gmres", "bicgstab"):
        info = dev.setup_krylov(SLOT_BDF2, sweeps=3, method=method, max_iter=300, rtol=1e-12)
        assert info["bytes"] < 6 * 12 * A.nnz  # O(nnz): nothing that grows like the fill
        xs, si = dev.solve(SLOT_BDF2, b)
        assert np.linalg.norm(xs - x0) <= 1e-10 * np.linalg.norm(x0), (method, si)
        assert si[1] < 1e-11 and 1 < si[0] <= 100, (method, si)  # the bar: <= 100 iterations without any factorisation
        report[method] = int(si[0])
    print(f"[{mesh}] factorisation-free iterations to 1e-12: {report}; device bytes {info['bytes'] / 1e6:.1f} MB, AMG levels {info['amg_levels']}")
    # no factors: the direct apply has nothing to apply
    dev.set_solver_options(refine=0, method="refine")
    with pytest.raises(FcError):
        dev.solve(SLOT_BDF2, b)
    # ... until the slot is factorised: the same slot switches mode
    dev.setup_solver(SLOT_BDF2)
    xs, si = dev.solve(SLOT_BDF2, b)
    assert np.linalg.norm(xs - x0) <= 1e-10 * np.linalg.norm(x0)
    dev.close()


def test_factor_free_solve_at_full_size_uses_a_fraction_of_the_factor_memory():
    """BASELINE config 3's mesh (cavity_fine, 877 k dofs): the factorisation-free solve against the SAME device's direct solve (the
    oracle's SuperLU needs minutes here; the direct path is pinned to the oracle on this mesh by tests/test_configs_gpu.py), the
    iteration bar of the small meshes (mesh-independent: ~40 from a zero guess), and the point of the mode: what it holds is O(nnz) —
    a quarter of the 3.9 GB of factor values the direct mode streams."""
    from flowcontrol_amd.device import SLOT_BDF2, DeviceSolver

    th = TaylorHood(read_xdmf_mesh(mesh_file("cavity_fine")))
    dev = DeviceSolver(th)
    x = th.node_coords
    dt, Re = 4e-4, 7500.0
    U0 = np.r_[0.5 * np.sin(2.0 * x[:, 0]) * np.cos(1.3 * x[:, 1]), 0.3 * np.cos(1.1 * x[:, 0] + 0.2) * np.sin(2.0 * x[:, 1])]
    dofs = _bc(th)
    dev.set_bc(dofs, np.zeros((dofs.size, 1)))
    dev.assemble_matrix(SLOT_BDF2, mass=1.5 / dt, nu=1.0 / Re, adv=U0, lin=U0)
    dev.apply_bc(SLOT_BDF2)
    b = np.random.default_rng(5).standard_normal(dev.N)
    b[dofs] = 0.0
    info = dev.setup_krylov(SLOT_BDF2, sweeps=2, method="gmres", max_iter=300, rtol=1e-12)
    xs, si = dev.solve(SLOT_BDF2, b)
    assert si[1] < 1e-11 and 1 < si[0] <= 100, si
    dev.setup_solver(SLOT_BDF2)  # the same slot, factorised now
    xd, sd = dev.solve(SLOT_BDF2, b)
    assert sd[1] < 1e-11
    assert np.linalg.norm(xs - xd) <= 1e-9 * np.linalg.norm(xd)
    factor_bytes = 8.0 * dev.factor_nnz[SLOT_BDF2]
    assert info["bytes"] < 0.3 * factor_bytes, (info["bytes"], factor_bytes)
    print(f"[cavity_fine] factorisation-free: {int(si[0])} GMRES iterations, {info['bytes'] / 1e6:.0f} MB held against {factor_bytes / 1e6:.0f} MB of factor values")
    dev.close()


def test_factor_free_cavity_with_body_force_follows_the_oracle(tmp_path_factory, golden_dir):
    """The cavity case (Re = 7500, dt = 4e-4, FORCE actuator: the body force enters the element loop, wall-shear integral sensor) with
    no factorisation: 10 steps of the reference's regression scenario against the oracle's series."""
    from flowcontrol_amd.examples.cavity.cavityflowsolver import CavityFlowSolver
    from flowcontrol_amd.fem.spaces import Function

    g = np.load(golden_dir / "cavity_coarse.npz")
    fs = CavityFlowSolver.make_default(Re=7500, path_out=tmp_path_factory.mktemp("free_cavity"), num_steps=10)
    fs.krylov_precond, fs.krylov_method, fs.krylov_max_iter, fs.krylov_rtol = "schur_amg", "gmres", 300, 1e-11
    U0, P0 = Function(fs.W, g["UP0"]).split()
    fs._assign_steady_state(U0, P0)
    fs.initialize_time_stepping(ic=None)
    its = []
    for _ in range(10):
        fs.step([0.0])
        assert fs.solve_info[1] < 1e-9
        its.append(int(fs.solve_info[0]))
    assert 1 <= max(its) <= 100
    ts = fs.timeseries
    y = ts[[c for c in ts.columns if c.startswith("y_meas")]].to_numpy()
    assert np.linalg.norm(y - g["y"][:11]) <= 1e-8 * np.linalg.norm(g["y"][:11])
    assert np.linalg.norm(ts["dE"].to_numpy() - g["dE"][:11]) <= 1e-8 * np.linalg.norm(g["dE"][:11])
    print(f"[factorisation-free, cavity_coarse] iterations per step {its}; held {fs.th.device().krylov_info(1)['bytes'] / 1e6:.1f} MB")
    fs.th.release_device()


@pytest.mark.parametrize("method", ["gmres", "bicgstab"])
def test_factor_free_time_steps_follow_the_oracle(method, tmp_path_factory, golden_dir):
    """50 open-loop steps of the cylinder case (O1, BDF1 then BDF2) with NO factorisation: sensors and energy within 1e-8 of
    the oracle's series; iterations per step reported (they start from the previous solution)."""
    from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver
    from flowcontrol_amd.fem.spaces import Function
    from flowcontrol_amd.flowsolverparameters import ParamIC

    n = 50
    g = np.load(golden_dir / "cylinder_O1.npz")
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp_path_factory.mktemp(f"free_{method}"), num_steps=n)
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    fs.krylov_precond, fs.krylov_method, fs.krylov_max_iter, fs.krylov_rtol = "schur_amg", method, 300, 1e-11
    U0, P0 = Function(fs.W, g["UP0"]).split()
    fs._assign_steady_state(U0, P0)
    fs.initialize_time_stepping(ic=None)
    its = []
    for _ in range(n):
        fs.step([0.0, 0.0])
        assert fs.solve_info[1] < 1e-9  # the tail's residual monitor checks the Krylov result against the matrix
        its.append(int(fs.solve_info[0]))
    assert 1 <= max(its) <= 100
    dev = fs.th.device()
    assert dev.factor_nnz.get(1, 0) == 0  # nothing was factorised
    ts = fs.timeseries
    y = ts[["y_meas_1", "y_meas_2", "y_meas_3"]].to_numpy()
    assert np.linalg.norm(y - g["ol_y"][: n + 1]) <= 1e-8 * np.linalg.norm(g["ol_y"][: n + 1])
    assert np.linalg.norm(ts["dE"].to_numpy() - g["ol_dE"][: n + 1]) <= 1e-8 * np.linalg.norm(g["ol_dE"][: n + 1])
    print(f"[factorisation-free, {method}] iterations per step: first {its[:3]} mean {np.mean(its):.1f} max {max(its)}; "
          f"held {dev.krylov_info(1)['bytes'] / 1e6:.1f} MB")
    fs.th.release_device()


def test_factor_free_crank_nicolson_steps_follow_the_oracle(tmp_path_factory, golden_dir):
    """time_scheme="cn" (NSForms._cn, nsforms.py:191-236: half of the linear terms explicit, an extra operator in the right-hand side)
    with no factorisation: 8 actuated steps against the oracle's Crank-Nicolson stepper."""
    from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver
    from flowcontrol_amd.fem.spaces import Function
    from flowcontrol_amd.flowsolverparameters import ParamIC
    from oracle import ns_oracle as O

    g = np.load(golden_dir / "cylinder_O1.npz")
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp_path_factory.mktemp("free_cn"), num_steps=8)
    fs.params_solver.time_scheme = "cn"
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    fs.krylov_precond, fs.krylov_method, fs.krylov_max_iter, fs.krylov_rtol = "schur_amg", "gmres", 300, 1e-11
    U0, P0 = Function(fs.W, g["UP0"]).split()
    fs._assign_steady_state(U0, P0)
    fs.initialize_time_stepping(ic=None)
    th = fs.th
    dofs, prof = fs._bc_tables()
    ts = O.TimeStepperCN(O.Disc.from_taylor_hood(th), 100.0, 0.005, g["UP0"][: 2 * th.nn], dofs, prof)
    rows = [s.row(fs) for s in fs.params_control.sensor_list]
    u_n = fs.fields.ic.u.vector().get_local()
    ys, its = [], []
    for k in range(8):
        uc = np.array([0.05 * np.sin(0.4 * k), -0.03])
        fs.step(uc)
        its.append(int(fs.solve_info[0]))
        up = ts.step(u_n, uc)
        u_n = up[: 2 * th.nn]
        ys.append([w @ up[i] for i, w in rows])
    y_dev = fs.timeseries[["y_meas_1", "y_meas_2", "y_meas_3"]].to_numpy()[1:]
    assert np.linalg.norm(y_dev - np.array(ys)) <= 1e-8 * np.linalg.norm(ys)
    assert np.linalg.norm(fs.fields.u_.vector().get_local() - u_n) <= 1e-8 * np.linalg.norm(u_n)
    assert 1 <= max(its) <= 100
    print(f"[factorisation-free, Crank-Nicolson] iterations per step {its}")
    fs.th.release_device()
