"""CPU tests of the host side: mesh/XDMF/HDF5 reading, Taylor–Hood tables, dolfin boundary
semantics, sensors, actuators, controller, exporter, parameter validation, ND factorisation.
Modelled on the reference's unit tests (tests/test_*.py) — same identities, our own objects."""
import json
import tempfile
from pathlib import Path

import numpy as np
import pytest
import scipy.signal

from tests.support import ndsolver
from flowcontrol_amd.actuator import ActuatorBCParabolicV, ActuatorBCRotation
from flowcontrol_amd.controller import Controller
from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver
from flowcontrol_amd.fem import element as el
from flowcontrol_amd.fem.boundary import DOLFIN_EPS, Constant, DirichletBC, SubDomain, between, combine_bcs, near
from flowcontrol_amd.fem.mesh import Mesh, read_xdmf_mesh
from flowcontrol_amd.fem.spaces import Function, TaylorHood
from flowcontrol_amd.flowsolverparameters import ParamIC, ParamTime
from flowcontrol_amd.sensor import SENSOR_TYPE, SensorHorizontalWallShear, SensorPoint
from oracle import ns_oracle as O
from tests.support import nd_numeric
from flowcontrol_amd.examples.data import mesh_file, controller_file  # noqa: E402

REF = Path("/root/reference/src/examples")


# ── mesh / readers ───────────────────────────────────────────────────────────────────────────
@pytest.mark.skipif(not REF.exists(), reason="reference data files are only mounted in the build container")
@pytest.mark.parametrize("rel,name", [("cylinder/data_input/O1.xdmf", "O1"), ("cavity/data_input/cavity_fine.xdmf", "cavity_fine"),
                                      ("pinball/data_input/mesh_middle_gmsh.xdmf", "mesh_middle_gmsh")])
def test_xdmf_hdf5_reader_matches_fixture(rel, name, golden_dir):
    """Chunked+deflate, int64/int32 topology and big-endian f8 geometry (cavity_fine) all decode."""
    a = read_xdmf_mesh(REF / rel)
    b = read_xdmf_mesh(mesh_file(name))
    assert np.array_equal(a.coords, b.coords) and np.array_equal(a.cells, b.cells)


@pytest.mark.skipif(not REF.exists(), reason="reference data files are only mounted in the build container")
def test_dolfin_layout_hdf5_contiguous():
    m = read_xdmf_mesh(REF / "lidcavity/data_input/mesh64.xdmf")
    assert m.num_cells > 0 and np.isclose(m.coords.min(), 0.0) and np.isclose(m.coords.max(), 1.0)


def test_o1_sizes_match_survey(golden_dir):
    m = read_xdmf_mesh(mesh_file("O1"))
    th = TaylorHood(m)
    assert (m.num_vertices, m.num_cells, m.num_edges, len(m.boundary_edges())) == (6327, 12284, 18611, 370)
    assert (2 * th.nn, th.nv, th.N) == (49876, 6327, 56203)
    assert np.all(th.detJ > 0)
    assert np.isclose(0.5 * th.detJ.sum(), 30 * 20 - np.pi * 0.25, rtol=1e-3)  # domain area minus the cylinder


def test_numbering_is_orientation_and_order_independent():
    m1 = Mesh.unit_square(5, 4)
    perm = np.random.default_rng(0).permutation(m1.num_cells)
    flipped = m1.cells[perm][:, [0, 2, 1]]
    m2 = Mesh.from_arrays(m1.coords, flipped)
    a = np.sort(np.sort(m1.coords[m1.cells].reshape(-1, 6), axis=1), axis=0)
    b = np.sort(np.sort(m2.coords[m2.cells].reshape(-1, 6), axis=1), axis=0)
    assert np.allclose(a, b)


def test_refine_quadruples_cells():
    m = Mesh.unit_square(3, 3)
    r = m.refine()
    assert r.num_cells == 4 * m.num_cells and r.num_vertices == m.num_vertices + m.num_edges
    assert np.isclose(TaylorHood(r).detJ.sum(), TaylorHood(m).detJ.sum())


# ── element tables ───────────────────────────────────────────────────────────────────────────
def test_quadrature_is_exact_to_degree_5():
    # ∫_T λ0^a λ1^b λ2^c = a! b! c! 2! / (a+b+c+2)!  (area 1/2 → weights sum to 1 times 1/2)
    from math import factorial as f

    for a in range(6):
        for b in range(6 - a):
            for c in range(6 - a - b):
                exact = f(a) * f(b) * f(c) * 2.0 / f(a + b + c + 2)
                got = (el.QUAD_W * el.QUAD_BARY[:, 0] ** a * el.QUAD_BARY[:, 1] ** b * el.QUAD_BARY[:, 2] ** c).sum()
                assert np.isclose(got, exact, rtol=1e-13, atol=1e-16)


def test_p2_basis_is_nodal_and_partition_of_unity():
    assert np.allclose(el.p2_basis(el.P2_NODES_BARY), np.eye(6))
    lam = np.random.default_rng(1).dirichlet(np.ones(3), 10)
    assert np.allclose(el.p2_basis(lam).sum(axis=1), 1.0)
    assert np.allclose(el.p2_grad_ref(lam).sum(axis=1), 0.0)


# ── boundary semantics ───────────────────────────────────────────────────────────────────────
def test_cylinder_boundary_counts_match_survey():
    """inlet 20 facets/41 nodes, outlet 20/41, walls 60/122, cylinder 252/506, slots 9/19."""
    fs = CylinderFlowSolver.make_default(path_out=tempfile.mkdtemp())
    want = {"inlet": (20, 41), "outlet": (20, 41), "walls": (60, 122), "cylinder": (252, 506), "actuator_up": (9, 19), "actuator_lo": (9, 19)}
    for name, (nf, nn) in want.items():
        bc = DirichletBC(fs.W.sub(0), Constant((0, 0)), fs.get_subdomain(name))
        assert (len(bc.facets), len(bc.nodes)) == (nf, nn), name
    dofs, prof = fs._bc_tables()
    assert len(dofs) == 1282 and prof.shape == (1282, 2)
    # later BCs overwrite earlier ones on shared dofs: the slot end points carry the actuator profile (0 there)
    assert np.all(prof[:, 0] * prof[:, 1] == 0)


def test_facet_needs_vertices_and_midpoint_inside():
    m = Mesh.unit_square(4, 4)
    left_half = SubDomain(lambda x, ob: ob & near(x[:, 0], 0.0, DOLFIN_EPS) & (x[:, 1] <= 0.3))
    marked = np.nonzero(left_half.mark_facets(m))[0]
    assert len(marked) == 1  # only [0, 0.25]; the facet [0.25, 0.5] has a vertex outside
    assert between(np.array([0.5]), 0.5, 1.0)[0] and not between(np.array([0.5 - 1e-3]), 0.5, 1.0)[0]


def test_subspace_bc_constrains_one_component():
    th = TaylorHood(Mesh.unit_square(3, 3))
    top = SubDomain(lambda x, ob: ob & near(x[:, 1], 1.0, DOLFIN_EPS))
    bc_v = DirichletBC(th.W.sub(0).sub(1), Constant(0), top)
    bc_uv = DirichletBC(th.W.sub(0), Constant((1.0, 2.0)), top)
    assert np.all((bc_v.dofs >= th.nn) & (bc_v.dofs < 2 * th.nn)) and len(bc_uv.dofs) == 2 * len(bc_v.dofs)
    dofs, vals = combine_bcs([bc_uv, bc_v], th.N)
    assert set(vals) == {0.0, 1.0}  # v overwritten by the later BC, u kept


# ── sensors / actuators ──────────────────────────────────────────────────────────────────────
def test_point_sensor_row_equals_eval():
    fs = CylinderFlowSolver.make_default(path_out=tempfile.mkdtemp())
    up = Function(fs.W, np.random.default_rng(0).standard_normal(fs.th.N))
    for s in fs.params_control.sensor_list:
        idx, w = s.row(fs)
        assert np.isclose(w @ up.vector().array()[idx], s.eval(up), rtol=1e-12)
    lin = Function(fs.W)
    lin.interpolate(lambda x: np.stack([2 * x[:, 0], 3 * x[:, 1] - x[:, 0], x[:, 0] + x[:, 1]], axis=1))
    assert np.allclose(lin((3.1, 1.0)), [6.2, -0.1, 4.1])
    with pytest.raises(RuntimeError):
        lin((100.0, 0.0))


def test_wall_shear_sensor_is_exact_for_linear_shear():
    class Dummy:
        pass

    th = TaylorHood(Mesh.unit_square(8, 8))
    fs = Dummy()
    fs.th = th
    s = SensorHorizontalWallShear(sensor_type=SENSOR_TYPE.OTHER, x_sensor_left=0.25, x_sensor_right=0.75, y_sensor=0.0)
    s.load(fs)
    up = Function(th.W)
    up.interpolate(lambda x: np.stack([3.0 * x[:, 1] + x[:, 1] ** 2, 0 * x[:, 0], 0 * x[:, 0]], axis=1))
    assert len(s.ds) == 4  # whole facets inside [0.25, 0.75]
    assert np.isclose(s.eval(up), 3.0 * 0.5, rtol=1e-12)  # ∂u_x/∂y = 3 + 2y = 3 on y = 0


def test_actuator_profiles():
    a = ActuatorBCParabolicV(width=0.2, position_x=1.0)
    e = a._load_expression(None, None)
    e.u_ctrl = 2.0
    x = np.array([[1.0, 0.0], [1.1, 0.0], [1.2, 0.0], [0.7, 0.0]])
    assert np.allclose(e(x), [[0, 2.0], [0, 1.5], [0, 0], [0, 0]])
    assert np.isclose(ActuatorBCParabolicV.angular_size_deg_to_width(10, 0.5), 0.5 * np.sin(np.deg2rad(5)))
    r = ActuatorBCRotation(diameter=1.0)._load_expression(None, None)
    r.u_ctrl = 1.0
    assert np.allclose(r(np.array([[0.5, 0.0], [0.0, 0.5]])), [[0, 0.5], [-0.5, 0]])


# ── controller ───────────────────────────────────────────────────────────────────────────────
def test_controller_zoh_matches_scipy_and_forced_response():
    rng = np.random.default_rng(0)
    A = rng.standard_normal((4, 4)) - 2 * np.eye(4)
    B, C, D = rng.standard_normal((4, 2)), rng.standard_normal((3, 4)), rng.standard_normal((3, 2))
    K = Controller(A, B, C, D)
    dt = 0.01
    Ad, Bd, Cd, Dd, _ = scipy.signal.cont2discrete((A, B, C, D), dt, method="zoh")
    x = np.zeros(4)
    for k in range(20):
        y = np.array([np.sin(0.3 * k), np.cos(0.2 * k)])
        u = K.step(y, dt)
        assert np.allclose(u, Cd @ x + Dd @ y)
        x = Ad @ x + Bd @ y
    assert np.allclose(K.x, x)
    K.reset()
    assert np.all(K.x == 0)


def test_controller_from_file_and_algebra(golden_dir):
    K = Controller.from_file(controller_file())
    assert (K.nstates, K.ninputs, K.noutputs) == (13, 1, 1) and K.file is not None
    assert np.max(np.linalg.eigvals(K.A).real) < 0
    G = Controller(-np.eye(2), np.ones((2, 1)), np.ones((1, 2)), np.array([[2.0]]))
    s = 0.3j
    tf = lambda S: S.C @ np.linalg.solve(s * np.eye(S.nstates) - S.A, S.B) + S.D  # noqa: E731
    assert np.allclose(tf(G + G), 2 * tf(G)) and np.allclose(tf(G * G), tf(G) @ tf(G))
    assert np.allclose(tf(G.inv()), np.linalg.inv(tf(G)))
    assert (G * G).x.shape == (4,)


# ── exporter / parameters ────────────────────────────────────────────────────────────────────
def test_exporter_schema_and_sidecar(tmp_path):
    fs = CylinderFlowSolver.make_default(path_out=tmp_path, save_every=5, num_steps=10)
    ex = fs.exporter
    ex.log_ic(t=0.0, y_meas=[1.0, 2.0, 3.0], dE=0.5)
    ex.log(u_ctrl=[0.1, 0.2], y_meas=[4.0, 5.0, 6.0], dE=0.6, t=0.005, runtime=1e-3)
    df = ex.to_dataframe()
    assert list(df.columns) == ["time", "dE", "runtime", "y_meas_1", "y_meas_2", "y_meas_3", "u_ctrl_1", "u_ctrl_2"]
    assert np.isnan(df.loc[0, "u_ctrl_1"]) and df.loc[1, "u_ctrl_2"] == 0.2
    ex.write_timeseries()
    assert fs.paths.timeseries.name == "timeseries1D_restart0,000.csv" and fs.paths.timeseries.exists()
    ex.write_metadata(restart_order=2)
    meta = json.loads(fs.paths.metadata.read_text())
    assert set(meta) == {"Tstart", "dt", "save_every", "checkpoints_written", "restart_order", "files"}
    assert meta["files"]["U"] == "U_restart0,000.xdmf"


def test_parameter_validation(tmp_path):
    with pytest.raises(ValueError):
        fs = CylinderFlowSolver.make_default(path_out=tmp_path)
        fs._validate_params(fs.params_flow, ParamTime(num_steps=1, dt=-1.0, Tstart=0.0), fs.params_save, fs.params_solver, fs.params_mesh, fs.params_control, fs.params_ic)
    with pytest.raises(FileNotFoundError):
        CylinderFlowSolver.make_default(path_out=tmp_path, meshpath=tmp_path / "nope.xdmf")
    assert ParamTime(num_steps=10, dt=0.5, Tstart=1.0).Tfinal == 5.0
    assert ParamIC().amplitude == 1.0


def test_default_ic_is_divergence_free():
    """‖P1-projection of div u‖_L2 < 1e-2 for the interpolated Gaussian vortex on 32×32
    (the reference's tests/test_physics.py:31-46, same mesh, centre and size)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    th = TaylorHood(Mesh.unit_square(32, 32))
    d = O.Disc.from_taylor_hood(th)
    u = O.div0_gaussian_nodal(th.node_coords, 0.5, 0.5, 0.1)
    assert np.abs(u).max() > 0.1
    _, grad = d.vel_at_q(np.r_[u[:, 0], u[:, 1]])
    div = grad[:, :, 0, 0] + grad[:, :, 1, 1]
    P1 = el.PHI1
    Me = np.einsum("cq,qm,qn->cmn", d.w, P1, P1)
    rows = np.repeat(d.cells, 3, axis=1).reshape(-1)
    cols = np.tile(d.cells, (1, 3)).reshape(-1)
    Mp = sp.coo_matrix((Me.reshape(-1), (rows, cols)), shape=(th.nv, th.nv)).tocsc()
    rhs = np.zeros(th.nv)
    np.add.at(rhs, d.cells.reshape(-1), np.einsum("cq,qm,cq->cm", d.w, P1, div).reshape(-1))
    proj = spla.spsolve(Mp, rhs)
    assert np.sqrt(proj @ (Mp @ proj)) < 1e-2


# ── nested-dissection selected inverse (host numerics of the device solver) ─────────────────
@pytest.mark.parametrize("depth,merge", [(3, 1), (4, 2), (6, 3)])
def test_nd_block_factors_solve_saddle_point_system(depth, merge):
    th = TaylorHood(Mesh.unit_square(10, 10))
    d = O.Disc.from_taylor_hood(th)
    x = th.node_coords
    U = np.r_[1 + 0.3 * np.sin(x[:, 0]), 0.2 * np.cos(x[:, 1])]
    m = th.mesh
    be = m.boundary_edges()
    be = be[m.edge_midpoints()[be, 0] < 1 - 1e-9]
    nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
    dofs = np.sort(np.r_[nodes, nodes + th.nn])
    A, _ = O.apply_bc_symmetric(O.assemble_matrix(d, mass=300.0, nu=0.01, adv=U, lin=U), None, dofs, np.zeros(len(dofs)))
    skip = np.zeros(th.N, bool)
    skip[dofs] = True
    tree = ndsolver.build_tree(th.cell_dofs, m.cell_centroids(), th.N, depth, skip, merge=merge)
    assert sorted(tree.perm) == list(range(th.N))
    b = np.random.default_rng(0).standard_normal(th.N)
    fb = nd_numeric.factorize_blocks(A, tree)
    xb = nd_numeric.block_solve(fb, b)
    assert np.linalg.norm(A @ xb - b) / np.linalg.norm(b) < 1e-12
    fc = nd_numeric.factorize(A, tree)
    assert fc.nnz == fb.nnz and np.allclose(fc.solve(b), xb, rtol=1e-12, atol=1e-14)
    # every stage only reads what earlier stages (or the other half of the buffer) produced
    for s in range(len(fb.stage_kind)):
        r0, nr = int(fb.stage_row0[s]), int(fb.stage_nrows[s])
        q0, q1 = int(fb.seg_ptr[fb.stage_begin[s]]), int(fb.seg_ptr[fb.stage_begin[s] + nr])
        cols = fb.seg_col[q0:q1]
        if fb.stage_kind[s] == 0:
            assert np.all(cols >= 0) and np.all(cols + fb.seg_len[q0:q1] <= r0)  # deeper levels come first


# ── dolfin look-alike: the reference's C++ predicate / expression strings evaluated with numpy ──
def test_compiled_subdomains_reproduce_the_case_file_boundaries():
    """The six cylinder boundaries built from the *strings* of the reference's case file
    (cylinderflowsolver.py:35-83, via the utils/fem.py string helpers) mark exactly the facets that the
    hand-written numpy predicates of flowcontrol_amd's case file mark."""
    from flowcontrol_amd import dolfin_compat as dolfin

    fs = CylinderFlowSolver.make_default(path_out=tempfile.mkdtemp())
    near_cpp, between_cpp, and_cpp, or_cpp, ob = dolfin.near_cpp, dolfin.between_cpp, dolfin.and_cpp(), dolfin.or_cpp(), dolfin.on_boundary_cpp()
    TOL = dolfin.DOLFIN_EPS
    radius, ldelta = 0.5, fs.params_control.actuator_list[0].width
    close = between_cpp("x[0]", "-radius", "radius") + and_cpp + between_cpp("x[1]", "-radius", "radius")
    cyl_b = ob + and_cpp + close
    cone_up = between_cpp("x[0]", "-ldelta", "ldelta", tol="0.01") + and_cpp + between_cpp("x[1]", "0", "radius")
    cone_lo = between_cpp("x[0]", "-ldelta", "ldelta", tol="0.01") + and_cpp + between_cpp("x[1]", "-radius", "0")
    subs = {
        "inlet": dolfin.CompiledSubDomain(ob + and_cpp + near_cpp("x[0]", "xinfa", "MESH_TOL"), xinfa=-10, MESH_TOL=TOL),
        "outlet": dolfin.CompiledSubDomain(ob + and_cpp + near_cpp("x[0]", "xinf", "MESH_TOL"), xinf=20, MESH_TOL=TOL),
        "walls": dolfin.CompiledSubDomain(ob + and_cpp + "(" + near_cpp("x[1]", "-yinf", "MESH_TOL") + or_cpp + near_cpp("x[1]", "yinf", "MESH_TOL") + ")", yinf=10, MESH_TOL=TOL),
        "cylinder": dolfin.CompiledSubDomain(cyl_b + and_cpp + "(" + between_cpp("x[0]", "-radius", "-ldelta") + or_cpp + between_cpp("x[0]", "ldelta", "radius") + ")", radius=radius, ldelta=ldelta),
        "actuator_up": dolfin.CompiledSubDomain(cyl_b + and_cpp + cone_up, radius=radius, ldelta=ldelta),
        "actuator_lo": dolfin.CompiledSubDomain(cyl_b + and_cpp + cone_lo, radius=radius, ldelta=ldelta),
    }
    for name, sd in subs.items():
        assert np.array_equal(sd.mark_facets(fs.mesh), fs.get_subdomain(name).mark_facets(fs.mesh)), name


def test_c_expression_strings_of_the_actuators():
    from flowcontrol_amd import dolfin_compat as dolfin

    # reference actuator.py:190-199 (parabolic slot) and :241-251 (rotation)
    e = dolfin.Expression(["0", "(x[0]-x0>=L || x[0]-x0<=-L) ? 0 : u_ctrl * -1*(x[0]-x0+L)*(x[0]-x0-L) / (L*L)"], element=None, L=0.2, x0=1.0, u_ctrl=0.0)
    a = ActuatorBCParabolicV(width=0.2, position_x=1.0)._load_expression(None, None)
    x = np.random.default_rng(0).uniform(0.5, 1.5, (50, 2))
    e.u_ctrl = a.u_ctrl = 1.7
    assert np.allclose(e(x), a(x))
    r = dolfin.Expression(["-sin(atan2(x[1]-y0,x[0]-x0))*u_ctrl*d/2", "cos(atan2(x[1]-y0,x[0]-x0))*u_ctrl*d/2"], element=None, y0=0.1, x0=-0.2, u_ctrl=0.0, d=1.0)
    b = ActuatorBCRotation(position_x=-0.2, position_y=0.1, diameter=1.0)._load_expression(None, None)
    r.u_ctrl = b.u_ctrl = -0.8
    assert np.allclose(r(x), b(x))
    f = dolfin.compile_c_expression("a > 1 && !(b <= 2) || -c * 2 + 1 == 3 ? pow(a, 2) : exp(0) / 4")
    assert np.allclose(f({"a": np.array([2.0, 0.0, 0.0]), "b": np.array([3.0, 0.0, 0.0]), "c": np.array([0.0, -1.0, 0.0])}), [4.0, 0.0, 0.25])
    with pytest.raises(ValueError):
        dolfin.compile_c_expression("foo(1)")({})


# ── HDF5 writer + XDMF checkpoints ───────────────────────────────────────────────────────────
def test_hdf5_writer_round_trip_and_xdmf_checkpoints(tmp_path):
    from flowcontrol_amd.fem.hdf5_min import MinimalHDF5, read_hdf5_tree, write_hdf5
    from flowcontrol_amd.io import read_xdmf, write_xdmf

    tree = {"Mesh": {"mesh": {"geometry": np.random.rand(7, 2), "topology": np.arange(12, dtype=np.int64).reshape(4, 3)}},
            "many": {str(i): np.full(3, float(i)) for i in range(40)}, "flags": np.array([1, 0, 1], dtype=np.uint8),
            "i4": np.arange(5, dtype=np.int32)}
    write_hdf5(tmp_path / "t.h5", tree)
    back = read_hdf5_tree(tmp_path / "t.h5")
    assert np.array_equal(back["Mesh"]["mesh"]["geometry"], tree["Mesh"]["mesh"]["geometry"])
    assert back["Mesh"]["mesh"]["topology"].dtype == np.int64 and back["i4"].dtype == np.int32 and back["flags"].dtype == np.uint8
    assert len(back["many"]) == 40 and back["many"]["39"][0] == 39.0
    assert MinimalHDF5(tmp_path / "t.h5").keys("/Mesh/mesh") == ["geometry", "topology"]

    th = TaylorHood(Mesh.unit_square(4, 4))
    u = Function(th.V, np.random.default_rng(0).standard_normal(2 * th.nn))
    assert write_xdmf(tmp_path / "U_restart0,000.xdmf", u, "U", 0.0, append=False) == 0
    u2 = Function(th.V, 2.0 * u.vector().array())
    assert write_xdmf(tmp_path / "U_restart0,000.xdmf", u2, "U", 0.025, append=True) == 1
    v = Function(th.V)
    assert read_xdmf(tmp_path / "U_restart0,000.xdmf", v, "U", 1) == 0.025 and np.array_equal(v.vector().array(), u2.vector().array())
    assert read_xdmf(tmp_path / "U_restart0,000.xdmf", v, "U", 0) == 0.0 and np.array_equal(v.vector().array(), u.vector().array())
    with pytest.raises(FileNotFoundError):
        read_xdmf(tmp_path / "U_restart0,000.xdmf", v, "U", 5)
    with pytest.raises(ValueError):
        read_xdmf(tmp_path / "U_restart0,000.xdmf", Function(th.P), "U", 0)
    # the XDMF is a valid temporal collection whose mesh our own reader (and ParaView) can open
    m = read_xdmf_mesh(tmp_path / "U_restart0,000.xdmf")
    assert m.num_cells == th.nc and m.num_vertices == th.nv


def _emulated_dolfin_read_checkpoint(xdmf, name, counter, coords_o, cells_o):
    """What ``dolfin.XDMFFile.read_checkpoint`` does (HDF5Utility::set_local_vector_values), on a P2-vector space built
    here from the ORIGINAL mesh arrays with dolfin's conventions (cells with ascending vertex ids, local dofs
    v0 v1 v2 e0 e1 e2 per component, edge i opposite vertex i) and a global numbering of its own (interleaved
    components, edges numbered by np.unique) — nothing of flowcontrol_amd's numbering is used.
    Returns (x, dof coordinates, dof component)."""
    import xml.etree.ElementTree as ET

    from flowcontrol_amd.fem.hdf5_min import MinimalHDF5

    cells_o = np.sort(np.asarray(cells_o, dtype=np.int64), axis=1)
    nv = coords_o.shape[0]
    pairs = np.stack([cells_o[:, [1, 2]], cells_o[:, [0, 2]], cells_o[:, [0, 1]]], axis=1)  # local edge i: opposite vertex i
    uniq, inv = np.unique(pairs.reshape(-1, 2), axis=0, return_inverse=True)
    node = np.hstack([cells_o, nv + inv.reshape(-1, 3)])  # (nc, 6) scalar P2 nodes, reader's numbering
    xy = np.vstack([coords_o, 0.5 * (coords_o[uniq[:, 0]] + coords_o[uniq[:, 1]])])
    dofmap = np.hstack([2 * node, 2 * node + 1])  # (nc, 12)
    grid = [g for g in ET.parse(xdmf).getroot().iter("Grid") if g.get("Name") == f"{name}_{counter}"][0]
    att = [a for a in grid.findall("Attribute") if a.get("ItemType") == "FiniteElementFunction"][0]
    assert (att.get("ElementFamily"), att.get("ElementDegree"), att.get("ElementCell")) == ("CG", "2", "triangle")
    data = []
    for d in att.findall("DataItem"):
        f, _, path = d.text.strip().partition(":")
        data.append(np.asarray(MinimalHDF5(Path(xdmf).parent / f).read(path)).reshape(-1))
    cell_dofs, vector, x_cell_dofs, cells = data
    x = np.full(2 * xy.shape[0], np.nan)
    for r, cell in enumerate(cells):
        for j in range(12):
            x[dofmap[cell, j]] = vector[cell_dofs[x_cell_dofs[r] + j]]
    return x, np.repeat(xy, 2, axis=0), np.tile([0, 1], xy.shape[0])


@pytest.mark.parametrize("mesh", ["square", "O1"])
def test_checkpoints_follow_dolfins_write_checkpoint_layout(mesh, tmp_path, golden_dir):
    """A velocity field written by write_xdmf, read back the way dolfin's read_checkpoint reads it — into a space with a
    different dof numbering built from the original mesh arrays — lands on the right nodes; files with another dof
    numbering and cell order (what dolfin itself would write) are read correctly by read_xdmf."""
    from flowcontrol_amd.fem.hdf5_min import read_hdf5_tree, write_hdf5
    from flowcontrol_amd.io import read_xdmf, write_xdmf

    if mesh == "square":
        base = Mesh.unit_square(5, 4, reorder=False)
        coords_o, cells_o = base.coords.copy(), base.cells.copy()
    else:
        z = np.load(mesh_file("O1"))
        coords_o, cells_o = z["coords"], z["cells"]
    th = TaylorHood(Mesh.from_arrays(coords_o, cells_o, reorder=True))  # Morton-ordered cells, renumbered vertices
    f = lambda xy, c: np.sin(1.3 * xy[:, 0] + 0.2 * c) * np.cos(0.7 * xy[:, 1]) + 0.1 * c * xy[:, 0]  # noqa: E731
    u = Function(th.V, np.r_[f(th.node_coords, 0), f(th.node_coords, 1)])
    path = tmp_path / "U.xdmf"
    write_xdmf(path, Function(th.V), "U", 0.0)
    write_xdmf(path, u, "U", 0.5, append=True)
    x, xy, comp = _emulated_dolfin_read_checkpoint(path, "U", 1, coords_o, cells_o)
    assert not np.isnan(x).any()
    assert np.array_equal(x[comp == 0], f(xy[comp == 0], 0)) and np.array_equal(x[comp == 1], f(xy[comp == 1], 1))
    # the other direction: rewrite frame 1 with a shuffled dof numbering and reversed cell rows, as a foreign writer may
    rng = np.random.default_rng(1)
    tree0 = read_hdf5_tree(tmp_path / "U.h5")["U"]["U_0"]
    tree1 = read_hdf5_tree(tmp_path / "U.1.h5")["U"]["U_1"]
    perm = rng.permutation(2 * th.nn)  # new dof id of old dof i
    vec = np.empty(2 * th.nn)
    vec[perm] = tree1["vector"].reshape(-1)
    cd = perm[tree0["cell_dofs"].reshape(th.nc, 12)][::-1]
    tree0.update(cell_dofs=cd.reshape(-1, 1), cells=tree0["cells"][::-1].copy())
    tree1.update(vector=vec.reshape(-1, 1))
    write_hdf5(tmp_path / "U.h5", {"U": {"U_0": tree0}})
    write_hdf5(tmp_path / "U.1.h5", {"U": {"U_1": tree1}})
    v = Function(th.V)
    assert read_xdmf(path, v, "U", 1) == 0.5
    assert np.array_equal(v.vector().array(), u.vector().array())


def test_factor_plan_replays_the_multifrontal_factorisation():
    """The symbolic plan handed to the device (fc_factor_plan) + its host replay reproduce the numpy
    multifrontal factors bit for bit, and the structure-only layout equals the numeric one."""
    import scipy.sparse as sp

    from tests.support import ndsolver as nd
    from flowcontrol_amd.fem.mesh import Mesh
    from flowcontrol_amd.fem.spaces import TaylorHood
    from oracle import ns_oracle as O

    th = TaylorHood(Mesh.unit_square(10, 10))
    d = O.Disc.from_taylor_hood(th)
    U = np.r_[np.ones(th.nn), 0.3 * np.ones(th.nn)]
    A = O.assemble_matrix(d, mass=100.0, nu=0.01, adv=U, lin=U).tocsr()
    A.sort_indices()
    indptr, indices = A.indptr.copy(), A.indices.copy()
    m = th.mesh
    be = m.boundary_edges()
    nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
    nodes = nodes[th.node_coords[nodes, 0] < 1 - 1e-9]
    dofs = np.sort(np.r_[nodes, nodes + th.nn])
    Abc, _ = O.apply_bc_symmetric(A, None, dofs, np.zeros(dofs.size))
    vals = np.asarray(Abc.tocsr()[np.repeat(np.arange(th.N), np.diff(indptr)), indices]).ravel()  # values on the FULL pattern
    skip = np.zeros(th.N, bool)
    skip[dofs] = True
    for top_bits in (0, 1):
        t = nd.build_tree(th.cell_dofs, m.cell_centroids(), th.N, 4, skip, merge=2, top_bits=top_bits)
        Afull = sp.csr_matrix((vals, indices, indptr), shape=(th.N, th.N))
        f0 = nd.factorize_blocks(None, t)
        assert not f0.vals.any()
        plan = nd.factor_plan(f0, indptr, indices, skip)
        assert plan.nodes[:, 5].max() < plan.nodes.shape[0] and plan.a_ptr[-1] == plan.a_src.size
        assert np.unique(plan.a_dst).size == plan.a_dst.size  # one front slot per matrix entry
        # the plan replayed on the host (what fc_refactor does on the device) against the independent CSR multifrontal
        f0.vals = nd_numeric.factorize_with_plan(plan, f0, vals)
        b = np.random.default_rng(top_bits).standard_normal(th.N)
        x_plan = nd_numeric.block_solve(f0, b)
        x_csr = nd_numeric.factorize(Afull, t).solve(b)
        assert np.linalg.norm(x_plan - x_csr) <= 1e-12 * np.linalg.norm(x_csr)
        assert np.linalg.norm(Afull @ x_plan - b) <= 1e-12 * np.linalg.norm(b)
        f1 = nd_numeric.factorize_blocks(Afull, t)  # the convenience route used by the GPU parity tests: same numbers
        assert np.array_equal(f1.vals, f0.vals)


def test_pressure_pin_only_for_enclosed_flows():
    """fem.boundary.pressure_pin: a pressure dof is pinned exactly when the velocity is Dirichlet on the
    whole boundary (lid-driven cavity); any open piece of boundary (cylinder outlet) → None.  The Dirichlet
    form used on the host (row replaced) and the diagonal shift used in the device factors select the same
    solution of the singular system."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    from flowcontrol_amd.fem.boundary import pressure_pin, with_pressure_pin
    from flowcontrol_amd.fem.mesh import Mesh
    from flowcontrol_amd.fem.spaces import TaylorHood
    from oracle import ns_oracle as O

    th = TaylorHood(Mesh.unit_square(6, 6))
    m = th.mesh
    be = m.boundary_edges()
    nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
    closed = np.sort(np.r_[nodes, nodes + th.nn])
    opened = closed[th.node_coords[closed % th.nn, 0] < 1 - 1e-9]
    assert pressure_pin(th, opened) is None
    pin = pressure_pin(th, closed)
    assert pin is not None and 2 * th.nn <= pin < th.N
    assert np.allclose(th.node_coords[pin - 2 * th.nn], [0.0, 0.0])
    d2, v2 = with_pressure_pin(th, closed, np.zeros((closed.size, 2)))
    assert d2.size == closed.size + 1 and pin in d2 and v2.shape == (closed.size + 1, 2) and np.all(np.diff(d2) > 0)
    # the two treatments of the singular enclosed system give the same solution
    d = O.Disc.from_taylor_hood(th)
    A = O.assemble_matrix(d, mass=50.0, nu=0.02).tocsr()
    rng = np.random.default_rng(0)
    b = rng.standard_normal(th.N)
    b[2 * th.nn :] = 0.0  # compatible: no net source in the continuity rows
    A1, b1 = O.apply_bc_symmetric(A, b.copy(), d2, np.zeros(d2.size))
    x1 = spla.spsolve(A1.tocsc(), b1)
    A2, b2 = O.apply_bc_symmetric(A, b.copy(), closed, np.zeros(closed.size))
    A2 = (A2 + sp.csr_matrix(([1.0], ([pin], [pin])), shape=A2.shape)).tocsc()
    x2 = spla.spsolve(A2, b2)
    assert abs(x2[pin]) < 1e-12
    assert np.linalg.norm(x1 - x2) < 1e-10 * np.linalg.norm(x1)


def test_checkpoint_series_roundtrip(tmp_path):
    """io.write_xdmf / read_xdmf: a series is an .xdmf index + <stem>.h5 (frame 0, mesh, cell tables: dolfin's file
    name) + one small HDF5 file per later frame; appending does not touch earlier frames; frames come back bit-exact
    by counter (−1 = last)."""
    from flowcontrol_amd.fem.mesh import Mesh
    from flowcontrol_amd.fem.spaces import Function, FunctionSpace, TaylorHood
    from flowcontrol_amd.io import read_xdmf, write_xdmf

    th = TaylorHood(Mesh.unit_square(5, 4))
    V, P = FunctionSpace(th, "V"), FunctionSpace(th, "P")
    rng = np.random.default_rng(3)
    frames = [rng.standard_normal(2 * th.nn) for _ in range(4)]
    path = tmp_path / "U_restart0,000.xdmf"
    for k, v in enumerate(frames):
        assert write_xdmf(path, Function(V, v), "U", time_step=0.25 * k, append=k > 0) == k
    first = (tmp_path / "U_restart0,000.h5").read_bytes()
    assert sorted(p.name for p in tmp_path.iterdir()) == sorted(
        ["U_restart0,000.xdmf", "U_restart0,000.h5"] + [f"U_restart0,000.{k}.h5" for k in range(1, 4)])
    f = Function(V)
    for k, v in enumerate(frames):
        assert read_xdmf(path, f, "U", counter=k) == 0.25 * k
        assert np.array_equal(f.vector().get_local(), v)
    assert read_xdmf(path, f, "U") == 0.75 and np.array_equal(f.vector().get_local(), frames[-1])
    assert (tmp_path / "U_restart0,000.h5").read_bytes() == first  # untouched by the appends
    xml = path.read_text()
    assert xml.count("<Time Value=") == 4 and "U_restart0,000.3.h5:/U/U_3/vector" in xml and "U_restart0,000.h5:/U/U_0/mesh/topology" in xml
    assert xml.count('ItemType="FiniteElementFunction"') == 4 and "U_restart0,000.h5:/U/U_0/cell_dofs" in xml
    # pressure and mixed fields take the same road
    pq = Function(P, rng.standard_normal(th.nv))
    write_xdmf(tmp_path / "P0.xdmf", pq, "P0")
    back = Function(P)
    assert read_xdmf(tmp_path / "P0.xdmf", back, "P0") == 0.0 and np.array_equal(back.vector().get_local(), pq.vector().get_local())
    with pytest.raises(FileNotFoundError):
        read_xdmf(path, f, "U", counter=4)
    with pytest.raises(ValueError):
        read_xdmf(path, Function(P), "U", counter=0)
    # starting over (append=False) resets the series
    assert write_xdmf(path, Function(V, frames[1]), "U", time_step=9.0, append=False) == 0
    assert read_xdmf(path, f, "U") == 9.0


def test_boundary_force_of_a_linear_flow_on_the_unit_square():
    """F = ∫ −σ·n ds with σ = ν(∇u + ∇uᵀ) − p I (reference physics.py:17-19, pinballflowsolver.py:202-232): for a
    linear velocity and a linear pressure the facet integrals are known in closed form; over the closed boundary
    the constant part of σ integrates to zero and the pressure gradient leaves −∇p · |Ω|."""
    from flowcontrol_amd.fem.forces import boundary_force
    from flowcontrol_amd.fem.mesh import Mesh
    from flowcontrol_amd.fem.spaces import TaylorHood

    th = TaylorHood(Mesh.unit_square(6, 6))
    x = th.node_coords
    a, b, c, d = 0.3, -1.1, 0.7, 0.45
    u = np.r_[a * x[:, 0] + b * x[:, 1], c * x[:, 0] + d * x[:, 1]]
    xv = th.mesh.coords
    p = 2.0 + 0.5 * xv[:, 0] - 0.25 * xv[:, 1]
    nu = 0.02
    m = th.mesh
    be = m.boundary_edges()
    mid = m.edge_midpoints()[be]
    right = be[np.isclose(mid[:, 0], 1.0)]
    G = np.array([[a, b], [c, d]])
    S = nu * (G + G.T)
    # right edge: n = (1, 0); ∫ p ds over x = 1, y in [0, 1] = 2 + 0.5 − 0.125
    Fx, Fy = boundary_force(th, right, nu, u, p)
    assert np.isclose(Fx, -(S[0, 0] - 2.375)) and np.isclose(Fy, -S[1, 0])
    Fx, Fy = boundary_force(th, be, nu, u, p)  # closed boundary: ∮ p n ds = ∇p |Ω|
    assert np.isclose(Fx, 0.5, atol=1e-12) and np.isclose(Fy, -0.25, atol=1e-12)


def test_export_subdomains_writes_facet_markers(tmp_path):
    """flu.export_subdomains (reference utils/io.py:171-185): 0 everywhere, i + 1 on the facets of sub-domain i, later entries
    win; an XDMF file over the mesh edges whose heavy data reads back."""
    from flowcontrol_amd import utils as flu
    from flowcontrol_amd.fem.boundary import SubDomain
    from flowcontrol_amd.fem.hdf5_min import read_dataset
    from flowcontrol_amd.fem.mesh import Mesh

    m = Mesh.unit_square(4, 4)
    left = SubDomain(lambda x, ob: ob & (np.abs(x[:, 0]) < 1e-12))
    walls = SubDomain(lambda x, ob: ob)  # every boundary facet, listed last: overwrites `left`
    top = SubDomain(lambda x, ob: ob & (np.abs(x[:, 1] - 1.0) < 1e-12))
    mk = flu.export_subdomains(m, [left, top], tmp_path / "sub.xdmf")
    assert np.bincount(mk).tolist() == [m.edges.shape[0] - 8, 4, 4]
    assert np.array_equal(read_dataset(tmp_path / "sub.h5", "/f").ravel(), mk)
    assert np.array_equal(read_dataset(tmp_path / "sub.h5", "/mesh/topology"), m.edges)
    xml = (tmp_path / "sub.xdmf").read_text()
    assert 'TopologyType="Polyline"' in xml and "sub.h5:/f" in xml
    mk2 = flu.export_subdomains(m, [left, walls], tmp_path / "sub2.xdmf")
    assert np.bincount(mk2).tolist() == [m.edges.shape[0] - 16, 0, 16]
    # the aggregator carries the predicate builders the case files use
    assert flu.near_cpp("x[0]", 1.0) == "near(x[0], 1.0, MESH_TOL)" and flu.on_boundary_cpp() == "on_boundary"


def test_signal_helpers():
    """flowcontrol_amd/signal.py (reference src/utils/signal.py): multisine spectrum, crest factor, dominant frequency."""
    from flowcontrol_amd import signal as sg

    assert sg.pad_upto([1, 2], 4, v=9) == [1, 2, 9, 9] and sg.pad_upto(np.array([1.0]), 3).tolist() == [1.0, 0.0, 0.0]
    with pytest.raises(TypeError):
        sg.pad_upto((1, 2), 3)
    assert (sg.saturate(5, 0, 1), sg.saturate(-5, 0, 1), sg.saturate(0.5, 0, 1)) == (1, 0, 0.5)
    assert np.allclose(sg.sample_lco(2.0, 10.0, 4), [10.0, 10.5, 11.0, 11.5])
    t = np.arange(4000) * 0.005
    assert sg.crest_factor(np.sin(2 * np.pi * 1.7 * t)) == pytest.approx(np.sqrt(2.0), rel=1e-3)
    assert sg.compute_signal_frequency(3.0 + np.sin(2 * np.pi * 1.7 * t) + 5 * np.exp(-t), Tf=20.0, dt=0.005) == pytest.approx(1.7, abs=0.02)
    np.random.seed(0)
    N, Fs = 256, 64.0
    y = sg.multisine(N, Fs, fmin=0.1, fmax=0.5)
    assert y.shape == (N,)
    # the record spans N - 1 sampling intervals (the reference's time axis), so project on the nominal harmonics directly
    tt = np.linspace(0.0, (N - 1) / Fs, N)
    k = np.arange(N // 2)
    amp = np.abs(np.exp(-2j * np.pi * np.outer(k * Fs / N, tt)) @ y) * 2 / N
    band = (k * Fs / N >= 0.1 * Fs / 2) & (k * Fs / N <= 0.5 * Fs / 2)
    assert amp[band].min() > 0.5 * amp[band].max() and amp[~band].max() < 0.2 * amp[band].max()
    assert np.mean(y * y) == pytest.approx(0.5, rel=0.1)  # nf unit sines / sqrt(nf): rms^2 = 1/2
    odd = sg.multisine(N, Fs, 0.1, 0.5, skip_even=True)
    amp_odd = np.abs(np.exp(-2j * np.pi * np.outer(k * Fs / N, tt)) @ odd)
    assert amp_odd[band & (k % 2 == 0)].max() < 0.2 * amp_odd[band & (k % 2 == 1)].max()
    np.random.seed(1)
    plain = np.mean([sg.crest_factor(sg.multisine(N, Fs, 0.1, 0.5)) for _ in range(10)])
    tuned = np.mean([sg.crest_factor(sg.multisine(N, Fs, 0.1, 0.5, opt_cf=30)) for _ in range(10)])
    assert tuned < plain
    mp_ = sg.multisine_MP(3, 2, unwrap=False, N=64, Fs=16.0, fmin=0.2, fmax=0.8)
    assert mp_.shape == (3, 128) and np.array_equal(mp_[:, :64], mp_[:, 64:]) and sg.multisine_MP(3, 2, N=64, Fs=16.0, fmin=0.2, fmax=0.8).shape == (384,)


def test_energy_field_on_the_p4_nodes_is_exact(tmp_path):
    """FlowSolver.compute_energy_field (reference flowsolver.py:831-841: u'.u' projected onto a P4 space): the product of two P2
    fields lies in P4, so the nodal values must equal the analytic product at every P4 node — vertices, three per edge, three per cell."""
    from types import SimpleNamespace

    from flowcontrol_amd.flowsolver import FlowSolver
    from flowcontrol_amd.fem.mesh import Mesh
    from flowcontrol_amd.fem.spaces import Function, TaylorHood

    th = TaylorHood(Mesh.unit_square(5, 4))
    x, y = th.node_coords[:, 0], th.node_coords[:, 1]
    ux, uy = 1.0 + x * y - 0.5 * y**2, 0.3 * x**2 - x + 2.0 * y  # P2 fields
    stub = SimpleNamespace(th=th, fields=SimpleNamespace(u_=Function(th.V, np.r_[ux, uy])))
    f = FlowSolver.compute_energy_field(stub, export=True, filename=tmp_path / "E.npz")
    X, Y = f.coords[:, 0], f.coords[:, 1]
    exact = (1.0 + X * Y - 0.5 * Y**2) ** 2 + (0.3 * X**2 - X + 2.0 * Y) ** 2
    assert f.values.size == th.nv + 3 * th.ne + 3 * th.nc
    assert np.allclose(f.vector().get_local(), exact, rtol=0, atol=1e-13)
    assert np.unique(np.round(f.coords, 12), axis=0).shape[0] == f.values.size  # every node once
    z = np.load(tmp_path / "E.npz")
    assert np.array_equal(z["values"], f.values)
