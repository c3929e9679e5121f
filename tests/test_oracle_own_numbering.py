"""The oracle with a numbering of its own.

Everywhere else the oracle is handed the product's discretisation (``Disc.from_taylor_hood(th)``): the arithmetic is
independent, the dof map is not — a mistake in ``TaylorHood.cell_nodes`` common to both sides would go unseen on a mesh
for which the reference holds no constants (pinball at Re = 100, the refined cylinder mesh).  Here the oracle numbers the
P2 nodes itself from the raw mesh arrays (``Disc.from_mesh_arrays``: its own orientation fix, its own edge numbering),
the two numberings are matched by the COORDINATES of their nodes, and the operators / right-hand sides / solutions must
agree under that permutation.  CPU part: product dof map vs oracle dof map through the oracle's assembly.  GPU part: the
HIP assembly and one full time step against the own-numbered oracle.
"""
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp

from flowcontrol_amd.fem.mesh import Mesh, read_xdmf_mesh
from flowcontrol_amd.fem.spaces import TaylorHood
from oracle import ns_oracle as O
from flowcontrol_amd.examples.data import mesh_file  # noqa: E402

GOLDEN = Path(__file__).resolve().parent / "golden"


def _raw(name):
    if name == "square":
        m = Mesh.unit_square(6, 5)
        rng = np.random.default_rng(3)
        order = rng.permutation(m.coords.shape[0])  # the raw file numbers its vertices differently, and some cells clockwise
        inv = np.empty_like(order)
        inv[order] = np.arange(order.size)
        cells = inv[m.cells]
        cells[::3] = cells[::3][:, [0, 2, 1]]
        return m.coords[order], cells
    z = np.load(mesh_file(name))
    return z["coords"], z["cells"]


def _match(a, b):
    """perm with a[i] == b[perm[i]] for two point sets that are permutations of each other (exact coordinates up to 1e-12)"""
    ka = np.lexsort((np.round(a[:, 1], 10), np.round(a[:, 0], 10)))
    kb = np.lexsort((np.round(b[:, 1], 10), np.round(b[:, 0], 10)))
    perm = np.empty(a.shape[0], dtype=np.int64)
    perm[ka] = kb
    assert np.abs(a - b[perm]).max() < 1e-12, "the two discretisations do not have the same nodes"
    return perm


def _field(x):
    return np.r_[1.0 + 0.3 * np.sin(1.3 * x[:, 0]) * np.cos(0.7 * x[:, 1]), 0.2 * np.cos(0.5 * x[:, 0] + 0.1) * np.sin(0.9 * x[:, 1])]


def _both(name):
    coords, cells = _raw(name)
    th = TaylorHood(Mesh.from_arrays(coords, cells))  # the product's discretisation (its own reordering inside)
    d_own = O.Disc.from_mesh_arrays(coords, cells)  # the oracle's
    assert (d_own.nn, d_own.nv, d_own.nc) == (th.nn, th.nv, th.nc)
    pn = _match(th.node_coords, d_own.node_coords())  # product scalar node i  <->  oracle node pn[i]
    pv = _match(th.mesh.coords, d_own.coords)
    P = np.r_[pn, d_own.nn + pn, 2 * d_own.nn + pv]  # W dofs: [ux | uy | p]
    return th, d_own, P


@pytest.mark.parametrize("name", ["square", "O1", "mesh_middle_gmsh"])
def test_product_dof_map_against_the_oracles_own(name):
    th, d_own, P = _both(name)
    d_prod = O.Disc.from_taylor_hood(th)
    U_prod, U_own = _field(th.node_coords), _field(d_own.node_coords())
    A_prod = O.assemble_matrix(d_prod, mass=300.0, nu=0.01, adv=U_prod, lin=U_prod)
    A_own = O.assemble_matrix(d_own, mass=300.0, nu=0.01, adv=U_own, lin=U_own)
    diff = A_prod - sp.csr_matrix(A_own)[P][:, P]
    assert abs(diff).max() <= 1e-12 * abs(A_prod).max()
    # a right-hand side with history terms, and the energy matrix
    rng = np.random.default_rng(0)
    u_own = rng.standard_normal(2 * d_own.nn)
    Pu = P[: 2 * th.nn]
    M_prod, M_own = O.velocity_mass(d_prod), O.velocity_mass(d_own)
    assert np.isclose(u_own[Pu] @ (M_prod @ u_own[Pu]), u_own @ (M_own @ u_own), rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["O1", "mesh_middle_gmsh"])
def test_hip_assembly_and_step_against_the_own_numbered_oracle(name):
    from flowcontrol_amd._lib import SLOT_BDF2, SLOT_MASS
    from flowcontrol_amd.device import DeviceSolver

    th, d_own, P = _both(name)
    dev = DeviceSolver(th, 0)
    try:
        m = th.mesh
        be = m.boundary_edges()
        be = be[m.edge_midpoints()[be, 0] < m.coords[:, 0].max() - 1e-9]  # everything but the outflow side
        nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
        dofs = np.sort(np.r_[nodes, nodes + th.nn])
        prof = np.sin(3.0 * th.node_coords[dofs % th.nn, 1] + 0.3 * (dofs >= th.nn))[:, None]  # a function of the dof's POINT
        dt, Re = 0.005, 100.0
        U_prod, U_own = _field(th.node_coords), _field(d_own.node_coords())
        dev.set_bc(dofs, prof)
        dev.set_time_scheme(dt, True)
        dev.assemble_matrix(SLOT_BDF2, mass=1.5 / dt, nu=1.0 / Re, adv=U_prod, lin=U_prod)
        A_dev = dev.matrix(SLOT_BDF2)
        A_own = sp.csr_matrix(O.assemble_matrix(d_own, mass=1.5 / dt, nu=1.0 / Re, adv=U_own, lin=U_own))[P][:, P]
        assert abs(A_dev - A_own).max() <= 1e-12 * abs(A_own).max()
        dev.apply_bc(SLOT_BDF2)
        dev.assemble_matrix(SLOT_MASS, mass=1.0, nu=0.0, pressure=0.0, divergence=0.0)
        dev.setup_solver(SLOT_BDF2)
        # one full step: the oracle's TimeStepper in ITS numbering (BC dofs and profile carried over by the permutation)
        inv = np.empty_like(P)
        inv[P] = np.arange(P.size)
        dofs_own = P[dofs]
        order = np.argsort(dofs_own)
        ts = O.TimeStepper(d_own, Re, dt, U_own, dofs_own[order], prof[order], orders=(2,))
        rng = np.random.default_rng(1)
        un_own, unn_own = 0.1 * rng.standard_normal(2 * d_own.nn), 0.1 * rng.standard_normal(2 * d_own.nn)
        Pu = P[: 2 * th.nn]
        dev.set_state(un_own[Pu], unn_own[Pu], np.zeros(th.nv))
        uc = np.array([0.25])
        y, dE, info = dev.step(SLOT_BDF2, uc)
        up_own = ts.solve(2, ts.rhs(2, un_own, unn_own, uc))
        up_dev = dev.get_solution()
        assert np.linalg.norm(up_dev - up_own[P]) <= 1e-9 * np.linalg.norm(up_own)
        assert info[1] < 1e-10
    finally:
        dev.close()
