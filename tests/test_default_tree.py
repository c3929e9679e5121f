"""Default shape of the elimination tree (``fc_setup_solver`` with depth 0): leaves of about 12 cells, to the NEAREST level the
fused levels allow on single-GPU handles; small meshes (factors that stay in the Infinity Cache) fuse one bisection more into each of
the two top levels.  The library rule (``csrc/fc_symbolic.hpp::default_bits``, reached through ``fc_sym_build``) and the numpy
specification (``tests/support/ndsolver.default_bits``) must agree."""
import numpy as np
import pytest

from tests.support import ndsolver
from flowcontrol_amd.fem.mesh import read_xdmf_mesh
from flowcontrol_amd.fem.spaces import TaylorHood
from flowcontrol_amd.examples.data import mesh_file  # noqa: E402


def test_default_depth_rule():
    # cells -> bisections (merge 2): O1, O1 refined, pinball, cavity_coarse, cavity_fine, lid cavity 64 x 64
    for nc, want in ((12284, 10), (49136, 12), (66668, 12), (51883, 12), (193916, 14), (8192, 10)):
        assert sum(ndsolver.default_bits(nc, 2, 0)) == want
    # ... and how they are fused into tree levels, root first: the small meshes (launch-bound sweeps) get two 8-ary top levels
    assert ndsolver.default_bits(12284, 2, 0) == [3, 3, 2, 2] and ndsolver.default_bits(8192, 2, 0) == [3, 3, 2, 2]
    assert ndsolver.default_bits(49136, 2, 0) == [2] * 6 and ndsolver.default_bits(193916, 2, 0) == [2] * 7
    assert ndsolver.default_bits(288, 2, 0) == [2, 2]  # (too shallow for the rule)
    # partitioned handles keep the rule of rounds 1-2: world-ary root, then merge-ary levels, rounded up
    assert ndsolver.default_bits(49136, 2, 3) == [3, 2, 2, 2, 2, 2] and ndsolver.default_bits(12284, 2, 1) == [1, 2, 2, 2, 2, 2]


@pytest.mark.parametrize("mesh", ["O1", "mesh_middle_gmsh"])
def test_library_default_tree_is_the_python_default_tree(mesh, golden_dir):
    from test_symbolic_cabi import _bc, _tables

    th = TaylorHood(read_xdmf_mesh(mesh_file(mesh)))
    dofs = _bc(th)
    skip = np.zeros(th.N, bool)
    skip[dofs] = True
    bits = ndsolver.default_bits(th.nc, 2, 0)
    tree = ndsolver.build_tree(th.cell_dofs, th.mesh.cell_centroids(), th.N, sum(bits), skip, bits=bits)
    get, free = _tables(th, dofs, 0, 2, 1, 0, 0)
    try:
        assert np.array_equal(get("perm"), tree.perm)
        assert np.array_equal(get("leaf_of_cell"), tree.leaf_of_cell)
    finally:
        free()
