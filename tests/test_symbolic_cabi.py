"""The in-library symbolic phase (csrc/fc_symbolic.hpp, behind fc_setup_solver) against its readable specification,
flowcontrol_amd/ndsolver.py: every table must be identical, entry by entry — single GPU, per-rank (world 4) and
truncated layouts.  No device involved (fc_sym_build)."""
import ctypes as C

import numpy as np
import pytest

from flowcontrol_amd import _lib
from tests.support import ndsolver
from flowcontrol_amd.fem.mesh import Mesh, read_xdmf_mesh
from flowcontrol_amd.fem.spaces import TaylorHood
from flowcontrol_amd.examples.data import mesh_file  # noqa: E402


def _tables(th, dofs, depth, merge, world, rank, truncate):
    lib = _lib.load()
    m = th.mesh
    sym = C.c_void_p()
    bd = np.ascontiguousarray(dofs, dtype=np.int32)
    _lib.check(lib.fc_sym_build(m.num_vertices, m.num_edges, m.num_cells, np.ascontiguousarray(m.coords, dtype=np.float64),
                                np.ascontiguousarray(m.cells, dtype=np.int32), np.ascontiguousarray(m.cell_edges, dtype=np.int32),
                                bd.size, _lib.ptr(bd), depth, merge, world, rank, truncate, C.byref(sym)))

    def get(name):
        n = C.c_int64()
        _lib.check(lib.fc_sym_size(sym, name.encode(), C.byref(n)))
        out = np.empty(n.value, dtype=np.int64)
        if n.value:
            _lib.check(lib.fc_sym_get(sym, name.encode(), out))
        return out

    return get, lambda: lib.fc_sym_free(sym)


def _bc(th):
    m = th.mesh
    be = m.boundary_edges()
    be = be[m.edge_midpoints()[be, 0] < m.coords[:, 0].max() - 1e-9]
    nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
    return np.sort(np.r_[nodes, nodes + th.nn])


@pytest.mark.parametrize("case", ["square", "O1", "O1_world4_rank2", "O1_world8_rank5", "O1_truncate2"])
def test_library_symbolic_phase_equals_the_python_specification(case, golden_dir):
    if case == "square":
        th, depth, world, rank, truncate = TaylorHood(Mesh.unit_square(12, 12)), 6, 1, 0, 0
    else:
        th = TaylorHood(read_xdmf_mesh(mesh_file("O1")))
        world = 4 if "world4" in case else 8 if "world8" in case else 1
        rank = 2 if "world4" in case else 5 if "world8" in case else 0
        depth, truncate = 10, (2 if "truncate" in case else 0)
    dofs = _bc(th)
    skip = np.zeros(th.N, bool)
    skip[dofs] = True
    p = int(np.log2(world))
    tree = ndsolver.build_tree(th.cell_dofs, th.mesh.cell_centroids(), th.N, depth, skip, merge=2, top_bits=p)
    keep = ndsolver.rank_keeps(tree, rank, world) if world > 1 else ((lambda k, n: k >= truncate) if truncate else None)
    fac = ndsolver.factorize_blocks(None, tree, numeric=False, keep=keep)
    if truncate:
        fac.stage_kind[tree.depth : tree.depth + truncate] = 2
    get, free = _tables(th, dofs, depth, 2, world, rank, truncate)
    try:
        assert np.array_equal(get("perm"), tree.perm)
        assert np.array_equal(get("leaf_of_cell"), tree.leaf_of_cell)
        for name in ("idx", "seg_ptr", "seg_val", "seg_col", "seg_len", "stage_begin", "stage_row0", "stage_nrows", "stage_kind"):
            assert np.array_equal(get(name), getattr(fac, name)), name
        assert np.array_equal(get("nodes"), fac.nodes.reshape(-1))
        assert get("n_val")[0] == fac.vals.size and get("nnz")[0] == fac.nnz
        # pattern of the handle = pattern of the host discretisation
        import scipy.sparse as sp

        cd = th.cell_dofs.astype(np.int64)
        mask = np.ones((15, 15), bool)
        mask[12:, 12:] = False
        ii, jj = np.nonzero(mask)
        P = sp.coo_matrix((np.ones(cd.shape[0] * ii.size), (cd[:, ii].reshape(-1), cd[:, jj].reshape(-1))), shape=(th.N, th.N)).tocsr()
        P.sum_duplicates()
        P.sort_indices()
        plan = ndsolver.factor_plan(fac, P.indptr, P.indices, skip, keep=keep)
        for name, ref in (("plan_nodes", plan.nodes.reshape(-1)), ("level_ptr", plan.level_ptr), ("a_src", plan.a_src), ("a_dst", plan.a_dst),
                          ("a_ptr", plan.a_ptr), ("ext_off", plan.ext_off), ("ext_p", plan.ext_p), ("ap_src", plan.ap_src)):
            assert np.array_equal(get(name), ref), name
        assert get("front_size")[0] == plan.front_size and get("max_slots")[0] == plan.max_slots
        part = ndsolver.partition(fac, rank, world)
        for name in ("rowkind", "local_cells", "seg_ptr", "seg_val", "seg_col", "seg_len", "stage_begin", "stage_row0", "stage_nrows", "stage_kind"):
            assert np.array_equal(get("part_" + name), getattr(part, name)), name
        assert list(get("part_ar")) == [part.ar_stage, part.ar_row0, part.ar_n, part.ar2_stage, part.root_row0, part.root_nrows]
        bb, bc, bl, bval, brow0, bnr, bi0, bni, bidx, bnb = ndsolver.down_blocks(fac, rank, world)
        for name, ref in (("begin", bb), ("count", bc), ("lpr", bl), ("val", bval), ("row0", brow0), ("nrows", bnr), ("i0", bi0), ("ni", bni),
                          ("idx", bidx), ("nb", bnb)):
            assert np.array_equal(get("blk_" + name), ref), name
    finally:
        free()
