"""Tree shapes of a mesh ranked by the launch model of the factor apply (no GPU): for every way of fusing D = 9, 10, 11 bisections into
3-5 tree levels of 2-4 bisections each, the in-library symbolic phase (FC_ND_SHAPE) gives the factor values; cost = (2 levels + 1) launches
x 3.5 us + 8 B x values / 5.4 TB/s.  O1: [3,2,2,3] and [3,3,2,2] lead (68.7 us) before [2,2,2,2,2] (71.0) -- as measured
(profiles/EXPERIMENTS.md).        python scripts/shape_model.py [mesh]"""
import ctypes as C
import itertools
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from flowcontrol_amd import _lib  # noqa: E402
from flowcontrol_amd.examples.data import mesh_file  # noqa: E402
from flowcontrol_amd.fem.mesh import read_xdmf_mesh  # noqa: E402
from flowcontrol_amd.fem.spaces import TaylorHood  # noqa: E402

lib = _lib.load()
th = TaylorHood(read_xdmf_mesh(mesh_file(sys.argv[1] if len(sys.argv) > 1 else "O1")))
m = th.mesh
be = m.boundary_edges()
be = be[m.edge_midpoints()[be, 0] < m.coords[:, 0].max() - 1e-9]  # Dirichlet everywhere but on the outlet
nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
dofs = np.sort(np.r_[nodes, nodes + th.nn]).astype(np.int32)


def factor_values(bits):
    os.environ["FC_ND_SHAPE"] = ",".join(map(str, bits))
    sym = C.c_void_p()
    _lib.check(lib.fc_sym_build(m.num_vertices, m.num_edges, m.num_cells, np.ascontiguousarray(m.coords, dtype=np.float64),
                                np.ascontiguousarray(m.cells, dtype=np.int32), np.ascontiguousarray(m.cell_edges, dtype=np.int32), dofs.size,
                                _lib.ptr(dofs), 0, 2, 1, 0, 0, C.byref(sym)))
    out = np.empty(1, dtype=np.int64)
    _lib.check(lib.fc_sym_get(sym, b"nnz", out))
    lib.fc_sym_free(sym)
    return int(out[0])


res = []
for D in (9, 10, 11):
    for L in (3, 4, 5):
        for bits in itertools.product((2, 3, 4), repeat=L):
            if sum(bits) == D:
                mb = 8e-6 * factor_values(bits)
                res.append((3.5 * (2 * L + 1) + mb / 5.4, bits, 2 * L + 1, mb))
res.sort()
for t, bits, launches, mb in res[:12]:
    print(f"{list(bits)}: {launches} launches, {mb:.1f} MB of factor values -> {t:.1f} us")
