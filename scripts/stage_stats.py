"""Per-stage size of the nested-dissection factors of the O1 operator (CPU only): rows, values,
bytes, longest row — the table DESIGN.md §5 quotes next to the per-launch times of profiles/."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from flowcontrol_amd import ndsolver as nd  # noqa: E402
from flowcontrol_amd.fem.mesh import Mesh  # noqa: E402
from flowcontrol_amd.fem.spaces import TaylorHood  # noqa: E402
from oracle import ns_oracle as O  # noqa: E402  (scripts/ is tooling, not the product path)


def main(merge=2, depth=None, refine=0):
    d = np.load(ROOT / "tests/golden/meshes/O1.npz")
    mesh = Mesh.from_arrays(d["coords"], d["cells"])
    for _ in range(refine):
        mesh = mesh.refine()
    th = TaylorHood(mesh)
    disc = O.Disc.from_taylor_hood(th)
    U0 = np.zeros(2 * th.nn)
    U0[: th.nn] = 1.0
    A = O.assemble_matrix(disc, mass=300.0, nu=0.01, adv=U0, lin=U0)
    m = th.mesh
    be = m.boundary_edges()
    nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
    x = th.node_coords
    nodes = nodes[x[nodes, 0] < x[:, 0].max() - 1e-9]
    dofs = np.sort(np.r_[nodes, nodes + th.nn])
    A, _ = O.apply_bc_symmetric(A, None, dofs, np.zeros(dofs.size))
    if not depth:
        depth = int(np.ceil(np.log2(th.nc / 12.0)))
    skip = np.zeros(th.N, bool)
    skip[dofs] = True
    t = nd.build_tree(th.cell_dofs, m.cell_centroids(), th.N, depth, skip, merge=merge)
    fac = nd.factorize_blocks(None, t, numeric=False)  # structure only: sizes do not need the numbers
    print(f"depth {t.depth} nnz {fac.nnz}")
    rowlen = np.zeros(fac.seg_ptr.size - 1, np.int64)
    np.add.at(rowlen, np.repeat(np.arange(rowlen.size), np.diff(fac.seg_ptr)), fac.seg_len)
    print("stage kind rows values MB maxrow meanrow segs/row")
    for s in range(len(fac.stage_kind)):
        r0, nr = int(fac.stage_begin[s]), int(fac.stage_nrows[s])
        rl = rowlen[r0 : r0 + nr]
        ns = np.diff(fac.seg_ptr[r0 : r0 + nr + 1])
        print(s, int(fac.stage_kind[s]), nr, int(rl.sum()), round(rl.sum() * 8 / 1e6, 2), int(rl.max()), round(rl.mean(), 1), round(ns.mean(), 2))


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
