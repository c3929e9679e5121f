"""Device time of the numeric factorisation (fc_refactor) per operator.   python scripts/refactor_time.py [mesh ...]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from flowcontrol_amd.device import SLOT_BDF2, DeviceSolver  # noqa: E402
from flowcontrol_amd.fem.mesh import read_xdmf_mesh  # noqa: E402
from flowcontrol_amd.fem.spaces import TaylorHood  # noqa: E402

from flowcontrol_amd.examples.data import mesh_file  # noqa: E402
for name in sys.argv[1:] or ["O1", "cavity_fine"]:
    th = TaylorHood(read_xdmf_mesh(mesh_file(name)))
    dev = DeviceSolver(th)
    x = th.node_coords
    U0 = np.r_[1.0 + 0.3 * np.sin(x[:, 0]) * np.cos(0.7 * x[:, 1]), 0.2 * np.cos(0.5 * x[:, 0] + 0.1) * np.sin(x[:, 1])]
    m = th.mesh
    be = m.boundary_edges()
    be = be[m.edge_midpoints()[be, 0] < m.coords[:, 0].max() - 1e-9]
    nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
    dofs = np.sort(np.r_[nodes, nodes + th.nn])
    dev.set_bc(dofs, np.zeros((dofs.size, 1)))
    dev.assemble_matrix(SLOT_BDF2, mass=300.0, nu=0.01, adv=U0, lin=U0)
    dev.apply_bc(SLOT_BDF2)
    dev.setup_solver(SLOT_BDF2)
    ms = [dev.refactor(SLOT_BDF2) for _ in range(5)]
    b = np.random.default_rng(0).standard_normal(dev.N)
    _, info = dev.solve(SLOT_BDF2, b)
    from tests.support import ndsolver  # noqa: E402

    nodes = ndsolver.factorize_blocks(None, ndsolver.tree_of(dev), numeric=False).nodes  # (level, index, i0, ni, nb, value offset, ...)
    ni, nf = nodes[:, 3].astype(float), (nodes[:, 3] + nodes[:, 4]).astype(float)
    flops = float((2.0 * ni * nf * nf).sum())  # Gauss-Jordan: ni pivot steps, each a rank-1 update of the nf x nf front
    print(f"{name}: {nodes.shape[0]} fronts, {flops / 1e9:.2f} GFLOP per factorisation -> {flops / (min(ms) * 1e-3) / 1e12:.2f} TFLOP/s "
          f"over the whole fc_refactor ({min(ms):.2f} ms), largest front {int(nf.max())}", flush=True)
    print(f"{name}: N={dev.N} factor values {dev._n_factor_values}  fc_refactor ms {['%.2f' % v for v in ms]}  solve residual {info[1]:.2e}", flush=True)
    dev.close()
