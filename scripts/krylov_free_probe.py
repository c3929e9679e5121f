"""Factorisation-free Krylov mode (fc_setup_krylov) on one GPU: iterations, time per solve and per time step, bytes held.

    python scripts/krylov_free_probe.py [mesh=O1] [steps=50]
"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from flowcontrol_amd._lib import SLOT_BDF2  # noqa: E402
from flowcontrol_amd.device import DeviceSolver  # noqa: E402
from flowcontrol_amd.examples.data import mesh_file  # noqa: E402
from flowcontrol_amd.fem.mesh import read_xdmf_mesh  # noqa: E402
from flowcontrol_amd.fem.spaces import TaylorHood  # noqa: E402


def main():
    mesh = sys.argv[1] if len(sys.argv) > 1 else "O1"
    th = TaylorHood(read_xdmf_mesh(mesh_file(mesh)))
    dev = DeviceSolver(th)
    x = th.node_coords
    U0 = np.r_[1.0 + 0.3 * np.sin(x[:, 0]) * np.cos(0.7 * x[:, 1]), 0.2 * np.cos(0.5 * x[:, 0] + 0.1) * np.sin(x[:, 1])]
    m = th.mesh
    be = m.boundary_edges()
    be = be[m.edge_midpoints()[be, 0] < m.coords[:, 0].max() - 1e-9]
    nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
    dofs = np.sort(np.r_[nodes, nodes + th.nn])
    dev.set_bc(dofs, np.zeros((dofs.size, 1)))
    dt, Re = (0.005, 100.0) if "cavity" not in mesh else (4e-4, 7500.0)
    dev.assemble_matrix(SLOT_BDF2, mass=1.5 / dt, nu=1.0 / Re, adv=U0, lin=U0)
    dev.apply_bc(SLOT_BDF2)
    rng = np.random.default_rng(3)
    b = rng.standard_normal(dev.N)
    b[dofs] = 0.0
    A = dev.matrix(SLOT_BDF2).tocsr()
    print(f"[{mesh}] N {dev.N} nnz {A.nnz}")
    for method in ("gmres", "bicgstab"):
        for sweeps in (1, 2, 3, 4, 5):
            info = dev.setup_krylov(SLOT_BDF2, sweeps=sweeps, method=method, max_iter=400, rtol=1e-10)
            xs, si = dev.solve(SLOT_BDF2, b)
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                xs, si = dev.solve(SLOT_BDF2, b)
            ms = 1e3 * (time.perf_counter() - t0) / reps
            res = np.linalg.norm(b - A @ xs) / np.linalg.norm(b)
            print(f"  {method:9s} sweeps {sweeps}: iterations {int(si[0]):4d} residual {res:.2e} {ms:8.2f} ms/solve  "
                  f"{ms / max(1, int(si[0])) * 1e3:7.1f} us/iteration | omega {info['jacobi_omega']:.3f} levels {info['amg_levels']} "
                  f"coarsest {info['coarsest_rows']} launches/apply {info['launches_per_apply']} bytes {info['bytes'] / 1e6:.1f} MB setup {info['setup_ms']} ms")
    dev.close()


if __name__ == "__main__":
    main()
