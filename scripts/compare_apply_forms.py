"""Row form (fc_nd_sweep / fc_nd_down_block, one right-hand side) against block form (fc_nd_block_b<4> + fc_nd_fold_b) of the factor
apply, launch by launch, on one mesh.  Run under rocprofv3 --kernel-trace; scripts/compare_apply_forms.sh summarises the trace.

    python scripts/compare_apply_forms.py cavity_fine
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from flowcontrol_amd.device import SLOT_BDF2, DeviceSolver  # noqa: E402
from flowcontrol_amd.fem.mesh import read_xdmf_mesh  # noqa: E402
from flowcontrol_amd.fem.spaces import TaylorHood  # noqa: E402

GOLDEN = Path(__file__).resolve().parents[1] / "tests" / "golden" / "meshes"


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cavity_fine"
    th = TaylorHood(read_xdmf_mesh(GOLDEN / f"{name}.npz"))
    dev = DeviceSolver(th)
    m = th.mesh
    be = m.boundary_edges()
    be = be[m.edge_midpoints()[be, 0] < m.coords[:, 0].max() - 1e-9]
    nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
    dofs = np.sort(np.r_[nodes, nodes + th.nn])
    dev.set_bc(dofs, np.zeros((dofs.size, 1)))
    U0 = np.r_[np.ones(th.nn), np.zeros(th.nn)]
    dev.assemble_matrix(SLOT_BDF2, mass=300.0, nu=0.01, adv=U0, lin=U0)
    dev.apply_bc(SLOT_BDF2)
    dev.setup_solver(SLOT_BDF2)
    b = np.random.default_rng(0).standard_normal(dev.N)
    for _ in range(30):
        dev.solve(SLOT_BDF2, b)
    dev.set_batch(4)
    ms = dev.bench_batch_apply(SLOT_BDF2, 30)
    print(f"{name}: batched apply (KB = 4) {ms * 1e3:.1f} us", flush=True)
    dev.close()


if __name__ == "__main__":
    main()
