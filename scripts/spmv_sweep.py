"""SpMV tuning probe (run on the GPU box): cavity_fine-sized matrix, lanes per row from FC_SPMV_LANES."""
import os, sys, subprocess, json
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from flowcontrol_amd.device import DeviceSolver, SLOT_SCRATCH
    from flowcontrol_amd.fem.mesh import read_xdmf_mesh
    from flowcontrol_amd.fem.spaces import TaylorHood
    out = {}
    for name in ("cavity_fine", "O1"):
        th = TaylorHood(read_xdmf_mesh(ROOT / "tests/golden/meshes" / f"{name}.npz"))
        dev = DeviceSolver(th)
        U = np.r_[np.ones(th.nn), np.zeros(th.nn)]
        dev.assemble_matrix(SLOT_SCRATCH, mass=300.0, nu=0.01, adv=U, lin=U)
        x = np.random.default_rng(0).standard_normal(dev.N)
        y = dev.spmv(SLOT_SCRATCH, x)
        ms = dev.bench_spmv(SLOT_SCRATCH, 300)
        byt = dev.nnz * 12 + dev.N * 16 + (dev.N + 1) * 4
        out[name] = round(byt / ms / 1e6, 1)
        dev.close()
    print(json.dumps(out))
else:
    for lanes in ("4", "8", "16", "32"):
        env = dict(os.environ, FC_SPMV_LANES=lanes)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print("lanes", lanes, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
