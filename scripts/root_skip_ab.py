import sys, tempfile, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from flowcontrol_amd.comm import run_threaded
import test_world8_gpu as T
outs = run_threaded(8, T._config4_rank, 4)
print("ratio", [round(o["flops_run"]/o["flops_full"],3) for o in outs], "refactor_ms", [round(o["refactor_ms"],2) for o in outs])
