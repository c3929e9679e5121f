"""Host side of the factorisation-free preconditioner (``csrc/fc_precond.hpp``: block extraction, algebraic Schur complement,
smoothed-aggregation AMG, folded transfer operators) without a GPU: a small C++ driver is compiled with g++ and its numbers checked."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_amg_hierarchy_and_folded_cycles(tmp_path):
    exe = tmp_path / "precond_host_check"
    subprocess.run(["g++", "-O2", "-std=c++17", str(ROOT / "tests" / "support" / "precond_host_check.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    kv = {}
    for line in out.strip().splitlines():
        k, *v = line.split()
        kv[k] = v
    assert int(kv["levels"][0]) >= 2 and 1 <= int(kv["coarse"][0]) <= 256  # 2 304 rows coarsen by ~8 per level to a dense coarsest level
    # the folded operators ARE the plain cycles (two products per level instead of five / seven)
    assert float(kv["fold11"][0]) < 1e-13 and float(kv["fold22"][0]) < 1e-13
    # ... and the cycles converge like multigrid should on a Poisson matrix, the two-sweep cycle faster
    assert float(kv["rate1"][0]) < 0.5 and float(kv["rate2"][0]) < float(kv["rate1"][0]) and float(kv["residual2"][0]) < 1e-3
    # saddle-point blocks in compact numberings (reversed permutation) and the algebraic Schur complement B diag(F)^-1 Bt
    assert kv["blocks"] == ["4", "1", "7", "4", "4"]  # F without its explicit zero
    assert abs(float(kv["schur"][0]) - (0.25 / 4 + 0.25 / 3 + 0.0625 / 5 + 0.0625 / 6)) < 1e-12
