"""ctypes binding of ``libfc_hip.so`` (C ABI in ``include/fc_hip.h``; array-level / bench / debug hooks in
``include/fc_hip_internal.h``) and its in-tree build.

The product path has no CPU fallback: if the library is missing or no MI355X is visible,
:class:`FcError` is raised — nothing silently degrades to numpy.
"""

from __future__ import annotations

import ctypes as C
import os
import sys
import shutil
import subprocess
from pathlib import Path

import numpy as np

CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = Path(os.environ["FC_LIB_PATH"]) if os.environ.get("FC_LIB_PATH") else CSRC / "libfc_hip.so"  # FC_LIB_PATH: tuning builds
SOURCES = [CSRC / "fc_hip.hip", *sorted(CSRC.glob("*.hip.h")), *sorted(CSRC.glob("*.hpp")), CSRC.parent.parent / "include" / "fc_hip.h",
           CSRC.parent.parent / "include" / "fc_hip_internal.h"]

FC_OK = 0
FC_ERR_INVALID, FC_ERR_HIP, FC_ERR_DIVERGED, FC_ERR_NOT_CONVERGED, FC_ERR_NOT_READY = -1, -2, -3, -4, -5
SLOT_BDF1, SLOT_BDF2, SLOT_MASS, SLOT_SCRATCH = 0, 1, 2, 3
METHOD_REFINE, METHOD_BICGSTAB, METHOD_GMRES = 0, 1, 2


class FcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libfc_hip error {code}: {msg}")
        self.code = code


class FcDiverged(FcError):
    pass


class FcCommInitError(FcError):
    """The RCCL communicator of a partitioned handle could not be created (library not loadable, ``ncclCommInitRank`` refused).
    Distinct from a communicator that exists and sums wrongly (``fc_comm_selftest``: plain :class:`FcError`, always fatal)."""

    def __init__(self, cause: FcError | str):
        if isinstance(cause, FcError):
            RuntimeError.__init__(self, str(cause))
            self.code = cause.code
        else:
            RuntimeError.__init__(self, str(cause))
            self.code = FC_ERR_HIP


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile ``libfc_hip.so`` for gfx950 in-tree with hipcc (cross-compiles without a GPU)."""
    if LIB_PATH.exists() and not force:
        newest = max(p.stat().st_mtime for p in SOURCES)
        if LIB_PATH.stat().st_mtime >= newest:
            return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        raise FcError(FC_ERR_HIP, "hipcc not found; cannot build libfc_hip.so")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *os.environ.get("FC_HIPCC_FLAGS", "").split(),
           "-o", str(LIB_PATH), str(SOURCES[0])]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise FcError(FC_ERR_HIP, f"hipcc failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_lp = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_H = C.c_void_p

#: name → argtypes; every entry point of include/fc_hip.h and include/fc_hip_internal.h (restype is int unless noted)
SIGNATURES: dict[str, list] = {
    "fc_device_count": [C.POINTER(C.c_int)],
    "fc_create": [C.POINTER(_H), C.c_int, C.c_int32, C.c_int32, C.c_int32, _dp, _ip, _ip],
    "fc_destroy": [_H],
    "fc_get_sizes": [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "fc_get_pattern": [_H, _ip, _ip],
    "fc_assemble_matrix": [_H, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_double, C.c_double],
    "fc_get_matrix_values": [_H, C.c_int, _dp],
    "fc_set_matrix_values": [_H, C.c_int, _dp],
    "fc_spmv": [_H, C.c_int, _dp, _dp],
    "fc_bench_spmv": [_H, C.c_int, C.c_int, C.POINTER(C.c_double)],
    "fc_set_bc": [_H, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p],
    "fc_set_force": [_H, C.c_int32, C.c_void_p],
    "fc_set_sensors": [_H, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    "fc_set_time_scheme": [_H, C.c_double, C.c_int],
    "fc_apply_bc": [_H, C.c_int],
    "fc_set_permutation": [_H, _ip],
    "fc_solver_setup": [_H, C.c_int, _ip, _ip, _dp, C.c_int32, _lp, _ip, _ip, _ip, _lp, C.c_int64, _lp, _ip, _ip, C.c_int64, _ip, C.c_int64, _dp, C.c_int32, C.c_int32, C.c_int32, C.c_int32],
    "fc_set_energy_matrix": [_H, _ip, _ip, _dp],
    "fc_factor_plan": [_H, C.c_int32, _lp, C.c_int32, _lp, C.c_int64, C.c_int64, _lp, _lp, _lp, _lp, C.c_int64, _ip, C.c_int64, _lp, C.c_int32],
    "fc_refactor": [_H, C.c_int, C.c_void_p],
    "fc_update_operator": [_H, C.c_int],
    "fc_set_front_shifts": [_H, C.c_int32, _lp, _dp],
    "fc_set_root_rows": [_H, C.c_int32, C.c_int32],
    "fc_undo_step": [_H],
    "fc_accept_factors": [_H, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int32)],
    "fc_get_factors_inexact": [_H, C.c_int, C.POINTER(C.c_int32)],
    "fc_get_partition_info": [_H, _ip],
    "fc_get_factor_values": [_H, C.c_int, C.c_int64, _dp],
    "fc_solver_set_blocks": [_H, C.c_int, C.c_int32, _lp, _ip, _ip, C.c_int64, _lp, _ip, _ip, _ip, _ip, _ip, _ip, C.c_int64, C.c_int64],
    "fc_set_solver_options": [_H, C.c_int, C.c_int, C.c_double, C.c_int],
    "fc_get_tree_info": [_H, _ip, C.POINTER(C.c_int32), C.c_void_p],
    "fc_comm_probe": [C.c_int],
    "fc_comm_destroy": [_H],
    "fc_setup_krylov": [_H, C.c_int, C.c_int32, C.c_int, C.c_int32, C.c_double, C.c_int32],
    "fc_get_krylov_info": [_H, C.c_int, _lp, C.POINTER(C.c_double)],
    "fc_set_state": [_H, _dp, _dp, C.c_void_p],
    "fc_get_state": [_H, C.c_void_p, C.c_void_p, C.c_void_p],
    "fc_get_solution": [_H, _dp],
    "fc_step": [_H, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p],
    "fc_step_begin": [_H, C.c_int, C.c_void_p, C.c_void_p, C.c_int],
    "fc_step_end": [_H, C.c_void_p, C.c_void_p, C.c_void_p],
    "fc_step_collect": [_H, C.c_void_p, C.c_void_p],
    "fc_set_rhs_operator": [_H, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "fc_run": [_H, C.c_int, C.c_int32, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int],
    "fc_assemble_rhs": [_H, C.c_int, C.c_void_p, _dp],
    "fc_solve": [_H, C.c_int, _dp, _dp, C.c_void_p],
    "fc_energy": [_H, _dp, C.POINTER(C.c_double)],
    "fc_measure": [_H, _dp, C.c_void_p],
    "fc_bench_sweeps": [_H, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int32)],
    "fc_set_partition": [_H, C.c_int32, C.c_void_p, C.c_void_p, C.c_int],
    "fc_comm_unique_id": [C.c_char_p],
    "fc_comm_init": [_H, C.c_int, C.c_int, C.c_char_p],
    "fc_comm_info": [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "fc_set_host_exchange": [_H, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "fc_comm_selftest": [_H, C.POINTER(C.c_double)],
    "fc_set_phase_timing": [_H, C.c_int],
    "fc_get_phase_timing": [_H, _dp, C.POINTER(C.c_int64)],
    "fc_set_timing": [_H, C.c_int],
    "fc_get_timing": [_H, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int64)],
    "fc_algorithmic_bytes": [_H, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "fc_setup_solver": [_H, C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32],
    "fc_get_permutation": [_H, _ip],
    "fc_set_pressure_pin": [_H, C.c_int32, C.c_double],
    "fc_get_local_cells": [_H, _ip],
    "fc_get_refactor_ms": [_H, C.c_int, C.POINTER(C.c_double)],
    "fc_get_refactor_flops": [_H, C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "fc_sym_build": [C.c_int32, C.c_int32, C.c_int32, _dp, _ip, _ip, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                     C.POINTER(C.c_void_p)],
    "fc_sym_size": [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)],
    "fc_sym_get": [C.c_void_p, C.c_char_p, _lp],
    "fc_sym_free": [C.c_void_p],
    "fc_get_solver_info": [_H, C.c_int, _lp],
    "fc_get_rowkind": [_H, np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")],
    "fc_set_stage_diag": [_H, C.c_int, _dp],
    "fc_set_factor_precision": [_H, C.c_int],
    "fc_get_factor_storage": [_H, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int64)],
    "fc_set_baseflow_bc": [_H, C.c_int32, C.c_void_p, C.c_void_p],
    "fc_picard_step": [_H, C.c_double, _dp, C.c_void_p, C.POINTER(C.c_double)],
    "fc_newton_step": [_H, C.c_double, _dp, C.c_void_p, C.POINTER(C.c_double), C.c_int],
    "fc_set_batch": [_H, C.c_int32],
    "fc_set_state_batch": [_H, C.c_int32, _dp, _dp, C.c_void_p],
    "fc_get_state_batch": [_H, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    "fc_get_solution_batch": [_H, C.c_int32, _dp],
    "fc_step_batch": [_H, C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p],
    "fc_step_batch_begin": [_H, C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_int],
    "fc_step_batch_end": [_H, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    "fc_reset_sim_batch": [_H, C.c_int32],
    "fc_step_batch_end_early": [_H, C.c_int32, C.c_void_p, C.c_void_p],
    "fc_step_batch_collect": [_H, C.c_int32, C.c_void_p, C.c_void_p],
    "fc_get_batch_info": [_H, _dp],
    "fc_bench_batch_apply": [_H, C.c_int, C.c_int, C.POINTER(C.c_double)],
    "fc_solve_batch": [_H, C.c_int, C.c_int32, _dp, _dp],
}

#: void (*fc_exchange_fn)(double* buf, int64_t n, void* user)
EXCHANGE_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int64, C.c_void_p)

_lib = None


def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm ships its own ``libamdhip64.so`` (soname
    ``libamdhip64.so.7``, but its dependants ask for it by the un-versioned file name): if libfc_hip.so were
    loaded first it would bind to /opt/rocm's copy and a later ``import torch`` would bring a second runtime
    into the process — streams and buffers of one are garbage to libraries (RCCL, rocBLAS) bound to the
    other.  Loading torch's copy first (when torch is installed) makes every later request resolve to it.
    ``FC_SYSTEM_HIP=1`` skips this (torch-free deployments)."""
    if os.environ.get("FC_SYSTEM_HIP", "0") == "1" or "torch" in sys.modules:
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
        if cand.exists():
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
    except OSError:
        pass  # fall back to the system runtime


def load(build_if_missing: bool = True) -> C.CDLL:
    """dlopen the library (building it first if the sources are newer) and set prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if build_if_missing and os.environ.get("FC_NO_BUILD", "0") != "1":
        try:
            build()
        except FcError:
            # a failed rebuild may only be ignored when the existing binary is not older than any source:
            # testing a stale library silently would void every parity claim
            if not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < max(p.stat().st_mtime for p in SOURCES):
                raise
    if not LIB_PATH.exists():
        raise FcError(FC_ERR_HIP, f"{LIB_PATH} is missing: the HIP extension is required (no CPU fallback)")
    _preload_hip_runtime()
    lib = C.CDLL(str(LIB_PATH))
    lib.fc_last_error.restype = C.c_char_p
    lib.fc_last_error.argtypes = []
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int) -> None:
    if code == FC_OK:
        return
    msg = (load().fc_last_error() or b"").decode(errors="replace")
    if code == FC_ERR_DIVERGED:
        raise FcDiverged(code, msg)
    raise FcError(code, msg)


def device_count() -> int:
    n = C.c_int(0)
    code = load().fc_device_count(C.byref(n))
    return n.value if code == FC_OK else 0


def ptr(a: np.ndarray | None):
    """void* of a contiguous array (None → NULL)."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)


__all__ = ["FcError", "FcDiverged", "build", "load", "check", "device_count", "ptr", "SIGNATURES", "LIB_PATH"]
