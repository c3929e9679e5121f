"""ctypes wrapper of ``oracle/cpu_step.cpp`` — TEST INFRASTRUCTURE / CPU BASELINE, never imported by the product.

``build()`` compiles the C++ restatement in-tree (``oracle/_build/libcpu_step.so``; g++ -O3 -march=x86-64-v3, one thread).
:class:`CompiledStepper` is the compiled counterpart of ``ns_oracle.TimeStepper``: same operators and SuperLU factors,
but the per-step right-hand side (element loop + lifting), the sensors and the energy run as compiled scalar code —
what FFC-generated kernels do for the reference (``flowsolver.py:721-762``)."""
from __future__ import annotations

import ctypes as C
import shutil
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SRC = HERE / "cpu_step.cpp"
LIB = HERE / "_build" / "libcpu_step.so"
_lib = None


def build(force: bool = False) -> Path:
    if LIB.exists() and not force and LIB.stat().st_mtime >= SRC.stat().st_mtime:
        return LIB
    gxx = shutil.which("g++")
    if gxx is None:
        raise RuntimeError("g++ not found: cannot build oracle/cpu_step.cpp")
    LIB.parent.mkdir(parents=True, exist_ok=True)
    cmd = [gxx, "-O3", "-march=x86-64-v3", "-std=c++17", "-fPIC", "-shared", "-o", str(LIB), str(SRC)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"g++ failed:\n{res.stderr}")
    return LIB


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(LIB))
        _lib.cpu_energy.restype = C.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class CompiledStepper:
    """Factor-once / solve-many time stepper with compiled per-step assembly (cf. ``ns_oracle.TimeStepper``)."""

    def __init__(self, ts, sensor_rows=None):
        """``ts``: an ``ns_oracle.TimeStepper`` (operators, BC tables, lazily built SuperLU factors are reused)."""
        self.ts = ts
        d = ts.d
        self.d = d
        self.lib = load()
        self.coords = np.ascontiguousarray(d.coords, dtype=np.float64)
        self.cells = np.ascontiguousarray(d.cells, dtype=np.int64)
        self.cell_nodes = np.ascontiguousarray(d.cell_nodes, dtype=np.int64)
        self.bc_dofs = np.ascontiguousarray(ts.bc_dofs, dtype=np.int64)
        self.prof = np.ascontiguousarray(ts.bc_profiles, dtype=np.float64)
        self.n_act = self.prof.shape[1]
        # lifting vectors A_full[:, D] profile_k per order (what SystemAssembler applies cell by cell)
        self.lift = {}
        for order, A in ts.A_full.items():
            L = np.zeros((max(self.n_act, 1), d.N))
            for k in range(self.n_act):
                g = np.zeros(d.N)
                g[self.bc_dofs] = self.prof[:, k]
                L[k] = A @ g
            self.lift[order] = np.ascontiguousarray(L)
        self.b = np.empty(d.N)
        rows = sensor_rows or []
        self.s_ptr = np.zeros(len(rows) + 1, dtype=np.int64)
        for i, (idx, _) in enumerate(rows):
            self.s_ptr[i + 1] = self.s_ptr[i] + len(idx)
        self.s_idx = np.ascontiguousarray(np.concatenate([r[0] for r in rows]) if rows else np.zeros(0), dtype=np.int64)
        self.s_w = np.ascontiguousarray(np.concatenate([r[1] for r in rows]) if rows else np.zeros(0), dtype=np.float64)
        self.y = np.empty(max(len(rows), 1))
        self.n_sens = len(rows)

    def rhs(self, order, u_n, u_nn, u_ctrl) -> np.ndarray:
        d, ts = self.d, self.ts
        u_ctrl = np.ascontiguousarray(np.atleast_1d(u_ctrl), dtype=np.float64)
        nl = 1.0 if ts.nonlinear else 0.0
        if order == 1:
            cm_n, cm_nn, cc_n, cc_nn = 1.0 / ts.dt, 0.0, -nl, 0.0
        else:
            cm_n, cm_nn, cc_n, cc_nn = 2.0 / ts.dt, -0.5 / ts.dt, -2.0 * nl, nl
        f = None
        if ts.force_profiles is not None:
            f = np.ascontiguousarray(ts.force_profiles @ u_ctrl, dtype=np.float64)
        u_n = np.ascontiguousarray(u_n, dtype=np.float64)
        u_nn = None if u_nn is None else np.ascontiguousarray(u_nn, dtype=np.float64)
        self.lib.cpu_rhs_elem(C.c_int(d.nc), C.c_int(d.nn), C.c_int(d.N), _p(self.coords), _p(self.cells), _p(self.cell_nodes), _p(u_n), _p(u_nn), _p(f),
                              C.c_double(cm_n), C.c_double(cm_nn), C.c_double(cc_n), C.c_double(cc_nn), _p(self.b))
        self.lib.cpu_lift(C.c_int(d.N), C.c_int(self.n_act), C.c_int(len(self.bc_dofs)), _p(self.lift[order]), _p(self.bc_dofs), _p(self.prof),
                          _p(u_ctrl), _p(self.b))
        return self.b

    def solve(self, order, b) -> np.ndarray:
        return self.ts.solve(order, b)

    def energy(self, u) -> float:
        d = self.d
        return float(self.lib.cpu_energy(C.c_int(d.nc), C.c_int(d.nn), _p(self.coords), _p(self.cells), _p(self.cell_nodes),
                                         _p(np.ascontiguousarray(u, dtype=np.float64))))

    def sensors(self, up) -> np.ndarray:
        self.lib.cpu_sensors(C.c_int(self.n_sens), _p(self.s_ptr), _p(self.s_idx), _p(self.s_w), _p(np.ascontiguousarray(up, dtype=np.float64)), _p(self.y))
        return self.y[: self.n_sens].copy()
