"""Continuous-time LTI controller with cached ZOH discretisation.

Mirror of the reference's ``src/flowcontrol/controller.py`` (``Controller(A, B, C, D, file, x0)``,
``from_file``, ``from_matrices``, ``step(y, dt)``, ``reset``, ``x`` and the ``+ * inv`` algebra).
The reference subclasses ``control.StateSpace`` and discretises with ``control.c2d(..., "zoh")``
(``controller.py:121-134``); python-control is not a dependency here — ZOH is the exact
augmented-matrix exponential, which is what ``c2d`` computes.
"""

from __future__ import annotations

import warnings
from pathlib import Path

import numpy as np
import scipy.io as sio
from numpy.typing import NDArray
from scipy.linalg import expm


def read_matfile(path):
    """Load a MATLAB v5 file without the duplicate-variable warning (``utils/lticontrol.py:20-24``)."""
    with warnings.catch_warnings():
        warnings.filterwarnings("ignore", "Duplicate variable name*")
        return sio.loadmat(str(path))


class Controller:
    def __init__(self, A, B, C, D, file: Path | None = None, x0: NDArray[np.float64] | None = None):
        A = np.atleast_2d(np.array(A, dtype=np.float64))
        n = A.shape[0] if A.size else 0
        A = A.reshape(n, n)
        B = np.array(B, dtype=np.float64)
        B = B.reshape(n, -1) if n else np.zeros((0, max(1, B.size)))
        m = B.shape[1]
        C = np.array(C, dtype=np.float64)
        C = C.reshape(-1, n) if n else np.zeros((max(1, C.size), 0))
        p = C.shape[0]
        D = np.array(D, dtype=np.float64)
        D = np.full((p, m), float(D.reshape(-1)[0])) if D.size == 1 else D.reshape(p, m)
        self.A, self.B, self.C, self.D = A, B, C, D
        self.nstates, self.ninputs, self.noutputs = n, m, p
        self.file = file
        self.x = np.array(x0, dtype=np.float64) if x0 is not None else np.zeros((n,))

    @classmethod
    def from_file(cls, file: Path, x0=None) -> "Controller":
        m = read_matfile(file)
        return cls(m["A"], m["B"], m["C"], m["D"], x0=x0, file=file)

    @classmethod
    def from_matrices(cls, A, B, C, D, file: Path | None = None, x0=None) -> "Controller":
        return cls(A, B, C, D, x0=x0, file=file)

    def _discretize(self, dt: float) -> None:
        n, m = self.nstates, self.ninputs
        M = np.zeros((n + m, n + m))
        M[:n, :n] = self.A * dt
        M[:n, n:] = self.B * dt
        E = expm(M)
        self._Ad, self._Bd = E[:n, :n], E[:n, n:]
        self._Cd, self._Dd = self.C, self.D
        self._dt = dt

    def step(self, y, dt: float) -> NDArray[np.float64]:
        """u = Cd x + Dd y ; x ← Ad x + Bd y  (``controller.py:136-159``)."""
        if not hasattr(self, "_dt") or self._dt != dt:
            self._discretize(dt)
        y = np.atleast_1d(np.asarray(y, dtype=np.float64))
        u = self._Cd @ self.x + self._Dd @ y
        self.x = self._Ad @ self.x + self._Bd @ y
        return u

    def reset(self) -> None:
        self.x = np.zeros((self.nstates,))

    # ── algebra (parallel, series, inverse) preserving the Controller type ───
    def _coerce(self, other) -> "Controller":
        if isinstance(other, Controller):
            return other
        g = np.atleast_2d(np.asarray(other, dtype=np.float64))
        return Controller(np.zeros((0, 0)), np.zeros((0, g.shape[1])), np.zeros((g.shape[0], 0)), g)

    def _with_state(self, K: "Controller", other) -> "Controller":
        if isinstance(other, Controller):
            K.x = np.concatenate((self.x, other.x), axis=0)
        return K

    def __add__(self, other) -> "Controller":
        o = self._coerce(other)
        n1, n2 = self.nstates, o.nstates
        A = np.block([[self.A, np.zeros((n1, n2))], [np.zeros((n2, n1)), o.A]])
        return self._with_state(Controller(A, np.vstack([self.B, o.B]), np.hstack([self.C, o.C]), self.D + o.D), other)

    __radd__ = __add__

    def __mul__(self, other) -> "Controller":
        """Series connection ``self ∘ other`` (other acts first), states ordered [self, other]."""
        o = self._coerce(other)
        n1, n2 = self.nstates, o.nstates
        A = np.block([[self.A, self.B @ o.C], [np.zeros((n2, n1)), o.A]])
        B = np.vstack([self.B @ o.D, o.B])
        C = np.hstack([self.C, self.D @ o.C])
        return self._with_state(Controller(A, B, C, self.D @ o.D), other)

    def __rmul__(self, other) -> "Controller":
        return self._coerce(other) * self

    def inv(self) -> "Controller":
        Di = np.linalg.inv(self.D)
        return Controller(self.A - self.B @ Di @ self.C, self.B @ Di, -Di @ self.C, Di)
