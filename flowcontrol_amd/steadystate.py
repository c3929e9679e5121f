"""Steady-state (base-flow) solvers: Newton and Picard.

API mirror of the reference's ``src/flowcontrol/steadystate.py``.  Setup-only code: every
iteration assembles its operator with the HIP element loop (``fc_assemble_matrix``), the sparse
direct solve of the iteration runs on the host (SuperLU with a nested-dissection ordering) —
moving it onto the device is SURVEY §8f "next" row 1.
"""

from __future__ import annotations

import logging

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import ndsolver
from ._lib import SLOT_MASS, SLOT_SCRATCH
from .fem.boundary import combine_bcs
from .fem.spaces import Function
from .nsforms import NSForms

logger = logging.getLogger(__name__)


class SteadyStateSolver:
    def __init__(self, W, bcu: list, forms: NSForms, verbose: bool = True) -> None:
        self.W = W
        self.bcu = bcu
        self.forms = forms
        self.verbose = verbose
        self._perm = None

    # ── helpers ──────────────────────────────────────────────────────────────
    def _device(self):
        return self.W.th.device()

    def _ordering(self, bc_dofs: np.ndarray) -> np.ndarray:
        if self._perm is None:
            th = self.W.th
            skip = np.zeros(th.N, dtype=bool)
            skip[bc_dofs] = True
            depth = max(2, int(np.ceil(np.log2(max(th.nc, 1) / 12.0))))
            self._perm = ndsolver.build_tree(th.cell_dofs, th.mesh.cell_centroids(), th.N, depth, skip).perm
        return self._perm

    def _solve(self, A: sp.csr_matrix, b: np.ndarray, bc_dofs: np.ndarray) -> np.ndarray:
        p = self._ordering(bc_dofs)
        lu = spla.splu(A[p][:, p].tocsc(), permc_spec="NATURAL", diag_pivot_thresh=0.01)
        x = np.empty_like(b)
        x[p] = lu.solve(b[p])
        return x

    def _assemble(self, coeff) -> sp.csr_matrix:
        dev = self._device()
        dev.assemble_matrix(SLOT_SCRATCH, mass=coeff.mass, nu=coeff.nu, adv=coeff.adv, lin=coeff.lin, pressure=coeff.pressure, divergence=coeff.divergence)
        return dev.matrix(SLOT_SCRATCH)

    def _load(self, f) -> np.ndarray:
        """∫ f·v for the P2-interpolated body force (zero when there is no FORCE actuator)."""
        th = self.W.th
        if f is None:
            return np.zeros(th.N)
        nodal = np.asarray(f(th.node_coords), dtype=np.float64)
        if not np.any(nodal):
            return np.zeros(th.N)
        dev = self._device()
        dev.assemble_matrix(SLOT_MASS, mass=1.0, nu=0.0, pressure=0.0, divergence=0.0)
        x = np.zeros(th.N)
        x[: th.nn], x[th.nn : 2 * th.nn] = nodal[:, 0], nodal[:, 1]
        return dev.spmv(SLOT_MASS, x)

    @staticmethod
    def _rows_to_identity(A: sp.csr_matrix, dofs: np.ndarray) -> sp.csr_matrix:
        keep = np.ones(A.shape[0])
        keep[dofs] = 0.0
        return (sp.diags(keep) @ A + sp.diags(1.0 - keep)).tocsr()

    # ── public API ───────────────────────────────────────────────────────────
    def newton(self, UP0: Function, f=None, max_iter: int = 25, rtol: float = 1e-9, atol: float = 1e-10) -> Function:
        """Newton iteration on the steady residual with dolfin's NewtonSolver defaults
        (residual criterion, rel 1e-9 / abs 1e-10) — reference ``steadystate.py:60-96``."""
        th = self.W.th
        dofs, vals = combine_bcs(self.bcu, th.N)
        up = UP0.vector().array()
        up[dofs] = vals
        Lf = self._load(f)
        dev = self._device()
        r0 = None
        for it in range(max_iter + 1):
            self._assemble(self.forms.steady(UP0, f))
            F = dev.spmv(SLOT_SCRATCH, up) - Lf
            F[dofs] = 0.0
            r = float(np.linalg.norm(F))
            r0 = r if r0 is None else r0
            if self.verbose:
                logger.info(f"Newton iteration {it}: r (abs) = {r:.3e} r (rel) = {r / max(r0, 1e-300):.3e}")
            if r < atol or r < rtol * r0:
                return UP0
            if it == max_iter:
                break
            J = self._rows_to_identity(self._assemble(self.forms.steady_jacobian(UP0)), dofs)
            up -= self._solve(J, F, dofs)
        raise RuntimeError("Newton solver did not converge")

    def picard(self, UP0: Function, f=None, max_iter: int = 10, tol: float = 1e-8) -> Function:
        """Fixed-point iteration with frozen advecting velocity — reference ``steadystate.py:98-159``."""
        th = self.W.th
        dofs, vals = combine_bcs(self.bcu, th.N)
        bp = self._load(f)
        bp[dofs] = vals
        UP1 = Function(self.W)
        for i in range(max_iter):
            a, _ = self.forms.picard(UP0, f)
            Ap = self._rows_to_identity(self._assemble(a), dofs)
            UP1.vector().set_local(self._solve(Ap, bp, dofs))
            diff = float(np.linalg.norm(UP1.vector().array() - UP0.vector().array()))
            base = float(np.linalg.norm(UP0.vector().array()))
            rel_err = diff / (base + 1e-14)
            UP0.assign(UP1)
            logger.info(f"Picard {i + 1}/{max_iter}  rel_err = {rel_err:.3e}")
            if rel_err < tol:
                logger.info(f"Picard converged (rel_err {rel_err:.3e} < tol {tol:.3e})")
                break
        return UP1
