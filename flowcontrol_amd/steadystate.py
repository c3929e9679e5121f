"""Steady-state (base-flow) solvers: Newton and Picard.

API mirror of the reference's ``src/flowcontrol/steadystate.py``.  Every iteration runs on the
device: the operator is assembled by the HIP element loop (``fc_assemble_matrix``), factorised by
the device multifrontal numeric phase (``fc_refactor``, the structure being laid out once) and
applied with the same sweep kernels as a time step; the host only forms norms and updates the
iterate.  On a partitioned (multi-GPU) handle every rank assembles the whole (small) operator, factorises its own
sub-tree and the root, and the solves are collectives (``fc_solve`` merges the ranks' parts); before time stepping
starts the handle is not partitioned yet and every rank simply computes the base flow on its own GPU.
There is no host solver in here: without the device the first call raises.
"""

from __future__ import annotations

import logging

import numpy as np

from . import _lib
from ._lib import SLOT_BDF1, SLOT_MASS, SLOT_SCRATCH
from .fem.boundary import combine_bcs, pressure_pin
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
        self._bc_set = False
        self.solve_info = None
        # linear solves of the iterations: factors are reused across iterations as BiCGStab preconditioner and
        # renewed when an iteration needed more than `refactor_after` Krylov steps (0 in the list = direct solve)
        self.lag_factors = True
        self.krylov_max_iter, self.krylov_rtol, self.refactor_after = 40, 1e-12, 12
        self.krylov_iterations: list[int] = []
        self._factors_age: int | None = None
        self._krylov_failures = 0

    # ── helpers ──────────────────────────────────────────────────────────────
    def _device(self):
        return self.W.th.device()

    def _assemble_on_device(self, coeff, slot=SLOT_SCRATCH) -> None:
        self._device().assemble_matrix(slot, mass=coeff.mass, nu=coeff.nu, adv=coeff.adv, lin=coeff.lin, pressure=coeff.pressure,
                                       divergence=coeff.divergence)

    def _solve_increment(self, coeff, r: np.ndarray, dofs: np.ndarray) -> np.ndarray:
        """δ with  A δ = r  on the free rows and δ = 0 on the Dirichlet dofs (``r[dofs]`` is 0)."""
        pin = pressure_pin(self.W.th, dofs)  # enclosed flow: the pressure level is fixed at one dof
        dev = self._device()
        if not self._bc_set:
            # increments vanish on the Dirichlet dofs: homogeneous symmetric elimination, no lifting
            dev.set_bc(dofs, np.zeros((len(dofs), 1)))
            dev.set_pressure_pin(pin)
            self._bc_set = True
        self._assemble_on_device(coeff, SLOT_BDF1)
        dev.apply_bc(SLOT_BDF1)
        partitioned = dev.world > 1 or getattr(dev, "_force_comm", False)  # collective solves (BiCGStab included); no refinement there
        if self.lag_factors and self._factors_age is not None and self._krylov_failures < 2:
            # keep the factors of an earlier iterate as preconditioner: BiCGStab on the new operator costs a few
            # sweeps, a numeric factorisation tens of milliseconds (the reference refactorises every iteration)
            dev.update_operator(SLOT_BDF1)
            dev.set_solver_options(refine=self.krylov_max_iter, method="bicgstab", rtol=self.krylov_rtol)
            try:
                x, info = dev.solve(SLOT_BDF1, r)
                self.solve_info = info
                self.krylov_iterations.append(int(info[0]))
                self._factors_age += 1
                self._krylov_failures = 0
                if info[0] > self.refactor_after:
                    self._factors_age = None  # the iterate has moved too far: refactorise next time
                return x
            except _lib.FcError as err:
                logger.info(f"BiCGStab with lagged factors gave up ({err}); refactorising")
                self._krylov_failures += 1  # twice in a row (e.g. a singular enclosed-flow system near convergence): stop trying
            finally:
                dev.set_solver_options(0, True)
        dev.setup_solver(SLOT_BDF1, refine=0 if partitioned else 2)  # numeric factorisation on the device (first call: + structure)
        self._factors_age = 0
        self.krylov_iterations.append(0)
        x, info = dev.solve(SLOT_BDF1, r)
        self.solve_info = info
        return x

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
            self._assemble_on_device(self.forms.steady(UP0, f))
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
            up -= self._solve_increment(self.forms.steady_jacobian(UP0), F, dofs)
        raise RuntimeError("Newton solver did not converge")

    def picard(self, UP0: Function, f=None, max_iter: int = 10, tol: float = 1e-8) -> Function:
        """Fixed-point iteration with frozen advecting velocity — reference ``steadystate.py:98-159``."""
        th = self.W.th
        dofs, vals = combine_bcs(self.bcu, th.N)
        bp = self._load(f)
        dev = self._device()
        UP1 = Function(self.W)
        for i in range(max_iter):
            a, _ = self.forms.picard(UP0, f)
            # x_new = x~ + δ with x~ = the iterate carrying the boundary values: A δ = b − A x~ on the free
            # rows, δ = 0 on the Dirichlet dofs — the same x_new as the row-replaced system A x = b
            xt = UP0.vector().array().copy()
            xt[dofs] = vals
            self._assemble_on_device(a)
            r = bp - dev.spmv(SLOT_SCRATCH, xt)
            r[dofs] = 0.0
            UP1.vector().set_local(xt + self._solve_increment(a, r, dofs))
            diff = float(np.linalg.norm(UP1.vector().array() - UP0.vector().array()))
            base = float(np.linalg.norm(UP0.vector().array()))
            rel_err = diff / (base + 1e-14)
            UP0.assign(UP1)
            logger.info(f"Picard {i + 1}/{max_iter}  rel_err = {rel_err:.3e}")
            if rel_err < tol:
                logger.info(f"Picard converged (rel_err {rel_err:.3e} < tol {tol:.3e})")
                break
        return UP1
