"""Linearised state-space operators (A, E, B, C) of a FlowSolver.

API mirror of the reference's ``src/flowcontrol/operatorgetter.py`` (``get_A``, ``get_mass_matrix``,
``get_B``, ``get_C``, ``get_all``); SURVEY §8f "next" row 3.  Every matrix comes out of the same HIP
element loop (``fc_assemble_matrix``) that assembles the time-stepping operators; results are
``scipy.sparse.csr_matrix`` in the W numbering instead of ``dolfin.PETScMatrix``.

    E dq/dt = A q + B u,   y = C q,     A = −dF/dq (BC rows → identity rows), E = velocity mass
"""

from __future__ import annotations

import logging

import numpy as np
import scipy.sparse as sp
from numpy.typing import NDArray

from ._lib import SLOT_MASS, SLOT_SCRATCH
from .actuator import ACTUATOR_TYPE
from .fem.boundary import DirichletBC, combine_bcs
from .fem.spaces import Function

logger = logging.getLogger(__name__)


class OperatorGetter:
    def __init__(self, flowsolver):
        self.flowsolver = flowsolver

    def _jacobian_raw(self, UP0: Function) -> sp.csr_matrix:
        """−(steady Jacobian) without boundary conditions: −[(U0·∇)u + (u·∇)U0 + ν∇u:∇v − p div v − q div u]."""
        fs = self.flowsolver
        dev = fs.th.device()
        u = UP0.vector().array()[: 2 * fs.th.nn].copy()
        # both the autodiff and the hand-written form of the reference (operatorgetter.py:58-77) are this operator
        dev.assemble_matrix(SLOT_SCRATCH, mass=0.0, nu=-fs.forms.invRe, adv=u, lin=u, adv_scale=-1.0, lin_scale=-1.0, pressure=1.0, divergence=1.0)
        return dev.matrix(SLOT_SCRATCH)

    def get_A(self, UP0: Function | None = None, autodiff: bool = True, u_ctrl: NDArray[np.float64] | None = None) -> sp.csr_matrix:
        """A = −dF/dUP0 with ``bc.apply(Jac)``: Dirichlet rows become identity rows, columns untouched
        (reference ``operatorgetter.py:25-83``)."""
        fs = self.flowsolver
        if UP0 is None:
            UP0 = fs.fields.UP0
        if u_ctrl is None:
            fs.flush_actuators_u_ctrl()
        else:
            fs.set_actuators_u_ctrl(u_ctrl)
        J = self._jacobian_raw(UP0)
        dofs, _ = combine_bcs(fs.bc.bcu, fs.th.N)
        keep = np.ones(fs.th.N)
        keep[dofs] = 0.0
        return (sp.diags(keep) @ J + sp.diags(1.0 - keep)).tocsr()

    def get_mass_matrix(self) -> sp.csr_matrix:
        """E: velocity mass matrix on W, pressure rows zero (reference ``operatorgetter.py:85-105``)."""
        fs = self.flowsolver
        dev = fs.th.device()
        dev.assemble_matrix(SLOT_MASS, mass=1.0, nu=0.0, pressure=0.0, divergence=0.0)
        fs._mass_ready = True
        E = dev.matrix(SLOT_MASS)
        E.eliminate_zeros()
        return E

    def get_B(self, UP0: Function | None = None) -> NDArray[np.float64]:
        """One column per actuator: FORCE → load vector ∫ b·v; BC → lifting A_raw · w_lift
        (reference ``operatorgetter.py:107-190``)."""
        fs = self.flowsolver
        if UP0 is None:
            UP0 = fs.fields.UP0
        th = fs.th
        acts = fs.params_control.actuator_list
        B = np.zeros((th.N, len(acts)))
        saved = fs.get_actuators_u_ctrl()
        try:
            A_raw = None
            if any(a.actuator_type is ACTUATOR_TYPE.BC for a in acts):
                fs.flush_actuators_u_ctrl()
                A_raw = self._jacobian_raw(UP0)
            fs._normalise_force_actuators()
            fs.set_actuators_u_ctrl(len(acts) * [1.0])
            E = None
            for ii, a in enumerate(acts):
                if a.actuator_type is ACTUATOR_TYPE.FORCE:
                    if E is None:
                        E = self.get_mass_matrix()
                    v = a.expression(th.node_coords)
                    B[:, ii] = E @ np.r_[v[:, 0], v[:, 1], np.zeros(th.nv)]
                elif a.actuator_type is ACTUATOR_TYPE.BC:
                    w = np.zeros(th.N)
                    d, val = DirichletBC(fs.W.sub(0), a.expression, a.boundary).dof_values()
                    w[d] = val
                    B[:, ii] = A_raw @ w
                else:
                    raise NotImplementedError(f"Actuator type {a.actuator_type} not supported in get_B")
        finally:
            fs.set_actuators_u_ctrl(saved)
        return B

    def get_C(self) -> NDArray[np.float64]:
        """One row per sensor: the assembled functional y = C q (reference ``operatorgetter.py:192-239``)."""
        fs = self.flowsolver
        C = np.zeros((fs.params_control.sensor_number, fs.th.N))
        for ii, s in enumerate(fs.params_control.sensor_list):
            idx, w = s.row(fs)
            np.add.at(C[ii], idx, w)
        return C

    def get_all(self, autodiff: bool = True, u_ctrl: NDArray[np.float64] | None = None) -> tuple:
        return self.get_A(autodiff=autodiff, u_ctrl=u_ctrl), self.get_mass_matrix(), self.get_B(), self.get_C()
