"""Actuators (mirror of the reference's ``src/flowcontrol/actuator.py``).

The reference stores a JIT-compiled C++ ``dolfin.Expression`` with a mutable ``u_ctrl`` parameter
on each actuator.  Every one of those expressions is *linear in u_ctrl*
(``actuator.py:190-199,241-251,269-276,297-313``), so here an expression is
``u_ctrl * profile(x)`` with a vectorised numpy ``profile``; the profile is tabulated once at the
Dirichlet / P2 nodes and shipped to the device (``fc_set_bc`` / ``fc_set_force``), and a step only
sends the ``u_ctrl`` scalars.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from enum import IntEnum
from typing import TYPE_CHECKING, Callable

import numpy as np
from numpy.typing import NDArray

if TYPE_CHECKING:
    from .flowsolver import FlowSolver


class ACTUATOR_TYPE(IntEnum):
    BC = 1
    FORCE = 2


class CYLINDER_ACTUATION_MODE(IntEnum):
    SUCTION = 1
    ROTATION = 2


class ActuatorExpression:
    """``value(x) = u_ctrl * profile(x)``; ``profile`` maps (n,2) points to (n,2) vectors."""

    def __init__(self, profile: Callable[[np.ndarray], np.ndarray], u_ctrl: float = 0.0, **params):
        self._profile = profile
        self.u_ctrl = u_ctrl
        for k, v in params.items():
            setattr(self, k, v)

    def profile(self, x: np.ndarray) -> np.ndarray:
        return np.asarray(self._profile(np.atleast_2d(np.asarray(x, dtype=np.float64))), dtype=np.float64)

    def __call__(self, x: np.ndarray) -> np.ndarray:
        return self.u_ctrl * self.profile(x)

    def value_shape(self) -> tuple[int]:
        return (2,)


@dataclass(kw_only=True)
class Actuator(ABC):
    actuator_type: ACTUATOR_TYPE
    expression: ActuatorExpression | None = None

    @abstractmethod
    def _load_expression(self, V, mesh) -> ActuatorExpression:
        ...

    def load_expression(self, flowsolver: "FlowSolver") -> ActuatorExpression:
        self.expression = self._load_expression(flowsolver.V, flowsolver.mesh)
        return self.expression


@dataclass(kw_only=True)
class ActuatorBC(Actuator):
    """Dirichlet actuator; ``boundary_name`` is resolved against ``FlowSolver.boundaries``."""

    boundary_name: str | None = None
    boundary: object | None = None

    def load_expression(self, flowsolver: "FlowSolver") -> ActuatorExpression:
        super().load_expression(flowsolver)
        if self.boundary_name is not None:
            try:
                self.boundary = flowsolver.get_subdomain(self.boundary_name)
            except KeyError:
                available = list(flowsolver.boundaries.index)
                raise KeyError(
                    f"Actuator boundary_name={self.boundary_name!r} not found in "
                    f"FlowSolver.boundaries. Available: {available}"
                ) from None
        return self.expression


@dataclass(kw_only=True)
class ActuatorBCParabolicV(ActuatorBC):
    """Parabolic wall-normal (y) velocity on the slot |x - x0| < L (reference ``actuator.py:170-232``)."""

    width: float = 0.0
    position_x: float = 0.0
    actuator_type: ACTUATOR_TYPE = ACTUATOR_TYPE.BC

    def _load_expression(self, V, mesh) -> ActuatorExpression:
        L, x0 = self.width, self.position_x

        def profile(x):
            d = x[:, 0] - x0
            v = np.where((d >= L) | (d <= -L), 0.0, -1.0 * (d + L) * (d - L) / (L * L))
            return np.stack([np.zeros_like(v), v], axis=1)

        return ActuatorExpression(profile, u_ctrl=0.0, L=L, x0=x0)

    @staticmethod
    def angular_size_deg_to_width(angular_size_deg: float, cylinder_radius: float) -> float:
        return cylinder_radius * np.sin(0.5 * angular_size_deg * np.pi / 180.0)


@dataclass(kw_only=True)
class ActuatorBCRotation(ActuatorBC):
    """Tangential velocity of a cylinder of diameter d spinning at rate u_ctrl (``actuator.py:235-260``)."""

    position_x: float = 0.0
    position_y: float = 0.0
    diameter: float = 1.0
    actuator_type: ACTUATOR_TYPE = ACTUATOR_TYPE.BC

    def _load_expression(self, V, mesh) -> ActuatorExpression:
        x0, y0, d = self.position_x, self.position_y, self.diameter

        def profile(x):
            th = np.arctan2(x[:, 1] - y0, x[:, 0] - x0)
            return np.stack([-np.sin(th) * d / 2.0, np.cos(th) * d / 2.0], axis=1)

        return ActuatorExpression(profile, u_ctrl=0.0, x0=x0, y0=y0, d=d)


@dataclass(kw_only=True)
class ActuatorBCUniformU(ActuatorBC):
    """Uniform streamwise velocity (u_ctrl, 0) (lid; ``actuator.py:263-283``)."""

    actuator_type: ACTUATOR_TYPE = ACTUATOR_TYPE.BC

    def _load_expression(self, V, mesh) -> ActuatorExpression:
        return ActuatorExpression(lambda x: np.stack([np.ones(x.shape[0]), np.zeros(x.shape[0])], axis=1), u_ctrl=0.0)


@dataclass(kw_only=True)
class ActuatorForceGaussianV(Actuator):
    """Unit-L2-norm Gaussian body force on the y-momentum equation (``actuator.py:286-313``).

    η = 1/‖f‖_L2 where the norm is that of the P2 interpolant of the Gaussian (``dolfin.norm`` of an
    ``Expression(element=V.ufl_element())``), evaluated with the velocity mass matrix on the device.
    """

    sigma: float
    position: NDArray[np.float64]
    actuator_type: ACTUATOR_TYPE = ACTUATOR_TYPE.FORCE

    def _load_expression(self, V, mesh) -> ActuatorExpression:
        sig, x10, x20 = self.sigma, float(self.position[0]), float(self.position[1])

        def gauss(x):
            r2 = (x[:, 0] - x10) ** 2 + (x[:, 1] - x20) ** 2
            return np.exp(-0.5 * r2 / (sig * sig))

        expr = ActuatorExpression(lambda x: np.stack([np.zeros(x.shape[0]), expr.eta * gauss(x)], axis=1), u_ctrl=0.0, eta=1.0, sig=sig, x10=x10, x20=x20)
        return expr

    def load_expression(self, flowsolver: "FlowSolver") -> ActuatorExpression:
        """Build the (un-normalised) expression; η is fixed by :meth:`normalise` the first time the
        solver needs the force (the norm is evaluated with the device mass matrix, and building a
        FlowSolver must not require a GPU)."""
        expr = super().load_expression(flowsolver)
        expr.eta = 1.0
        expr.u_ctrl = 0.0
        self._normalised = False
        return expr

    def normalise(self, l2_norm_of_unit_profile: float) -> None:
        self.expression.eta = 1.0 / l2_norm_of_unit_profile
        self._normalised = True
