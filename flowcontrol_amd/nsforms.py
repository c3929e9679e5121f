"""Variational forms of the linearised incompressible Navier–Stokes equations.

API mirror of the reference's ``src/flowcontrol/nsforms.py``.  The reference returns UFL forms
that FFC compiles to element kernels; here every form is a *coefficient record* for the two
hand-written HIP element loops (``fc_mat_elem`` for bilinear parts, ``fc_rhs_elem`` for linear
parts), which integrate exactly the same integrands with a degree-5 rule:

    bilinear:  mass·(u,v) + ((adv·∇)u, v) + ((u·∇)lin, v) + nu·(∇u,∇v) − (p, div v) − (q, div u)
    linear:    cm_n (u_n,v) + cm_nn (u_nn,v) + cc_n ((u_n·∇)u_n, v) + cc_nn ((u_nn·∇)u_nn, v) + (f,v)
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class BilinearCoefficients:
    mass: float = 0.0
    nu: float = 0.0
    adv: np.ndarray | None = None  # advecting velocity, (2 nn,)
    lin: np.ndarray | None = None  # field whose gradient multiplies the trial function
    pressure: float = -1.0
    divergence: float = -1.0
    adv_scale: float = 1.0
    lin_scale: float = 1.0


@dataclass
class LinearCoefficients:
    cm_n: float = 0.0
    cm_nn: float = 0.0
    cc_n: float = 0.0
    cc_nn: float = 0.0


@dataclass
class TransientForm:
    order: int | str
    a: BilinearCoefficients
    L: LinearCoefficients
    explicit: BilinearCoefficients | None = None  # operator C with rhs -= C u_n (Crank–Nicolson)


class NSForms:
    def __init__(self, W, Re: float, dt: float, is_nonlinear: bool = True, shift: float = 0.0) -> None:
        self.W = W
        self.invRe = 1.0 / Re
        self.dt = dt
        self.is_nonlinear = is_nonlinear
        self.shift = shift

    def transient(self, order, U0, u_n=None, f=None, u_nn=None, f_n=None) -> TransientForm:
        """BDF1 / BDF2 form around the base flow U0 (reference ``nsforms.py:62-114,238-305``)."""
        U = None if U0 is None else np.asarray(U0.vector().array() if hasattr(U0, "vector") else U0)
        nl = 1.0 if self.is_nonlinear else 0.0
        dt = float(self.dt)
        if order == 1:
            a = BilinearCoefficients(mass=1.0 / dt - self.shift, nu=self.invRe, adv=U, lin=U)
            L = LinearCoefficients(cm_n=1.0 / dt, cc_n=-nl)
        elif order == 2:
            if u_nn is None:
                raise ValueError("u_nn is required for order-2 form")
            a = BilinearCoefficients(mass=1.5 / dt - self.shift, nu=self.invRe, adv=U, lin=U)
            L = LinearCoefficients(cm_n=2.0 / dt, cm_nn=-0.5 / dt, cc_n=-2.0 * nl, cc_nn=nl)
        elif order == "cn":
            if f_n is None:
                raise ValueError("f_n is required for Crank-Nicolson form")
            # θ = ½ on the linear terms, explicit (u_n·∇)u_n, implicit pressure (nsforms.py:191-236)
            a = BilinearCoefficients(mass=1.0 / dt - self.shift, nu=0.5 * self.invRe, adv=U, lin=U, adv_scale=0.5, lin_scale=0.5)
            L = LinearCoefficients(cm_n=1.0 / dt, cc_n=-nl)
            C = BilinearCoefficients(mass=0.0, nu=0.5 * self.invRe, adv=U, lin=U, adv_scale=0.5, lin_scale=0.5, pressure=0.0, divergence=0.0)
            return TransientForm(order, a, L, explicit=C)
        else:
            raise ValueError(f"order must be 1, 2, or 'cn', got {order}")
        return TransientForm(order, a, L)

    def steady(self, UP0, f=None) -> BilinearCoefficients:
        """Operator whose action on UP0 (minus the load of f) is the residual of the steady
        equations (``nsforms.py:116-147``): ((U·∇)U, v) + nu(∇U,∇v) − (P, div v) − (q, div U)."""
        u = np.asarray(UP0.vector().array())[: 2 * self.W.th.nn].copy()
        return BilinearCoefficients(mass=0.0, nu=self.invRe, adv=u, lin=None)

    def steady_jacobian(self, UP0) -> BilinearCoefficients:
        u = np.asarray(UP0.vector().array())[: 2 * self.W.th.nn].copy()
        return BilinearCoefficients(mass=0.0, nu=self.invRe, adv=u, lin=u)

    def picard(self, U0, f=None) -> tuple[BilinearCoefficients, None]:
        """Oseen operator with frozen advecting velocity (``nsforms.py:149-187``)."""
        u = np.asarray(U0.vector().array() if hasattr(U0, "vector") else U0)[: 2 * self.W.th.nn].copy()
        return BilinearCoefficients(mass=0.0, nu=self.invRe, adv=u, lin=None), None
