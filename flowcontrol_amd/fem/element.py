"""Reference-element tables for the Taylor–Hood (P2 velocity / P1 pressure) triangle.

The reference builds these through UFL/FFC (``VectorElement("CG", 2)``,
``FiniteElement("CG", 1)`` in ``src/flowcontrol/flowsolver.py:242-250``).  All integrands
of ``src/flowcontrol/nsforms.py`` are polynomials of degree ≤ 5 on affine triangles, so a
degree-5 rule integrates them exactly; we use the 7-point Radon rule.

Local node order: 0,1,2 = vertices; 3,4,5 = midpoints of the edges opposite vertex 0,1,2.
Reference coordinates (ξ, η): λ0 = 1-ξ-η, λ1 = ξ, λ2 = η.
"""

from __future__ import annotations

import numpy as np

_S15 = np.sqrt(15.0)
_A1 = (6.0 - _S15) / 21.0
_A2 = (6.0 + _S15) / 21.0
_W0 = 9.0 / 40.0
_W1 = (155.0 - _S15) / 1200.0
_W2 = (155.0 + _S15) / 1200.0

#: barycentric quadrature points (7, 3) and weights summing to 1 (multiply by cell area)
QUAD_BARY = np.array(
    [
        [1.0 / 3.0, 1.0 / 3.0, 1.0 / 3.0],
        [1.0 - 2.0 * _A1, _A1, _A1],
        [_A1, 1.0 - 2.0 * _A1, _A1],
        [_A1, _A1, 1.0 - 2.0 * _A1],
        [1.0 - 2.0 * _A2, _A2, _A2],
        [_A2, 1.0 - 2.0 * _A2, _A2],
        [_A2, _A2, 1.0 - 2.0 * _A2],
    ]
)
QUAD_W = np.array([_W0, _W1, _W1, _W1, _W2, _W2, _W2])
NQ = 7

# derivatives of barycentric coordinates w.r.t. (ξ, η)
DLAM_REF = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])

_EDGE_VERTS = ((1, 2), (2, 0), (0, 1))  # local edge k is opposite vertex k


def p1_basis(lam: np.ndarray) -> np.ndarray:
    """P1 basis values at barycentric points ``lam`` (..., 3) → (..., 3)."""
    return np.asarray(lam, dtype=np.float64)


def p2_basis(lam: np.ndarray) -> np.ndarray:
    """P2 basis values at barycentric points (..., 3) → (..., 6)."""
    lam = np.asarray(lam, dtype=np.float64)
    out = np.empty(lam.shape[:-1] + (6,))
    for i in range(3):
        out[..., i] = lam[..., i] * (2.0 * lam[..., i] - 1.0)
    for k, (i, j) in enumerate(_EDGE_VERTS):
        out[..., 3 + k] = 4.0 * lam[..., i] * lam[..., j]
    return out


def p2_grad_ref(lam: np.ndarray) -> np.ndarray:
    """Reference gradients d/d(ξ,η) of the P2 basis at barycentric points → (..., 6, 2)."""
    lam = np.asarray(lam, dtype=np.float64)
    out = np.empty(lam.shape[:-1] + (6, 2))
    for i in range(3):
        out[..., i, :] = (4.0 * lam[..., i, None] - 1.0) * DLAM_REF[i]
    for k, (i, j) in enumerate(_EDGE_VERTS):
        out[..., 3 + k, :] = 4.0 * (lam[..., i, None] * DLAM_REF[j] + lam[..., j, None] * DLAM_REF[i])
    return out


#: tables at the 7 quadrature points
PHI2 = p2_basis(QUAD_BARY)  # (7, 6)
DPHI2 = p2_grad_ref(QUAD_BARY)  # (7, 6, 2)
PHI1 = p1_basis(QUAD_BARY)  # (7, 3)
DPHI1 = DLAM_REF.copy()  # (3, 2), constant

#: reference coordinates of the 6 P2 nodes (barycentric)
P2_NODES_BARY = np.array(
    [[1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0]], dtype=np.float64
)

__all__ = [
    "QUAD_BARY",
    "QUAD_W",
    "NQ",
    "DLAM_REF",
    "PHI2",
    "DPHI2",
    "PHI1",
    "DPHI1",
    "P2_NODES_BARY",
    "p1_basis",
    "p2_basis",
    "p2_grad_ref",
]
