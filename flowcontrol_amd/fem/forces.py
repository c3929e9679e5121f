"""Boundary force of a flow state: ``F = ∫_Γ −σ·n ds`` with ``σ = 2ν sym(∇u) − p I`` and ``n`` the outward normal.

Host-side post-processing (numpy) behind the case files' ``compute_force_coefficients`` (reference
``src/examples/pinball/pinballflowsolver.py:202-232``, ``src/utils/physics.py:17-19``): off the hot path, called once
per base flow / at the end of a run.  On affine P2/P1 triangles the integrand is linear along a facet: 2-point Gauss
is exact."""
from __future__ import annotations

import numpy as np

from . import element as el


def boundary_force(th, facets: np.ndarray, nu: float, u: np.ndarray, p: np.ndarray) -> tuple[float, float]:
    """(F_x, F_y) over the boundary facets (edge ids) ``facets``; ``u`` = [ux(nn) | uy(nn)], ``p`` = (nv,)."""
    mesh = th.mesh
    nn = th.nn
    g = 0.5 / np.sqrt(3.0)
    F = np.zeros(2)
    for e in np.asarray(facets, dtype=np.int64):
        c = int(mesh.edge_cells[e, 0])
        k = int(np.nonzero(mesh.cell_edges[c] == e)[0][0])  # the facet is opposite local vertex k
        i, j = (k + 1) % 3, (k + 2) % 3
        xi, xj, xk = (mesh.coords[mesh.cells[c, m]] for m in (i, j, k))
        t = xj - xi
        length = float(np.hypot(*t))
        n = np.array([t[1], -t[0]]) / length
        if n @ (xk - xi) > 0:  # point away from the cell's third vertex
            n = -n
        nodes = th.cell_nodes[c]
        ux, uy = u[nodes], u[nn + nodes]
        pv = p[mesh.cells[c]]
        for s in (0.5 - g, 0.5 + g):
            lam = np.zeros(3)
            lam[i], lam[j] = 1.0 - s, s
            dphi = el.p2_grad_ref(lam) @ th.Jinv[c]  # (6, 2): physical gradients of the P2 basis
            G = np.array([ux @ dphi, uy @ dphi])  # G[a, b] = d u_a / d x_b
            sigma = nu * (G + G.T) - (lam @ pv) * np.eye(2)
            F += -0.5 * length * (sigma @ n)
    return float(F[0]), float(F[1])


def force_coefficients(fs, u, p, surfaces) -> dict:
    """{surface: (cl, cd)} for the named boundaries of ``fs`` (facet markers of ``fs._mark_boundaries``)."""
    D = fs.params_flow.user_data["D"]
    uinf = fs.params_flow.uinf
    nu = uinf * D / fs.params_flow.Re
    uv, pv = u.vector().get_local(), p.vector().get_local()
    out = {}
    for name in surfaces:
        idx = int(fs.boundaries.loc[name].idx)
        facets = np.nonzero(fs.bnd_markers == idx)[0]
        drag, lift = boundary_force(fs.th, facets, nu, uv, pv)
        q = 0.5 * uinf**2 * D
        out[name] = (lift / q, drag / q)
    return out
