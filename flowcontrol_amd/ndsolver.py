"""Nested-dissection selected-inverse factorisation (host-side *setup* of the device solver).

The reference hands its time-invariant LHS to ``dolfin.LUSolver("mumps")`` once
(``src/flowcontrol/flowsolver.py:693-699,812-814``) and then performs two triangular sweeps per
step (``:729``).  Sparse triangular sweeps are the worst possible shape for a GPU (thousands of
dependent levels); because the operator never changes and HBM is plentiful, we instead build —
once — a block LDU factorisation along an *element-based nested-dissection tree* in which every
pivot block is inverted explicitly:

    P A Pᵀ = (I + L) · D · (I + U),   D = blockdiag(F11_t),   L_t = F21_t F11_t⁻¹,  U_t = F11_t⁻¹ F12_t

so that a solve is ``2·depth + 1`` *level-wide sparse mat-vecs* (no dependent recurrences inside
a level), which is what the HIP kernels in ``csrc/fc_hip.hip`` execute:

    up-sweep    k = depth-1 … 0 :  y_k = b_k − L[k, deeper] · y_deeper
    down-sweep  k = 0 … depth   :  x_k = D_k⁻¹ · y_k − U[k, shallower] · x_shallower

The Krylov / iterative-refinement wrapper on the device uses this as its preconditioner; with
fp64 factors it is exact to round-off, so one or two refinement steps reach LU-grade residuals.

This module only does the symbolic analysis and the (dense-block) numeric factorisation on the
host — the analogue of MUMPS' analysis + factorisation phase, executed at setup.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp


@dataclass
class NDTree:
    depth: int  # leaves live at this depth; level k has 2**k nodes
    perm: np.ndarray  # new → old dof index
    iperm: np.ndarray  # old → new
    node_ptr: list[np.ndarray]  # per level k: (2**k + 1,) offsets into the *new* ordering
    level_ptr: np.ndarray  # (depth + 2,) new-index offsets of levels, deepest level FIRST
    bnd: list[list[np.ndarray]]  # per level k, per node: boundary dofs (new indices, sorted)


def _bisect_cells(cent: np.ndarray, depth: int) -> np.ndarray:
    """Leaf index in [0, 2**depth) per cell by recursive coordinate-median bisection."""
    nc = cent.shape[0]
    leaf = np.zeros(nc, dtype=np.int64)
    groups = [np.arange(nc)]
    for _ in range(depth):
        nxt = []
        for g in groups:
            if g.size == 0:
                nxt += [g, g]
                continue
            c = cent[g]
            ax = int(np.argmax(c.max(axis=0) - c.min(axis=0)))
            order = np.argsort(c[:, ax], kind="stable")
            half = g.size // 2
            a, b = g[order[:half]], g[order[half:]]
            leaf[a] = leaf[a] * 2
            leaf[b] = leaf[b] * 2 + 1
            nxt += [a, b]
        groups = nxt
    return leaf


def build_tree(cell_dofs: np.ndarray, centroids: np.ndarray, N: int, depth: int, skip: np.ndarray | None = None) -> NDTree:
    """Element-based nested dissection.

    A dof is owned by the deepest tree node whose cell set contains every cell touching it
    (leaf ⇒ subdomain interior; internal node ⇒ separator).  ``skip`` marks dofs that are
    decoupled identity rows (Dirichlet) — they are parked in the leaves.
    """
    nc, nl = cell_dofs.shape
    leaf = _bisect_cells(centroids, depth)
    lo = np.full(N, np.iinfo(np.int64).max)
    hi = np.full(N, -1)
    flat = cell_dofs.reshape(-1).astype(np.int64)
    lf = np.repeat(leaf, nl)
    np.minimum.at(lo, flat, lf)
    np.maximum.at(hi, flat, lf)
    if np.any(hi < 0):
        raise ValueError("dof without any cell")
    if skip is not None:
        hi = np.where(skip, lo, hi)
    x = lo ^ hi
    nbits = np.zeros(N, dtype=np.int64)
    nz = x > 0
    nbits[nz] = np.floor(np.log2(x[nz])).astype(np.int64) + 1
    level = depth - nbits  # owner depth
    prefix = lo >> nbits  # owner index within its level
    # ordering: deepest level first, then node, then original index (locality)
    key = np.lexsort((np.arange(N), prefix, -level))
    perm = key.astype(np.int64)
    iperm = np.empty(N, dtype=np.int64)
    iperm[perm] = np.arange(N)
    node_ptr: list[np.ndarray] = [None] * (depth + 1)
    level_ptr = np.zeros(depth + 2, dtype=np.int64)
    pos = 0
    for i, k in enumerate(range(depth, -1, -1)):
        cnt = np.bincount(prefix[level == k], minlength=2**k)
        node_ptr[k] = pos + np.r_[0, np.cumsum(cnt)]
        pos += int(cnt.sum())
        level_ptr[i + 1] = pos
    # boundary sets: dofs touching cells of subtree(t) that are owned by a proper ancestor
    bnd: list[list[np.ndarray]] = [None] * (depth + 1)
    new_cell_dofs = iperm[cell_dofs.astype(np.int64)]
    lvl_new = level[perm]
    for k in range(depth, -1, -1):
        sub = leaf >> (depth - k)
        order = np.argsort(sub, kind="stable")
        starts = np.searchsorted(sub[order], np.arange(2**k + 1))
        out = []
        for t in range(2**k):
            cells = order[starts[t] : starts[t + 1]]
            dd = np.unique(new_cell_dofs[cells].reshape(-1))
            out.append(dd[lvl_new[dd] < k])
        bnd[k] = out
    return NDTree(depth, perm, iperm, node_ptr, level_ptr, bnd)


@dataclass
class NDFactors:
    """Level-wise sparse factors in the permuted ordering (CSR arrays, ready for upload)."""

    tree: NDTree
    N: int
    up: list[sp.csr_matrix]  # up[i]: rows of level k = depth-1-i, shape (rows_k, N), entries = −L
    down: list[sp.csr_matrix]  # down[k]: rows of level k, shape (rows_k, 2N): [D⁻¹ on y | −U on x]
    nnz: int

    def solve(self, b: np.ndarray) -> np.ndarray:
        """Host reference of the device apply (used by the CPU tests of this module)."""
        t = self.tree
        y = b[t.perm].astype(np.float64).copy()
        row0 = lambda k: int(t.node_ptr[k][0])  # noqa: E731
        row1 = lambda k: int(t.node_ptr[k][-1])  # noqa: E731
        for i, k in enumerate(range(t.depth - 1, -1, -1)):
            y[row0(k) : row1(k)] += self.up[i] @ y
        buf = np.concatenate([y, np.zeros(self.N)])
        for k in range(0, t.depth + 1):
            buf[self.N + row0(k) : self.N + row1(k)] = self.down[k] @ buf
        x = np.empty(self.N)
        x[t.perm] = buf[self.N :]
        return x


def factorize(A: sp.csr_matrix, tree: NDTree) -> NDFactors:
    """Numeric multifrontal factorisation with explicit pivot-block inverses."""
    N = A.shape[0]
    t = tree
    Ap = A[t.perm][:, t.perm].tocsr()
    Ap.sort_indices()
    updates: dict[tuple[int, int], tuple[np.ndarray, np.ndarray]] = {}
    Lr, Lc, Lv = [], [], []  # −L entries (row in B_t, col in I_t)
    Ur, Uc, Uv = [], [], []  # −U entries (row in I_t, col in B_t)
    Dr, Dc, Dv = [], [], []
    for k in range(t.depth, -1, -1):
        for n in range(2**k):
            i0, i1 = int(t.node_ptr[k][n]), int(t.node_ptr[k][n + 1])
            ni = i1 - i0
            B = t.bnd[k][n]
            nb = B.size
            if ni == 0:
                # nothing owned here: forward children's updates unchanged (merged)
                if k < t.depth:
                    idx = B
                    F = np.zeros((nb, nb))
                    for ch in (2 * n, 2 * n + 1):
                        cb, cu = updates.pop((k + 1, ch))
                        if cb.size:
                            p = np.searchsorted(idx, cb)
                            F[np.ix_(p, p)] += cu
                    updates[(k, n)] = (idx, F)
                else:
                    updates[(k, n)] = (B, np.zeros((nb, nb)))
                continue
            idx = np.concatenate([np.arange(i0, i1), B])
            nf = ni + nb
            F = np.zeros((nf, nf))
            rows = Ap[i0:i1]
            # original entries: A[I, I ∪ B] and A[B, I]
            coo = rows.tocoo()
            later = coo.col >= i0  # columns < i0 belong to deeper nodes: assembled there as A[B, I]
            coo = sp.coo_matrix((coo.data[later], (coo.row[later], coo.col[later])), shape=coo.shape)
            pos = np.searchsorted(idx[ni:], coo.col)
            inI = (coo.col >= i0) & (coo.col < i1)
            ok = inI.copy()
            if nb:
                posc = np.minimum(pos, nb - 1)
                inB = (~inI) & (idx[ni:][posc] == coo.col)
            else:
                posc = pos
                inB = np.zeros_like(inI)
            ok |= inB
            if not np.all(ok | (coo.data == 0.0)):
                raise RuntimeError("matrix entry outside the front: tree/boundary sets inconsistent")
            cc = np.where(inI, coo.col - i0, ni + posc)
            F[coo.row[ok], cc[ok]] += coo.data[ok]
            if nb:
                cols = Ap[:, i0:i1].tocsc()[B].tocoo()  # A[B, I]
                F[ni + cols.row, cols.col] += cols.data
            if k < t.depth:
                for ch in (2 * n, 2 * n + 1):
                    cb, cu = updates.pop((k + 1, ch))
                    if cb.size:
                        p = np.searchsorted(idx, cb)
                        if not np.array_equal(idx[p], cb):
                            raise RuntimeError("child boundary not contained in parent front")
                        F[np.ix_(p, p)] += cu
            F11 = F[:ni, :ni]
            Dinv = np.linalg.inv(F11)
            rr, cc2 = np.meshgrid(np.arange(i0, i1), np.arange(i0, i1), indexing="ij")
            Dr.append(rr.ravel()), Dc.append(cc2.ravel()), Dv.append(Dinv.ravel())
            if nb:
                F12, F21, F22 = F[:ni, ni:], F[ni:, :ni], F[ni:, ni:]
                Wt = F21 @ Dinv
                Vt = Dinv @ F12
                updates[(k, n)] = (B, F22 - Wt @ F12)
                rr, cc2 = np.meshgrid(B, np.arange(i0, i1), indexing="ij")
                Lr.append(rr.ravel()), Lc.append(cc2.ravel()), Lv.append(-Wt.ravel())
                rr, cc2 = np.meshgrid(np.arange(i0, i1), B, indexing="ij")
                Ur.append(rr.ravel()), Uc.append(cc2.ravel()), Uv.append(-Vt.ravel())
            else:
                updates[(k, n)] = (B, np.zeros((0, 0)))

    def cat(xs, dt):
        return np.concatenate(xs) if xs else np.zeros(0, dtype=dt)

    Lm = sp.csr_matrix((cat(Lv, float), (cat(Lr, np.int64), cat(Lc, np.int64))), shape=(N, N))
    Um = sp.csr_matrix((cat(Uv, float), (cat(Ur, np.int64), cat(Uc, np.int64))), shape=(N, N))
    Dm = sp.csr_matrix((cat(Dv, float), (cat(Dr, np.int64), cat(Dc, np.int64))), shape=(N, N))
    up, down = [], []
    for k in range(t.depth - 1, -1, -1):
        r0, r1 = int(t.node_ptr[k][0]), int(t.node_ptr[k][-1])
        up.append(Lm[r0:r1].tocsr())
    DU = sp.hstack([Dm, Um]).tocsr()
    for k in range(0, t.depth + 1):
        r0, r1 = int(t.node_ptr[k][0]), int(t.node_ptr[k][-1])
        down.append(DU[r0:r1].tocsr())
    nnz = int(Lm.nnz + Um.nnz + Dm.nnz)
    return NDFactors(t, N, up, down, nnz)


__all__ = ["NDTree", "NDFactors", "build_tree", "factorize"]
