"""Numeric host multifrontal of the nested-dissection selected inverse — TEST INFRASTRUCTURE.

The product (``flowcontrol_amd``) computes factor values on the device only (``fc_refactor``); this module is the
independent numpy implementation the tests check it against (moved out of ``flowcontrol_amd/ndsolver.py`` in round 3):

  factorize / NDFactors        level-wise CSR factors straight from a matrix (dense fronts, numpy inverses)
  factorize_blocks(A, tree)    the device's block layout filled with numbers: layout (ndsolver) + plan + factorize_with_plan
  factorize_with_plan          host replay of fc_refactor's order of operations (per-rank plans, root all-reduce)
  block_solve / to_csr_stages  host apply of block factors
  solve_partitioned_reference  what one rank's device does in a partitioned solve (gloo tests)
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp

from tests.support import ndsolver
from tests.support.ndsolver import BlockFactors, FactorPlan, NDTree, RankPartition


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
        for n in range(t.nnodes(k)):
            i0, i1 = int(t.node_ptr[k][n]), int(t.node_ptr[k][n + 1])
            ni = i1 - i0
            B = t.bnd[k][n]
            nb = B.size
            if ni == 0:
                # nothing owned here: forward children's updates unchanged (merged)
                if k < t.depth:
                    idx = B
                    F = np.zeros((nb, nb))
                    for ch in t.children(k, n):
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
                for ch in t.children(k, n):
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


# ──────────────────────────────────────────────────────────────────────────────────────────
# Block ("segment list") factors: what the device actually consumes.


def factorize_with_plan(plan: FactorPlan, fac: BlockFactors, values: np.ndarray, lead: bool = True, allreduce=None) -> np.ndarray:
    """Host replay of exactly what ``fc_refactor`` does on the device (same order of operations): the
    factor values for the CSR ``values`` (original numbering).  Test reference, not a product path.

    Per-rank plans (``keep=`` of :func:`factor_plan`): ``lead`` — this rank scatters the matrix entries of the root
    front; ``allreduce(array)`` sums the root front over the ranks before the root is eliminated."""
    F = np.zeros(plan.front_size)
    vals = np.zeros(fac.vals.size)
    nodes = plan.nodes
    n_lower = int(plan.a_ptr[-2]) if allreduce is not None else int(plan.a_ptr[-1])  # entries below the root level
    np.add.at(F, plan.a_dst[:n_lower], values[plan.a_src[:n_lower]])
    if allreduce is not None and lead:
        np.add.at(F, plan.a_dst[n_lower:], values[plan.a_src[n_lower:]])
    nlev = plan.level_ptr.size - 1
    for li in range(nlev):
        g0, g1 = int(plan.level_ptr[li]), int(plan.level_ptr[li + 1])
        # children of this level's nodes were finished in the previous round: add their update blocks
        if li > 0:
            c0, c1 = int(plan.level_ptr[li - 1]), int(plan.level_ptr[li])
            for s in range(plan.max_slots):
                for gc in range(c0, c1):
                    if plan.ext_off[gc] < 0 or nodes[gc, 6] != s:
                        continue
                    _, fo, nf, ni, _, par, _ = nodes[gc]
                    nbc = nf - ni
                    S = F[fo : fo + nf * nf].reshape(nf, nf)[ni:, ni:]
                    pp = plan.ext_p[plan.ext_off[gc] : plan.ext_off[gc] + nbc]
                    _, pfo, pnf = nodes[par, 0], nodes[par, 1], nodes[par, 2]
                    P = F[pfo : pfo + pnf * pnf].reshape(pnf, pnf)
                    P[np.ix_(pp, pp)] += S
        if allreduce is not None and li == nlev - 1:
            _, fo, nf = nodes[g0, 0], nodes[g0, 1], nodes[g0, 2]
            root = F[fo : fo + nf * nf]
            allreduce(root)  # every rank's sub-tree contributes its Schur complement, the lead the matrix entries
        for g in range(g0, g1):
            _, fo, nf, ni, vo, _, _ = nodes[g]
            if ni == 0:
                continue
            Fm = F[fo : fo + nf * nf].reshape(nf, nf)
            Dinv = np.linalg.inv(Fm[:ni, :ni])
            nb = nf - ni
            if nb:
                mVt = -(Dinv @ Fm[:ni, ni:])
                mWt = -(Fm[ni:, :ni] @ Dinv)
                Fm[ni:, ni:] += mWt @ Fm[:ni, ni:]
                vals[vo : vo + ni * nf] = np.hstack([Dinv, mVt]).ravel()
                vals[vo + ni * nf : vo + ni * nf + nb * ni] = mWt.ravel()
            elif plan.root_rows is not None and li == nlev - 1:
                a, b = plan.root_rows[0] - int(plan.node_i0[g]), plan.root_rows[1] - int(plan.node_i0[g])
                vals[vo : vo + (b - a) * ni] = Dinv[a:b].ravel()  # this rank's rows of the root's pivot-block inverse
            else:
                vals[vo : vo + ni * ni] = Dinv.ravel()
    return vals


def factorize_blocks(A: sp.csr_matrix, tree: NDTree) -> BlockFactors:
    """Block factors WITH values for ``A`` (original numbering): the layout of ``ndsolver.factorize_blocks`` filled by the
    host replay of the device factorisation."""
    fac = ndsolver.factorize_blocks(None, tree)
    A = sp.csr_matrix(A, copy=True)
    A.eliminate_zeros()  # rows / columns of eliminated Dirichlet dofs hold structural zeros outside their (leaf) fronts
    A.sort_indices()
    plan = ndsolver.factor_plan(fac, A.indptr.astype(np.int64), A.indices.astype(np.int64))
    fac.vals = factorize_with_plan(plan, fac, np.asarray(A.data, dtype=np.float64))
    return fac


def to_csr_stages(self: BlockFactors):
    """(up, down) lists of scipy CSR matrices — host reference used by the CPU tests."""
    mats = []
    for s in range(len(self.stage_kind)):
        r0 = int(self.stage_begin[s])
        nr = int(self.stage_nrows[s])
        rows, cols, vals = [], [], []
        for r in range(nr):
            for q in range(int(self.seg_ptr[r0 + r]), int(self.seg_ptr[r0 + r + 1])):
                n = int(self.seg_len[q])
                c = int(self.seg_col[q])
                cc = np.arange(c, c + n) if c >= 0 else self.idx[-(c + 1) : -(c + 1) + n]
                rows.append(np.full(n, r))
                cols.append(cc)
                vals.append(self.vals[int(self.seg_val[q]) : int(self.seg_val[q]) + n])
        if rows:
            M = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nr, 2 * self.N))
        else:
            M = sp.csr_matrix((nr, 2 * self.N))
        mats.append(M)
    return mats

def block_solve(self: BlockFactors, b: np.ndarray) -> np.ndarray:
    """Host reference of the device apply (small meshes only: builds CSR stages)."""
    t = self.tree
    buf = np.concatenate([b[t.perm].astype(np.float64), np.zeros(self.N)])
    for s, M in enumerate(to_csr_stages(self)):
        r0, nr = int(self.stage_row0[s]), int(self.stage_nrows[s])
        if self.stage_kind[s] == 0:
            buf[r0 : r0 + nr] += M @ buf
        else:
            buf[self.N + r0 : self.N + r0 + nr] = M @ buf
    x = np.empty(self.N)
    x[t.perm] = buf[self.N :]
    return x


def solve_partitioned_reference(fac: BlockFactors, part: RankPartition, b_local_perm: np.ndarray, allreduce) -> np.ndarray:
    """Host emulation of what one rank's device does (CPU tests with gloo): ``b_local_perm`` holds
    this rank's share of the permuted right-hand side (owned rows + its partial of the root rows);
    ``allreduce(array)`` sums an array over the ranks in place.  Returns the x-half of the buffer
    (valid on owned and root rows)."""
    N = fac.N
    buf = np.concatenate([b_local_perm.astype(np.float64), np.zeros(N)])
    for s in range(len(part.stage_kind)):
        g0, nr, r0 = int(part.stage_begin[s]), int(part.stage_nrows[s]), int(part.stage_row0[s])
        acc = np.zeros(nr)
        for r in range(nr):
            for q in range(int(part.seg_ptr[g0 + r]), int(part.seg_ptr[g0 + r + 1])):
                n, c, vo = int(part.seg_len[q]), int(part.seg_col[q]), int(part.seg_val[q])
                xs = buf[c : c + n] if c >= 0 else buf[fac.idx[-(c + 1) : -(c + 1) + n]]
                acc[r] += fac.vals[vo : vo + n] @ xs
        if part.stage_kind[s] == 0:
            buf[r0 : r0 + nr] += acc
        else:
            if s == part.ar2_stage:
                buf[N + part.ar_row0 : N + part.ar_row0 + part.ar_n] = 0.0  # the other ranks' blocks
            buf[N + r0 : N + r0 + nr] = acc
        if s == part.ar_stage and part.ar_n > 0:
            seg = buf[part.ar_row0 : part.ar_row0 + part.ar_n].copy()
            allreduce(seg)
            buf[part.ar_row0 : part.ar_row0 + part.ar_n] = seg
        if s == part.ar2_stage and part.ar_n > 0:
            seg = buf[N + part.ar_row0 : N + part.ar_row0 + part.ar_n].copy()
            allreduce(seg)
            buf[N + part.ar_row0 : N + part.ar_row0 + part.ar_n] = seg
    return buf[N:]
