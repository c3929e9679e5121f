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

TEST INFRASTRUCTURE (moved out of the package in round 5): this module is the readable numpy specification of the SYMBOLIC
analysis (tree, factor layout, elimination plan, partition, tiles, dependencies).  The product has exactly one implementation
of it, the library's ``csrc/fc_symbolic.hpp`` behind ``fc_setup_solver``; ``tests/test_symbolic_cabi.py`` compares the two
table by table, the golden-fixture generators use the tree's permutation as a fill-reducing ordering for SuperLU, and the
numeric host multifrontal the device factorisation is checked against is ``tests/support/nd_numeric.py``.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp


@dataclass
class NDTree:
    depth: int  # leaves live at this tree level; level k has 2**cum[k] nodes
    perm: np.ndarray  # new → old dof index
    iperm: np.ndarray  # old → new
    node_ptr: list[np.ndarray]  # per level k: (2**k + 1,) offsets into the *new* ordering
    level_ptr: np.ndarray  # (depth + 2,) new-index offsets of levels, deepest level FIRST
    bnd: list[list[np.ndarray]]  # per level k, per node: boundary dofs (new indices, sorted)
    cum: tuple = ()  # cum[k] = number of binary bisections above tree level k (cum[0] = 0)

    def nnodes(self, k: int) -> int:
        return 1 << self.cum[k]

    def children(self, k: int, n: int) -> range:
        """Node ids at level k+1 below node n of level k."""
        b = self.cum[k + 1] - self.cum[k]
        return range(n << b, (n + 1) << b)


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


def uniform_bits(depth: int, merge: int, top_bits: int = 0) -> list[int]:
    """Bisections fused per tree level, root first: ``depth`` bisections, ``merge`` at a time, below a 2**top_bits-ary root."""
    bits = [top_bits] if top_bits > 0 else []
    while sum(bits) < depth:
        bits.append(merge)
    return bits


def default_bits(nc: int, merge: int = 2, top_bits: int = 0) -> list[int]:
    """Shape of the default elimination tree (``csrc/fc_symbolic.hpp::default_bits`` is the same rule): leaves of about 12 cells,
    i.e. log2(nc / 12) bisections, to the nearest count the fused levels allow.  Small meshes on one GPU (≤ 16 000 cells: factors
    that stay in the Infinity Cache, every sweep launch on its ≈ 3.5 µs floor) fuse one bisection more into each of the two top
    levels — O1: [3, 3, 2, 2] instead of [2, 2, 2, 2, 2], two launches fewer for 11 % more factor bytes, + 3.8 % steps/s."""
    import os

    shape = os.environ.get("FC_ND_SHAPE")  # tuning aid (partitioned handles: the levels BELOW the 2**top_bits-ary rank level)
    if shape:
        bits = [int(b) for b in shape.split(",") if b.strip() and int(b) > 0]
        if bits:
            return ([top_bits] if top_bits > 0 else []) + bits
    levels = float(np.log2(max(nc, 1) / 12.0))
    if top_bits > 0:  # partitioned handles: rounded up, as ever
        return uniform_bits(max(merge + top_bits, int(np.ceil(levels))), merge, top_bits)
    d = merge * max(1, int(np.floor(levels / merge + 0.5)))
    if merge == 2 and nc <= 16000 and d >= 8:
        return [3, 3] + [2] * ((d - 6) // 2)
    return uniform_bits(d, merge, 0)


def build_tree(cell_dofs: np.ndarray, centroids: np.ndarray, N: int, depth: int, skip: np.ndarray | None = None,
               merge: int = 1, top_bits: int = 0, bits: list[int] | None = None) -> NDTree:
    """Element-based nested dissection (``bits``, bisections fused per tree level root first, overrides depth / merge / top_bits).

    A dof is owned by the deepest tree node whose cell set contains every cell touching it
    (leaf ⇒ subdomain interior; internal node ⇒ separator).  ``skip`` marks dofs that are
    decoupled identity rows (Dirichlet) — they are parked in the leaves.

    ``depth`` counts binary bisections.  ``merge = m`` fuses every m consecutive binary levels
    into one level of a 2**m-ary tree (the separators of those levels become one pivot block):
    fewer, fatter sweep stages on the device at the price of somewhat denser pivot inverses.
    ``top_bits = p`` makes the root 2**p-ary first (one sub-tree per GPU of a 2**p-rank run).
    """
    nc, nl = cell_dofs.shape
    bits = list(bits) if bits is not None else uniform_bits(depth, merge, top_bits)
    depth_bin = sum(bits)
    cum = np.concatenate([[0], np.cumsum(bits)]).astype(np.int64)
    K = len(bits)  # leaves at tree level K
    leaf = _bisect_cells(centroids, depth_bin)
    lo = np.full(N, np.iinfo(np.int64).max)
    hi = np.full(N, -1)
    flat = cell_dofs.reshape(-1).astype(np.int64)
    lf = np.repeat(leaf, nl)
    np.minimum.at(lo, flat, lf)
    np.maximum.at(hi, flat, lf)
    if np.any(hi < 0):
        raise ValueError("dof without any cell")
    if skip is not None:
        # decoupled (Dirichlet) rows are parked in a leaf — unless cells of different top-level
        # sub-trees (= different GPUs) touch them: those stay in the replicated root so that every
        # rank that needs their value has it
        same_top = (lo >> (depth_bin - top_bits)) == (hi >> (depth_bin - top_bits))
        hi = np.where(skip & same_top, lo, hi)
    x = lo ^ hi
    nb_bin = np.zeros(N, dtype=np.int64)
    nz = x > 0
    nb_bin[nz] = np.floor(np.log2(x[nz])).astype(np.int64) + 1
    beta = depth_bin - nb_bin  # owner depth in the binary tree
    level = np.searchsorted(cum, beta, side="right") - 1  # tree level: separators join their coarser group
    nbits = depth_bin - cum[level]
    prefix = lo >> nbits  # owner index within its level
    # ordering: deepest level first, then node, then original index (locality)
    key = np.lexsort((np.arange(N), prefix, -level))
    perm = key.astype(np.int64)
    iperm = np.empty(N, dtype=np.int64)
    iperm[perm] = np.arange(N)
    node_ptr: list[np.ndarray] = [None] * (K + 1)
    level_ptr = np.zeros(K + 2, dtype=np.int64)
    pos = 0
    for i, k in enumerate(range(K, -1, -1)):
        cnt = np.bincount(prefix[level == k], minlength=1 << int(cum[k]))
        node_ptr[k] = pos + np.r_[0, np.cumsum(cnt)]
        pos += int(cnt.sum())
        level_ptr[i + 1] = pos
    # boundary sets: dofs touching cells of subtree(t) that are owned by a proper ancestor
    bnd: list[list[np.ndarray]] = [None] * (K + 1)
    new_cell_dofs = iperm[cell_dofs.astype(np.int64)]
    lvl_new = level[perm]
    for k in range(K, -1, -1):
        sub = leaf >> int(depth_bin - cum[k])
        order = np.argsort(sub, kind="stable")
        nn_k = 1 << int(cum[k])
        starts = np.searchsorted(sub[order], np.arange(nn_k + 1))
        out = []
        for t in range(nn_k):
            cells = order[starts[t] : starts[t + 1]]
            dd = np.unique(new_cell_dofs[cells].reshape(-1))
            out.append(dd[lvl_new[dd] < k])
        bnd[k] = out
    tree = NDTree(K, perm, iperm, node_ptr, level_ptr, bnd, tuple(int(c) for c in cum))
    tree.leaf_of_cell = leaf  # binary leaf index of every cell (partitioning of the element loops)
    tree.depth_bin = depth_bin
    return tree


__all__ = ["NDTree", "build_tree"]


# ──────────────────────────────────────────────────────────────────────────────────────────
@dataclass
class BlockFactors:
    """Selected-inverse factors as dense blocks + per-row segment lists.

    Every factor row is a short list of *segments* ``(val_off, col, len)``: ``len`` consecutive
    fp64 values at ``vals[val_off:]`` multiplied with either ``len`` consecutive entries of the
    work buffer starting at ``col`` (``col >= 0``) or with the entries ``idx[-(col+1) + j]``
    (``col < 0``; the index list is shared by all rows of a tree node).  Values therefore cost
    8 B/nnz of HBM traffic and the column information a fraction of a byte, instead of CSR's
    12 B/nnz.

    Work buffer layout: ``buf = [y (N) | x (N)]``.
      stage kind 0 (up,   level k = depth-1 … 0):  y[r] += Σ segments   (segments hold −L rows)
      stage kind 1 (down, level k = 0 … depth):    x[r]  = Σ segments   ([D⁻¹ | −U] rows)
    """

    tree: NDTree
    N: int
    vals: np.ndarray  # float64
    idx: np.ndarray  # int32 shared column lists (entries index into buf)
    seg_val: np.ndarray  # int64 (nseg,)
    seg_col: np.ndarray  # int32 (nseg,)
    seg_len: np.ndarray  # int32 (nseg,)
    seg_ptr: np.ndarray  # int64 (total_rows + 1,) rows of all stages concatenated
    stage_row0: np.ndarray  # int32 first destination row of each stage (permuted numbering)
    stage_nrows: np.ndarray  # int32
    stage_kind: np.ndarray  # int32
    stage_begin: np.ndarray  # int64 offset of the stage's first row in seg_ptr
    nnz: int
    # per tree node with rows (for the LDS-tiled down-sweep kernel): level, node id, first row, rows,
    # boundary size, offset of its [D⁻¹ | −U] rows in ``vals`` (row stride ni+nb), offset of its index list
    nodes: np.ndarray | None = None  # (n_nodes, 7) int64: level, n, i0, ni, nb, val_off, idx_off
    #: multi-GPU: rows [lo, hi) of the root's pivot-block inverse are the only ones stored (val_off of the root = row lo)
    root_rows: tuple[int, int] | None = None


def root_row_block(tree: NDTree, rank: int, world: int) -> tuple[int, int]:
    """Rows [lo, hi) (permuted numbering) of the root's pivot-block inverse that rank ``rank`` applies — and therefore the
    only ones it stores: ceil(|root| / world) consecutive rows per rank."""
    lo, hi = int(tree.node_ptr[0][0]), int(tree.node_ptr[0][-1])
    if world <= 1:
        return lo, hi
    blk = -(-(hi - lo) // world)
    a = min(hi, lo + rank * blk)
    return a, min(hi, a + blk)


def rank_keeps(tree: NDTree, rank: int, world: int):
    """Predicate (level, node) → this rank stores and factorises the node: its own sub-tree of the ``world``-ary
    root, and the root itself — of whose pivot-block inverse it STORES only the rows it applies
    (``keep.root_rows``, :func:`root_row_block`; every rank still eliminates the whole root front)."""
    p = int(np.log2(world)) if world > 1 else 0

    def keep(k: int, n: int) -> bool:
        return world == 1 or k == 0 or (n >> (tree.cum[k] - p)) == rank

    if world > 1:
        keep.root_rows = root_row_block(tree, rank, world)
    return keep


def factorize_blocks(A: sp.csr_matrix | None, tree: NDTree, numeric: bool = False, keep=None) -> BlockFactors:
    """Layout of the block factors along ``tree``: segment lists, index lists, value offsets (``vals`` all zero, ``A`` is
    ignored and may be None).  The VALUES are computed on the device (:func:`factor_plan`, ``fc_refactor``); the numeric
    host multifrontal that used to live here is test infrastructure now (``tests/support/nd_numeric.py``).
    ``keep(level, node)``: nodes for which it is False get no storage at all — a rank of a multi-GPU run lays out its own
    sub-tree and the root (:func:`rank_keeps`), so its factor array is ~1/world of the whole.  ``keep.root_rows = (lo, hi)``:
    of the root's pivot-block inverse only the rows [lo, hi) get storage (at the root's value offset, row ``lo`` first);
    the other root rows get empty segments."""
    t = tree
    root_rows = getattr(keep, "root_rows", None)
    if numeric:
        raise ValueError("the product lays out structure only: the numeric host multifrontal is tests/support/nd_numeric.py")
    N = int(t.perm.size)
    vals_chunks: list[np.ndarray] = []
    vpos = 0
    idx_chunks: list[np.ndarray] = []
    ipos = 0
    up_rows, up_val, up_col, up_len = [], [], [], []
    dn_val = np.zeros((N, 2), dtype=np.int64)
    dn_col = np.zeros((N, 2), dtype=np.int32)
    dn_len = np.zeros((N, 2), dtype=np.int32)
    nnz = 0
    node_rows: list[tuple] = []
    for k in range(t.depth, -1, -1):
        for n in range(t.nnodes(k)):
            i0, i1 = int(t.node_ptr[k][n]), int(t.node_ptr[k][n + 1])
            ni = i1 - i0
            B = t.bnd[k][n]
            nb = B.size
            if keep is not None and not keep(k, n):
                continue
            if ni == 0:
                continue
            nf = ni + nb
            rows = np.arange(i0, i1)
            if nb:
                vals_chunks.append(np.zeros(ni * nf + nb * ni))  # [D^-1 | -U] (ni, ni+nb) row-major, then -L (nb, ni) row-major
                node_rows.append((k, n, i0, ni, nb, vpos, ipos))
                dn_val[rows, 0] = vpos + np.arange(ni) * nf
                dn_val[rows, 1] = vpos + np.arange(ni) * nf + ni
                vpos += ni * nf
                idx_chunks.append((N + B).astype(np.int32))
                dn_col[rows, 0], dn_len[rows, 0] = i0, ni
                dn_col[rows, 1], dn_len[rows, 1] = -(ipos + 1), nb
                ipos += nb
                up_rows.append(B)
                up_val.append(vpos + np.arange(nb, dtype=np.int64) * ni)
                up_col.append(np.full(nb, i0, dtype=np.int32))
                up_len.append(np.full(nb, ni, dtype=np.int32))
                vpos += nb * ni
                nnz += ni * ni + 2 * ni * nb
            else:
                a, b = (i0, i1) if (k != 0 or root_rows is None) else (max(i0, root_rows[0]), min(i1, root_rows[1]))
                nst = max(0, b - a)  # stored rows: all of them, or this rank's block of the root
                vals_chunks.append(np.zeros(nst * ni))
                node_rows.append((k, n, i0, ni, 0, vpos, 0))
                stored = (rows >= a) & (rows < b)
                dn_val[rows, 0] = np.where(stored, vpos + (rows - a) * ni, vpos)
                vpos += nst * ni
                dn_col[rows, 0], dn_len[rows, 0] = i0, np.where(stored, ni, 0)
                nnz += nst * ni
    vals = np.concatenate(vals_chunks) if vals_chunks else np.zeros(0)
    idx = np.concatenate(idx_chunks) if idx_chunks else np.zeros(0, dtype=np.int32)
    # up segments grouped by destination row (stable ⇒ fixed summation order: deeper nodes first)
    if up_rows:
        ur = np.concatenate(up_rows)
        uo = np.argsort(ur, kind="stable")
        ur = ur[uo]
        uv, uc, ul = np.concatenate(up_val)[uo], np.concatenate(up_col)[uo], np.concatenate(up_len)[uo]
    else:
        ur = np.zeros(0, dtype=np.int64)
        uv, uc, ul = np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32)
    up_cnt = np.bincount(ur, minlength=N)
    up_start = np.concatenate([[0], np.cumsum(up_cnt)])
    seg_val, seg_col, seg_len, seg_cnt = [], [], [], []
    row0, nrows, kind = [], [], []
    for k in range(t.depth - 1, -1, -1):  # up stages
        r0, r1 = int(t.node_ptr[k][0]), int(t.node_ptr[k][-1])
        a, b = int(up_start[r0]), int(up_start[r1])
        seg_val.append(uv[a:b]), seg_col.append(uc[a:b]), seg_len.append(ul[a:b])
        seg_cnt.append(up_cnt[r0:r1])
        row0.append(r0), nrows.append(r1 - r0), kind.append(0)
    for k in range(0, t.depth + 1):  # down stages
        r0, r1 = int(t.node_ptr[k][0]), int(t.node_ptr[k][-1])
        has2 = dn_len[r0:r1, 1] > 0
        cnt = 1 + has2.astype(np.int64)
        m = np.ones((r1 - r0, 2), dtype=bool)
        m[:, 1] = has2
        seg_val.append(dn_val[r0:r1][m]), seg_col.append(dn_col[r0:r1][m]), seg_len.append(dn_len[r0:r1][m])
        seg_cnt.append(cnt)
        row0.append(r0), nrows.append(r1 - r0), kind.append(1)
    cnt_all = np.concatenate(seg_cnt) if seg_cnt else np.zeros(0, np.int64)
    seg_ptr = np.concatenate([[0], np.cumsum(cnt_all)]).astype(np.int64)
    nrows = np.array(nrows, dtype=np.int32)
    begin = np.concatenate([[0], np.cumsum(nrows)[:-1]]).astype(np.int64)
    return BlockFactors(
        tree=t, N=N, vals=np.ascontiguousarray(vals), idx=np.ascontiguousarray(idx, dtype=np.int32),
        seg_val=np.ascontiguousarray(np.concatenate(seg_val), dtype=np.int64),
        seg_col=np.ascontiguousarray(np.concatenate(seg_col), dtype=np.int32),
        seg_len=np.ascontiguousarray(np.concatenate(seg_len), dtype=np.int32),
        seg_ptr=seg_ptr, stage_row0=np.array(row0, dtype=np.int32), stage_nrows=nrows,
        stage_kind=np.array(kind, dtype=np.int32), stage_begin=begin, nnz=int(nnz),
        nodes=np.array(node_rows, dtype=np.int64).reshape(-1, 7), root_rows=root_rows,
    )


@dataclass
class FactorPlan:
    """Everything the device needs to redo the NUMERIC factorisation on its own (``fc_refactor``).

    Fronts of all tree nodes live in one buffer (row-major, ``nf = ni + nb`` square).  ``nodes`` rows, in
    elimination order (deepest level first): level, front offset, nf, ni, value offset of the node's
    ``[D⁻¹ | −U]`` rows in the factor array (−1: empty pivot block), parent row (−1: root), child slot.
    Matrix entries are scattered with (``a_src`` → CSR value index in the ORIGINAL numbering,
    ``a_dst`` → offset in the front buffer), grouped per level by ``a_ptr``; a child's update block is
    added into its parent through the position list ``ext_p[ext_off : ext_off + nb_child]``.
    ``ap_src`` refreshes the permuted matrix of the residual monitor."""

    nodes: np.ndarray  # (n_nodes, 7) int64
    level_ptr: np.ndarray  # (n_levels + 1,) node ranges per level, deepest level first
    front_size: int
    a_src: np.ndarray
    a_dst: np.ndarray
    a_ptr: np.ndarray  # (n_levels + 1,)
    ext_off: np.ndarray  # (n_nodes,) int64, −1 for nodes without an update block / parent
    ext_p: np.ndarray  # int32
    ap_src: np.ndarray  # (nnz of the permuted matrix,) int64
    max_slots: int
    node_i0: np.ndarray | None = None  # (n_nodes,) first permuted dof of every plan node
    root_rows: tuple[int, int] | None = None  # multi-GPU: stored rows of the root's pivot-block inverse (fc_set_root_rows)


def factor_plan(fac: BlockFactors, indptr: np.ndarray, indices: np.ndarray, skip: np.ndarray | None = None, keep=None) -> FactorPlan:
    """Symbolic side of the device factorisation for the CSR pattern (``indptr``, ``indices``; original
    numbering) — pure index work, done once per (tree, pattern).  ``skip`` marks the decoupled
    (Dirichlet) dofs: their off-diagonal entries are structural zeros after the symmetric elimination and
    are left out of the fronts.  ``keep(level, node)``: as in :func:`factorize_blocks` — only the kept nodes get a
    front; the root's front then holds this rank's share only (its sub-tree's Schur complement; the matrix
    entries of the root front are scattered by the lead rank) and is summed over the ranks by ``fc_refactor``."""
    t = fac.tree
    N = fac.N
    nnz = int(indptr[-1])
    # permuted pattern carrying the original value index (+1 so that no entry is an explicit zero)
    tag = sp.csr_matrix((np.arange(1, nnz + 1, dtype=np.float64), indices, indptr), shape=(N, N))
    Ap = tag[t.perm][:, t.perm].tocsr()
    Ap.sort_indices()
    ap_src = np.round(Ap.data).astype(np.int64) - 1
    coo = Ap.tocoo()
    r_, c_, src = coo.row.astype(np.int64), coo.col.astype(np.int64), ap_src
    if skip is not None:
        sk = np.asarray(skip, dtype=bool)[t.perm]
        coupled = ~((sk[r_] | sk[c_]) & (r_ != c_))
        r_, c_, src = r_[coupled], c_[coupled], src[coupled]
    K = t.depth
    # all tree nodes in elimination order
    lv, nn_, i0s, nis, nbs = [], [], [], [], []
    for k in range(K, -1, -1):
        for n in range(t.nnodes(k)):
            if keep is not None and not keep(k, n):
                continue
            a, b = int(t.node_ptr[k][n]), int(t.node_ptr[k][n + 1])
            lv.append(k), nn_.append(n), i0s.append(a), nis.append(b - a), nbs.append(int(t.bnd[k][n].size))
    lv, nn_, i0s, nis, nbs = (np.array(x, dtype=np.int64) for x in (lv, nn_, i0s, nis, nbs))
    nfs = nis + nbs
    gid = {(int(k), int(n)): g for g, (k, n) in enumerate(zip(lv, nn_))}
    front_off = np.concatenate([[0], np.cumsum(nfs * nfs)])
    voff = np.full(lv.size, -1, dtype=np.int64)
    for k, n, _i0, _ni, _nb, vo, _io in fac.nodes:
        voff[gid[(int(k), int(n))]] = vo
    # owner (node with a non-empty pivot block) of every permuted dof
    owner = np.full(N, -1, dtype=np.int64)
    for g in np.nonzero(nis > 0)[0]:
        owner[i0s[g] : i0s[g] + nis[g]] = g
    own = owner[np.minimum(r_, c_)]
    mine = own >= 0  # entries of fronts that another rank builds are not ours
    r_, c_, src, own = r_[mine], c_[mine], src[mine], own[mine]
    order = np.argsort(own, kind="stable")
    r_, c_, src, own = r_[order], c_[order], src[order], own[order]
    beg = np.searchsorted(own, np.arange(lv.size + 1))
    a_dst = np.empty(r_.size, dtype=np.int64)
    for g in range(lv.size):
        a, b = int(beg[g]), int(beg[g + 1])
        if a == b:
            continue
        i0, ni, nb, nf = int(i0s[g]), int(nis[g]), int(nbs[g]), int(nfs[g])
        B = t.bnd[int(lv[g])][int(nn_[g])]
        er, ec = r_[a:b], c_[a:b]
        rin, cin = er < i0 + ni, ec < i0 + ni
        pr = np.where(rin, er - i0, ni + np.searchsorted(B, er))
        pc = np.where(cin, ec - i0, ni + np.searchsorted(B, ec))
        if nb:
            bad = (~rin & (B[np.clip(pr - ni, 0, nb - 1)] != er)) | (~cin & (B[np.clip(pc - ni, 0, nb - 1)] != ec))
        else:
            bad = ~rin | ~cin
        if np.any(bad):
            raise RuntimeError("matrix entry outside the front: tree/boundary sets inconsistent")
        a_dst[a:b] = front_off[g] + pr * nf + pc
    # per level ranges (nodes are level-sorted, deepest first; `own` ascending ⇒ entries too)
    level_ptr = np.searchsorted(-lv, -np.arange(K, -2, -1), side="left").astype(np.int64)
    a_ptr = beg[level_ptr]
    # extend-add lists
    parent = np.full(lv.size, -1, dtype=np.int64)
    slot = np.zeros(lv.size, dtype=np.int64)
    ext_off = np.full(lv.size, -1, dtype=np.int64)
    ext_chunks = []
    pos = 0
    max_slots = 1
    for g in range(lv.size):
        k, n = int(lv[g]), int(nn_[g])
        if k == K:
            continue
        ch = list(t.children(k, n))
        max_slots = max(max_slots, len(ch))
        idxs = np.concatenate([np.arange(i0s[g], i0s[g] + nis[g]), t.bnd[k][n]])
        for c, chn in enumerate(ch):
            gc = gid.get((k + 1, int(chn)))
            if gc is None:
                continue  # another rank's sub-tree
            cb = t.bnd[k + 1][int(chn)]
            if cb.size == 0:
                continue
            pp = np.searchsorted(idxs, cb)
            if np.any(idxs[np.clip(pp, 0, idxs.size - 1)] != cb):
                raise RuntimeError("child boundary outside the parent front")
            parent[gc], slot[gc], ext_off[gc] = g, c, pos
            ext_chunks.append(pp.astype(np.int32))
            pos += cb.size
    ext_p = np.concatenate(ext_chunks) if ext_chunks else np.zeros(1, dtype=np.int32)
    nodes = np.stack([lv, front_off[:-1], nfs, nis, voff, parent, slot], axis=1).astype(np.int64)
    return FactorPlan(nodes=np.ascontiguousarray(nodes), level_ptr=level_ptr, front_size=int(front_off[-1]),
                      a_src=np.ascontiguousarray(src), a_dst=np.ascontiguousarray(a_dst), a_ptr=np.ascontiguousarray(a_ptr, dtype=np.int64),
                      ext_off=ext_off, ext_p=np.ascontiguousarray(ext_p), ap_src=np.ascontiguousarray(ap_src), max_slots=max_slots,
                      node_i0=i0s.copy(), root_rows=fac.root_rows)


def front_diagonal_slot(plan: FactorPlan, tree: NDTree, dof: int) -> int:
    """Offset, in the front buffer, of the diagonal entry of ``dof`` (original numbering) in the front of
    the plan node that eliminates it; −1 when that node is not in this rank's plan."""
    ip = int(tree.iperm[dof])
    i0, ni = plan.node_i0, plan.nodes[:, 3]
    hit = np.nonzero((i0 <= ip) & (ip < i0 + ni))[0]
    if hit.size == 0:
        return -1
    g = int(hit[0])
    fo, nf = int(plan.nodes[g, 1]), int(plan.nodes[g, 2])
    return fo + (ip - int(i0[g])) * (nf + 1)


def split_up_segments(fac: BlockFactors, maxlen: int) -> BlockFactors:
    """Cut the up-sweep segments into pieces of ≤ ``maxlen`` values (same values, same order).

    A sub-group of lanes covers ``maxlen`` values with one trip of loads; with every segment a single
    trip, the sub-groups of a row run side by side and the row is one memory round trip deep instead
    of one per (long) segment."""
    if maxlen <= 0:
        return fac
    nseg = fac.seg_len.size
    up = np.zeros(nseg, dtype=bool)
    for s in range(len(fac.stage_kind)):
        if fac.stage_kind[s] == 0:
            b, nr = int(fac.stage_begin[s]), int(fac.stage_nrows[s])
            up[int(fac.seg_ptr[b]) : int(fac.seg_ptr[b + nr])] = True
    pieces = np.where(up, np.maximum(1, -(-fac.seg_len.astype(np.int64) // maxlen)), 1)
    cp = np.concatenate([[0], np.cumsum(pieces)])
    rep = np.repeat(np.arange(nseg), pieces)
    within = np.arange(int(cp[-1])) - cp[:-1][rep]
    off = within * maxlen
    new_len = np.where(up[rep], np.minimum(maxlen, fac.seg_len[rep] - off), fac.seg_len[rep])
    from dataclasses import replace

    return replace(
        fac, seg_val=np.ascontiguousarray(fac.seg_val[rep] + off, dtype=np.int64),
        seg_col=np.ascontiguousarray(fac.seg_col[rep] + np.where(up[rep], off, 0), dtype=np.int32),
        seg_len=np.ascontiguousarray(new_len, dtype=np.int32), seg_ptr=np.ascontiguousarray(cp[fac.seg_ptr], dtype=np.int64),
    )


def down_blocks(fac: BlockFactors, rank: int = 0, world: int = 1, max_rows: int = 32, target_blocks: int = 1024,
                min_blocks: int = 512):
    """Workgroup tiles of the down-sweep stages for ``fc_solver_set_blocks``.

    Returns per-stage arrays (begin, count, lanes-per-row) aligned with the stage list of
    :func:`partition` (up stages get count 0) and the block arrays.  A block is ≤ ``max_rows``
    consecutive rows of one tree node; lanes per row follow the row width (4 × lanes ≥ width when
    possible: the whole row is then one trip of loads) and rows per block shrink for stages with few
    rows so that a launch still has ≳ ``target_blocks`` workgroups.  With ``world > 1`` only the
    rank's nodes (and the root) are tiled.  A stage is tiled when that gives ≥ ``min_blocks``
    workgroups even at the smallest tile; a stage of few, long rows (the root of a small mesh) stays
    on the segment kernel, which puts a whole workgroup on each row (measured on the 56 k-dof
    cylinder mesh, per-launch medians: leaves 10.1 → 7.0 µs, level 4: 8.6 → 6.6, level 3:
    7.5 → 6.3, level 2: 8.4 → 6.9, level 1: 9.1 → 8.4, but root 5.6 → 7.1)."""
    t = fac.tree
    p = int(np.log2(world)) if world > 1 else 0
    nodes = fac.nodes
    nst = len(fac.stage_kind)
    begin = np.zeros(nst, dtype=np.int64)
    count = np.zeros(nst, dtype=np.int32)
    lpr = np.full(nst, 64, dtype=np.int32)
    cols: list[list[int]] = [[] for _ in range(7)]
    nblk = 0
    for s in range(nst):
        if fac.stage_kind[s] != 1:
            begin[s] = nblk
            continue
        k = s - t.depth
        sel = nodes[nodes[:, 0] == k]
        if world > 1 and k >= 1:
            sh = t.cum[k] - p
            sel = sel[(sel[:, 1] >> sh) == rank]
        sel = sel[np.argsort(sel[:, 2])]
        lo, hi = 0, np.iinfo(np.int64).max  # rows of the node this rank sweeps (the root is split in row blocks)
        if world > 1 and k == 0:
            r_lo, r_hi = int(t.node_ptr[0][0]), int(t.node_ptr[0][-1])
            bs = -(-(r_hi - r_lo) // world)
            lo = min(r_hi, r_lo + rank * bs)
            hi = min(r_hi, lo + bs)
        nrows_of = np.clip(np.minimum(sel[:, 2] + sel[:, 3], hi) - np.maximum(sel[:, 2], lo), 0, None)
        values = float(((sel[:, 3] + sel[:, 4]) * nrows_of).sum())
        rows = int(nrows_of.sum())
        wd_mean = values / max(rows, 1)
        lpr[s] = 16 if wd_mean <= 64 else (32 if wd_mean <= 128 else 64)
        slots = 256 // int(lpr[s])
        if rows // slots < min_blocks:
            begin[s] = nblk
            continue
        rc = max_rows
        while rc > slots and rows // rc < target_blocks:
            rc //= 2
        begin[s] = nblk
        for _, n, i0, ni, nb, voff, ioff in sel:
            wd = ni + nb
            first, last = max(0, lo - int(i0)), min(int(ni), hi - int(i0))
            skipped = (fac.root_rows[0] - int(i0)) if (k == 0 and fac.root_rows is not None) else 0  # root rows before the stored block
            for r0 in range(first, last, rc):
                nr = min(rc, last - r0)
                for lst, v in zip(cols, (voff + (r0 - skipped) * wd, i0 + r0, nr, i0, ni, ioff, nb)):
                    lst.append(int(v))
                nblk += 1
        count[s] = nblk - begin[s]
    arr = lambda i, dt: np.ascontiguousarray(np.array(cols[i], dtype=dt))  # noqa: E731
    return begin, count, lpr, arr(0, np.int64), arr(1, np.int32), arr(2, np.int32), arr(3, np.int32), arr(4, np.int32), arr(5, np.int32), arr(6, np.int32)


def schur_diagonal_scaling(A: sp.csr_matrix, nn: int) -> np.ndarray:
    """Diagonal stand-in for the inverse of the operator on rows whose pivot blocks are not kept (truncated factors):
    velocity rows 1 / A_ii; pressure rows the SIMPLE-type estimate of the Schur complement −D F⁻¹ G,
    1 / (−Σ_j D_ij G_ji / F_jj), for the saddle-point matrix [[F, G], [D, 0]] (W numbering, velocity dofs first)."""
    A = A.tocsr()
    nn2 = 2 * nn
    diag = A.diagonal()
    out = np.ones(A.shape[0])
    fd = np.where(diag[:nn2] != 0.0, diag[:nn2], 1.0)
    out[:nn2] = 1.0 / fd
    D, G = A[nn2:, :nn2], A[:nn2, nn2:]
    s = -np.asarray(D.multiply(G.T.tocsr()) @ (1.0 / fd)).reshape(-1)
    out[nn2:] = np.where(s != 0.0, 1.0 / np.where(s != 0.0, s, 1.0), 1.0)
    return out


__all__ += ["BlockFactors", "factorize_blocks", "rank_keeps", "schur_diagonal_scaling", "down_blocks", "split_up_segments", "FactorPlan", "factor_plan",
            "front_diagonal_slot"]


# ──────────────────────────────────────────────────────────────────────────────────────────
# Multi-GPU partition: one sub-tree of the root per rank, the root separator replicated.
# ──────────────────────────────────────────────────────────────────────────────────────────
@dataclass
class RankPartition:
    """What rank ``rank`` of ``world`` executes (permuted numbering unless stated).

    * rows of tree levels ≥ 1 that lie in the rank's sub-tree (contiguous ranges per level);
    * the root level: every rank sweeps *its* columns of the root's L rows into a partial
      right-hand side, an all-reduce sums the partials (exchange 1), every rank applies ITS block of
      rows of the root's D⁻¹ (rows ``[root_row0, root_row0 + root_nrows)`` of the root), and a second
      all-reduce of the zero-padded blocks assembles the root solution on every rank (exchange 2: an
      all-gather in effect) — no factor value is applied twice, and no halo exchange is needed afterwards.
    """

    rank: int
    world: int
    rank_of_dof: np.ndarray  # (N,) permuted numbering; -1 = root separator
    rowkind: np.ndarray  # (N,) ORIGINAL numbering: 0 other rank, 1 owned, 2 root (replicated)
    local_cells: np.ndarray  # cell ids assembled by this rank
    seg_ptr: np.ndarray
    seg_val: np.ndarray
    seg_col: np.ndarray
    seg_len: np.ndarray
    stage_row0: np.ndarray
    stage_nrows: np.ndarray
    stage_kind: np.ndarray
    stage_begin: np.ndarray
    ar_stage: int
    ar_row0: int
    ar_n: int
    ar2_stage: int = -1  # the root's down stage: its result x[ar_row0 .. +ar_n) is summed over the ranks as well
    root_row0: int = 0  # this rank's block of root rows (permuted numbering)
    root_nrows: int = 0


def partition(fac: BlockFactors, rank: int, world: int) -> RankPartition:
    t = fac.tree
    N = fac.N
    p = int(np.log2(world))
    if (1 << p) != world:
        raise ValueError("world size must be a power of two")
    if world > 1 and (len(t.cum) < 2 or t.cum[1] != p):
        raise ValueError("tree was not built with top_bits = log2(world)")
    rank_of = np.full(N, -1, dtype=np.int64)
    if world == 1:
        rank_of[:] = 0
    else:
        for k in range(1, t.depth + 1):
            sh = t.cum[k] - p
            for n in range(t.nnodes(k)):
                rank_of[int(t.node_ptr[k][n]) : int(t.node_ptr[k][n + 1])] = n >> sh
    rowkind = np.zeros(N, dtype=np.uint8)
    kind_new = np.where(rank_of < 0, 2, np.where(rank_of == rank, 1, 0)).astype(np.uint8)
    rowkind[t.perm] = kind_new
    leaf = t.leaf_of_cell
    local_cells = np.nonzero((leaf >> (t.depth_bin - p)) == rank)[0].astype(np.int32) if world > 1 else np.arange(leaf.size, dtype=np.int32)
    seg_ptr = [np.zeros(1, dtype=np.int64)]
    sv, sc, sl = [], [], []
    row0, nrows, kinds = [], [], []
    ar_stage = ar2_stage = -1
    nseg = 0
    nstages = len(fac.stage_kind)
    root_lo, root_hi = int(t.node_ptr[0][0]), int(t.node_ptr[0][-1])
    blk = -(-(root_hi - root_lo) // world)
    my_lo = min(root_hi, root_lo + rank * blk)
    my_hi = min(root_hi, my_lo + blk)
    for s in range(nstages):
        k = (t.depth - 1 - s) if s < t.depth else (s - t.depth)  # level of this stage
        g0 = int(fac.stage_begin[s])
        gr0, gn = int(fac.stage_row0[s]), int(fac.stage_nrows[s])
        ptr = fac.seg_ptr[g0 : g0 + gn + 1]
        if k >= 1 and world > 1:
            sh = t.cum[k] - p
            n0, n1 = rank << sh, (rank + 1) << sh
            r0, r1 = int(t.node_ptr[k][n0]), int(t.node_ptr[k][n1])
            a, b = r0 - gr0, r1 - gr0
            q0, q1 = int(ptr[a]), int(ptr[b])
            cnt = np.diff(ptr[a : b + 1])
            sel = slice(q0, q1)
            row0.append(r0), nrows.append(r1 - r0)
            sv.append(fac.seg_val[sel]), sc.append(fac.seg_col[sel]), sl.append(fac.seg_len[sel])
        elif k == 0 and world > 1 and fac.stage_kind[s] == 0:
            # root up-sweep: keep only the segments that read this rank's part of y
            q0, q1 = int(ptr[0]), int(ptr[-1])
            cols = fac.seg_col[q0:q1]
            keep = rank_of[cols] == rank
            row_of_seg = np.repeat(np.arange(gn), np.diff(ptr))
            cnt = np.bincount(row_of_seg[keep], minlength=gn)
            row0.append(gr0), nrows.append(gn)
            sv.append(fac.seg_val[q0:q1][keep]), sc.append(cols[keep]), sl.append(fac.seg_len[q0:q1][keep])
            ar_stage = len(kinds)
        elif k == 0 and world > 1 and fac.stage_kind[s] == 1:
            # root down-sweep: this rank's block of rows only (the blocks are assembled by the second exchange)
            a, b = my_lo - gr0, my_hi - gr0
            q0, q1 = int(ptr[a]), int(ptr[b])
            cnt = np.diff(ptr[a : b + 1])
            row0.append(my_lo), nrows.append(my_hi - my_lo)
            sv.append(fac.seg_val[q0:q1]), sc.append(fac.seg_col[q0:q1]), sl.append(fac.seg_len[q0:q1])
            ar2_stage = len(kinds)
        else:
            q0, q1 = int(ptr[0]), int(ptr[-1])
            cnt = np.diff(ptr)
            row0.append(gr0), nrows.append(gn)
            sv.append(fac.seg_val[q0:q1]), sc.append(fac.seg_col[q0:q1]), sl.append(fac.seg_len[q0:q1])
        kinds.append(int(fac.stage_kind[s]))
        seg_ptr.append(nseg + np.cumsum(cnt))
        nseg += int(cnt.sum())
    nrows = np.array(nrows, dtype=np.int32)
    root0, root1 = int(t.node_ptr[0][0]), int(t.node_ptr[0][-1])
    return RankPartition(
        rank=rank, world=world, rank_of_dof=rank_of, rowkind=rowkind, local_cells=local_cells,
        seg_ptr=np.concatenate(seg_ptr).astype(np.int64),
        seg_val=np.ascontiguousarray(np.concatenate(sv), dtype=np.int64),
        seg_col=np.ascontiguousarray(np.concatenate(sc), dtype=np.int32),
        seg_len=np.ascontiguousarray(np.concatenate(sl), dtype=np.int32),
        stage_row0=np.array(row0, dtype=np.int32), stage_nrows=nrows, stage_kind=np.array(kinds, dtype=np.int32),
        stage_begin=np.concatenate([[0], np.cumsum(nrows)[:-1]]).astype(np.int64),
        ar_stage=ar_stage if world > 1 else -1, ar_row0=root0, ar_n=(root1 - root0) if world > 1 else 0,
        ar2_stage=ar2_stage if world > 1 else -1, root_row0=my_lo if world > 1 else root0,
        root_nrows=(my_hi - my_lo) if world > 1 else root1 - root0,
    )


__all__ += ["RankPartition", "partition"]


def tree_of(dev) -> "NDTree":
    """The elimination tree of a :class:`flowcontrol_amd.device.DeviceSolver` whose solver has been set up, rebuilt here from the
    shape the library reports (``fc_get_tree_info``); its permutation must be the handle's."""
    bits = dev.tree_info()["bits"]
    top = int(np.log2(dev.world)) if dev.world > 1 else 0
    skip = np.zeros(dev.N, dtype=bool)
    skip[dev.bc_dofs] = True
    t = build_tree(dev.th.cell_dofs, dev.th.mesh.cell_centroids(), dev.N, sum(bits), skip, top_bits=top, bits=bits)
    assert np.array_equal(t.perm, dev.perm), "the numpy specification and the library disagree on the permutation"
    return t
