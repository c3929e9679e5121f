// Symbolic phase of the nested-dissection selected-inverse solver, host C++ (no HIP): elimination tree, segment
// tables, workgroup tiles, factorisation plan, task dependencies — what MUMPS' analysis phase is to the reference
// (LUSolver.set_operator, src/flowcontrol/flowsolver.py:697,812-814).  fc_setup_solver (fc_hip.hip) runs it from the
// mesh the handle already holds, so a caller of the C ABI needs no Python.  flowcontrol_amd/ndsolver.py is the
// readable specification of every routine here and the tests compare the two (tests/test_symbolic_cabi.py).
#pragma once
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <map>
#include <numeric>
#include <stdexcept>
#include <exception>
#include <thread>
#include <vector>

namespace fcsym {

// index work over independent rows / nodes on a few host threads (FC_SYM_THREADS, default min(16, cores))
template <class F>
inline void parallel_for(int64_t n, F&& fn) {
  int nt = 0;
  if (const char* e = getenv("FC_SYM_THREADS")) nt = atoi(e);
  if (nt <= 0) nt = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
  nt = (int)std::min<int64_t>(nt, std::max<int64_t>(1, n / 1024));
  if (nt <= 1) {
    for (int64_t i = 0; i < n; ++i) fn(i);
    return;
  }
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err((size_t)nt);
  for (int w = 0; w < nt; ++w)
    th.emplace_back([&, w] {
      try {
        const int64_t a = n * w / nt, b = n * (w + 1) / nt;
        for (int64_t i = a; i < b; ++i) fn(i);
      } catch (...) {
        err[(size_t)w] = std::current_exception();
      }
    });
  for (auto& t : th) t.join();
  for (auto& e : err)
    if (e) std::rethrow_exception(e);
}


struct Tree {
  int depth = 0;       // tree levels 0 .. depth (leaves at `depth`)
  int depth_bin = 0;   // binary bisections
  std::vector<int> cum;  // cum[k] = bisections above level k
  std::vector<int> perm, iperm;
  std::vector<std::vector<int64_t>> node_ptr;       // per level: nnodes + 1 offsets (permuted numbering)
  std::vector<std::vector<std::vector<int>>> bnd;   // per level, per node: boundary dofs (permuted, sorted)
  std::vector<int> leaf_of_cell;
  int nnodes(int k) const { return 1 << cum[k]; }
  int child_bits(int k) const { return cum[k + 1] - cum[k]; }
};

// leaf index in [0, 2^depth) per cell by recursive coordinate-median bisection (ndsolver._bisect_cells)
inline std::vector<int> bisect_cells(const std::vector<double>& cent /* [nc][2] */, int nc, int depth) {
  std::vector<int> leaf(nc, 0);
  std::vector<std::vector<int>> groups(1);
  groups[0].resize(nc);
  std::iota(groups[0].begin(), groups[0].end(), 0);
  for (int level = 0; level < depth; ++level) {
    std::vector<std::vector<int>> next;
    next.reserve(groups.size() * 2);
    for (auto& g : groups) {
      if (g.empty()) {
        next.emplace_back();
        next.emplace_back();
        continue;
      }
      double mn[2] = {1e300, 1e300}, mx[2] = {-1e300, -1e300};
      for (int c : g)
        for (int d = 0; d < 2; ++d) {
          mn[d] = std::min(mn[d], cent[2 * (size_t)c + d]);
          mx[d] = std::max(mx[d], cent[2 * (size_t)c + d]);
        }
      const int ax = (mx[1] - mn[1]) > (mx[0] - mn[0]) ? 1 : 0;  // numpy argmax: first maximum
      std::vector<int> order(g.size());
      std::iota(order.begin(), order.end(), 0);
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cent[2 * (size_t)g[a] + ax] < cent[2 * (size_t)g[b] + ax]; });
      const size_t half = g.size() / 2;
      std::vector<int> a, b;
      a.reserve(half);
      b.reserve(g.size() - half);
      for (size_t i = 0; i < g.size(); ++i) {
        const int c = g[order[i]];
        if (i < half) {
          leaf[c] = leaf[c] * 2;
          a.push_back(c);
        } else {
          leaf[c] = leaf[c] * 2 + 1;
          b.push_back(c);
        }
      }
      next.push_back(std::move(a));
      next.push_back(std::move(b));
    }
    groups.swap(next);
  }
  return leaf;
}

// bisections fused per tree level, root first (ndsolver.uniform_bits / default_bits)
inline std::vector<int> uniform_bits(int depth, int merge, int top_bits) {
  std::vector<int> bits;
  if (top_bits > 0) bits.push_back(top_bits);
  int sum = top_bits > 0 ? top_bits : 0;
  while (sum < depth) {
    bits.push_back(merge);
    sum += merge;
  }
  return bits;
}
// The default tree: leaves of about 12 cells, i.e. log2(nc / 12) bisections, taken to the NEAREST count the fused levels allow (rounding
// up instead gave the pinball and cavity_coarse 16 384 leaves of 3-4 cells: two more launches of blocks too small to fill a workgroup).
// Small meshes (single GPU, <= 16 000 cells: factors that stay in the Infinity Cache, every sweep launch on its ~3.5 us floor) fuse one
// bisection more into each of the two top levels: O1 [3,3,2,2] instead of [2,2,2,2,2] is 9 launches and 236 MB instead of 11 and 212 MB
// per apply = 74.9 instead of 77.5 us, + 3.8 % steps/s (launches x 3.5 us + bytes / 5.4 TB/s reproduces both; the other shapes of ten
// bisections measure worse, profiles/EXPERIMENTS.md).  Where the factors stream from HBM the extra fill costs far more than two launches.
inline std::vector<int> default_bits(int nc, int merge, int top_bits) {
  if (const char* e = std::getenv("FC_ND_SHAPE")) {  // tuning aid: e.g. "2,2,3,3" (partitioned handles: the levels BELOW the 2^top_bits-ary rank level)
    std::vector<int> bits;
    if (top_bits > 0) bits.push_back(top_bits);
    const size_t fixed = bits.size();
    for (const char* q = e; *q;) {
      const int b = std::atoi(q);
      if (b > 0) bits.push_back(b);
      while (*q && *q != ',') ++q;
      if (*q == ',') ++q;
    }
    if (bits.size() > fixed) return bits;
  }
  const double levels = std::log2(std::max(nc, 1) / 12.0);
  if (top_bits > 0) return uniform_bits(std::max(merge + top_bits, (int)std::ceil(levels)), merge, top_bits);  // partitioned handles: rounded up, as ever
  const int d = merge * std::max(1, (int)std::floor(levels / merge + 0.5));
  if (merge == 2 && nc <= 16000 && d >= 8) {
    std::vector<int> bits = {3, 3};
    for (int sum = 6; sum < d; sum += 2) bits.push_back(2);
    return bits;
  }
  return uniform_bits(d, merge, 0);
}

inline Tree build_tree(const std::vector<int>& cell_dofs, int nl, const std::vector<double>& cent, int nc, int N, const std::vector<int>& bits,
                       const std::vector<unsigned char>* skip, int top_bits);
// ... with `depth` bisections fused `merge` at a time below a 2^top_bits-ary root
inline Tree build_tree(const std::vector<int>& cell_dofs, int nl, const std::vector<double>& cent, int nc, int N, int depth,
                       const std::vector<unsigned char>* skip, int merge, int top_bits) {
  return build_tree(cell_dofs, nl, cent, nc, N, uniform_bits(depth, merge, top_bits), skip, top_bits);
}

// element-based nested dissection (ndsolver.build_tree); `bits`: bisections fused per tree level, root first (top_bits > 0: bits[0] is the
// 2^top_bits-ary rank partition of a multi-GPU tree)
inline Tree build_tree(const std::vector<int>& cell_dofs, int nl, const std::vector<double>& cent, int nc, int N, const std::vector<int>& bits,
                       const std::vector<unsigned char>* skip, int top_bits) {
  Tree t;
  int sum = 0;
  for (int b : bits) sum += b;
  t.depth_bin = sum;
  t.cum.assign(1, 0);
  for (int b : bits) t.cum.push_back(t.cum.back() + b);
  const int K = (int)bits.size();
  t.depth = K;
  t.leaf_of_cell = bisect_cells(cent, nc, t.depth_bin);
  const auto& leaf = t.leaf_of_cell;
  std::vector<int64_t> lo((size_t)N, INT64_MAX), hi((size_t)N, -1);
  for (int c = 0; c < nc; ++c)
    for (int k = 0; k < nl; ++k) {
      const int d = cell_dofs[(size_t)c * nl + k];
      lo[d] = std::min<int64_t>(lo[d], leaf[c]);
      hi[d] = std::max<int64_t>(hi[d], leaf[c]);
    }
  for (int d = 0; d < N; ++d)
    if (hi[d] < 0) throw std::runtime_error("dof without any cell");
  if (skip) {
    const int sh = t.depth_bin - top_bits;
    for (int d = 0; d < N; ++d)
      if ((*skip)[d] && (lo[d] >> sh) == (hi[d] >> sh)) hi[d] = lo[d];
  }
  std::vector<int> level((size_t)N);
  std::vector<int64_t> prefix((size_t)N);
  for (int d = 0; d < N; ++d) {
    const int64_t x = lo[d] ^ hi[d];
    int nb_bin = 0;
    if (x > 0) nb_bin = 64 - __builtin_clzll((unsigned long long)x);  // floor(log2 x) + 1
    const int beta = t.depth_bin - nb_bin;
    // searchsorted(cum, beta, side="right") - 1
    const int lv = (int)(std::upper_bound(t.cum.begin(), t.cum.end(), beta) - t.cum.begin()) - 1;
    level[d] = lv;
    prefix[d] = lo[d] >> (t.depth_bin - t.cum[lv]);
  }
  t.perm.resize(N);
  std::iota(t.perm.begin(), t.perm.end(), 0);
  std::stable_sort(t.perm.begin(), t.perm.end(), [&](int a, int b) {
    if (level[a] != level[b]) return level[a] > level[b];  // deepest level first
    return prefix[a] < prefix[b];
  });
  t.iperm.resize(N);
  for (int i = 0; i < N; ++i) t.iperm[t.perm[i]] = i;
  t.node_ptr.assign((size_t)K + 1, {});
  int64_t pos = 0;
  for (int k = K; k >= 0; --k) {
    const int nn_k = 1 << t.cum[k];
    std::vector<int64_t> cnt((size_t)nn_k, 0);
    for (int d = 0; d < N; ++d)
      if (level[d] == k) cnt[(size_t)prefix[d]]++;
    t.node_ptr[k].assign((size_t)nn_k + 1, pos);
    for (int n = 0; n < nn_k; ++n) t.node_ptr[k][n + 1] = t.node_ptr[k][n] + cnt[n];
    pos = t.node_ptr[k][nn_k];
  }
  // boundary sets: dofs touching cells of subtree(t) that are owned by a proper ancestor
  std::vector<int> lvl_new((size_t)N);
  for (int i = 0; i < N; ++i) lvl_new[i] = level[t.perm[i]];
  t.bnd.assign((size_t)K + 1, {});
  std::vector<int> order(nc);
  std::iota(order.begin(), order.end(), 0);
  for (int k = K; k >= 0; --k) {
    const int sh = t.depth_bin - t.cum[k];
    const int nn_k = 1 << t.cum[k];
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return (leaf[a] >> sh) < (leaf[b] >> sh); });
    t.bnd[k].assign((size_t)nn_k, {});
    size_t p0 = 0;
    for (int n = 0; n < nn_k; ++n) {
      size_t p1 = p0;
      while (p1 < (size_t)nc && (leaf[order[p1]] >> sh) == n) ++p1;
      std::vector<int> dd;
      dd.reserve((p1 - p0) * nl);
      for (size_t q = p0; q < p1; ++q)
        for (int j = 0; j < nl; ++j) {
          const int nd = t.iperm[cell_dofs[(size_t)order[q] * nl + j]];
          if (lvl_new[nd] < k) dd.push_back(nd);
        }
      std::sort(dd.begin(), dd.end());
      dd.erase(std::unique(dd.begin(), dd.end()), dd.end());
      t.bnd[k][n] = std::move(dd);
      p0 = p1;
    }
  }
  return t;
}

using Keep = std::function<bool(int, int)>;

struct Factors {  // ndsolver.BlockFactors, structure only
  int N = 0;
  int64_t n_val = 0, nnz = 0;
  std::vector<int> idx;
  std::vector<int64_t> seg_val, seg_ptr, stage_begin;
  std::vector<int> seg_col, seg_len, stage_row0, stage_nrows, stage_kind;
  std::vector<int64_t> nodes;  // [n][7]: level, n, i0, ni, nb, val_off, idx_off
  int64_t root_lo = -1, root_hi = -1;  // multi-GPU: the only stored rows of the root's pivot-block inverse (val_off of the root = row root_lo); -1: all
};

// ndsolver.root_row_block: the rows of the root's pivot-block inverse that rank `rank` applies (and stores)
inline std::pair<int64_t, int64_t> root_row_block(const Tree& t, int rank, int world) {
  const int64_t lo = t.node_ptr[0].front(), hi = t.node_ptr[0].back();
  if (world <= 1) return {lo, hi};
  const int64_t blk = (hi - lo + world - 1) / world;
  const int64_t a = std::min(hi, lo + rank * blk);
  return {a, std::min(hi, a + blk)};
}

// ndsolver.factorize_blocks(None, tree, numeric=False, keep)  (root_lo >= 0: keep.root_rows = (root_lo, root_hi))
inline Factors layout_factors(const Tree& t, const Keep& keep, int64_t root_lo = -1, int64_t root_hi = -1) {
  Factors f;
  f.root_lo = root_lo;
  f.root_hi = root_hi;
  const int N = (int)t.perm.size();
  f.N = N;
  int64_t vpos = 0, ipos = 0;
  std::vector<std::array<int64_t, 2>> dn_val((size_t)N, {0, 0});
  std::vector<std::array<int, 2>> dn_col((size_t)N, {0, 0}), dn_len((size_t)N, {0, 0});
  struct UpSeg {
    int row;
    int64_t val;
    int col, len;
  };
  std::vector<UpSeg> up;
  for (int k = t.depth; k >= 0; --k)
    for (int n = 0; n < t.nnodes(k); ++n) {
      const int64_t i0 = t.node_ptr[k][n], i1 = t.node_ptr[k][n + 1];
      const int64_t ni = i1 - i0;
      const std::vector<int>& B = t.bnd[k][n];
      const int64_t nb = (int64_t)B.size();
      if (keep && !keep(k, n)) continue;
      if (ni == 0) continue;
      const int64_t nf = ni + nb;
      if (nb) {
        f.nodes.insert(f.nodes.end(), {k, n, i0, ni, nb, vpos, ipos});
        for (int64_t r = 0; r < ni; ++r) {
          dn_val[(size_t)(i0 + r)] = {vpos + r * nf, vpos + r * nf + ni};
          dn_col[(size_t)(i0 + r)] = {(int)i0, (int)(-(ipos + 1))};
          dn_len[(size_t)(i0 + r)] = {(int)ni, (int)nb};
        }
        vpos += ni * nf;
        for (int b : B) f.idx.push_back(N + b);
        ipos += nb;
        for (int64_t j = 0; j < nb; ++j) up.push_back({B[(size_t)j], vpos + j * ni, (int)i0, (int)ni});
        vpos += nb * ni;
        f.nnz += ni * ni + 2 * ni * nb;
      } else {
        // stored rows [a, b): all of them, or (the root of a multi-GPU layout) this rank's block
        const int64_t a = (k != 0 || root_lo < 0) ? i0 : std::max(i0, root_lo), b = (k != 0 || root_lo < 0) ? i1 : std::min(i1, root_hi);
        const int64_t nst = std::max<int64_t>(0, b - a);
        f.nodes.insert(f.nodes.end(), {k, n, i0, ni, 0, vpos, 0});
        for (int64_t r = i0; r < i1; ++r) {
          const bool stored = r >= a && r < b;
          dn_val[(size_t)r] = {stored ? vpos + (r - a) * ni : vpos, 0};
          dn_col[(size_t)r] = {(int)i0, 0};
          dn_len[(size_t)r] = {stored ? (int)ni : 0, 0};
        }
        vpos += nst * ni;
        f.nnz += nst * ni;
      }
    }
  f.n_val = vpos;
  // up segments grouped by destination row (stable: deeper nodes first)
  std::stable_sort(up.begin(), up.end(), [](const UpSeg& a, const UpSeg& b) { return a.row < b.row; });
  std::vector<int64_t> up_start((size_t)N + 1, 0);
  for (const UpSeg& s : up) up_start[(size_t)s.row + 1]++;
  for (int i = 0; i < N; ++i) up_start[i + 1] += up_start[i];
  f.seg_ptr.push_back(0);
  int64_t total_rows = 0;
  for (int k = t.depth - 1; k >= 0; --k) {  // up stages
    const int64_t r0 = t.node_ptr[k].front(), r1 = t.node_ptr[k].back();
    f.stage_begin.push_back(total_rows);
    f.stage_row0.push_back((int)r0);
    f.stage_nrows.push_back((int)(r1 - r0));
    f.stage_kind.push_back(0);
    for (int64_t r = r0; r < r1; ++r) {
      for (int64_t q = up_start[r]; q < up_start[r + 1]; ++q) {
        f.seg_val.push_back(up[(size_t)q].val);
        f.seg_col.push_back(up[(size_t)q].col);
        f.seg_len.push_back(up[(size_t)q].len);
      }
      f.seg_ptr.push_back((int64_t)f.seg_val.size());
    }
    total_rows += r1 - r0;
  }
  for (int k = 0; k <= t.depth; ++k) {  // down stages
    const int64_t r0 = t.node_ptr[k].front(), r1 = t.node_ptr[k].back();
    f.stage_begin.push_back(total_rows);
    f.stage_row0.push_back((int)r0);
    f.stage_nrows.push_back((int)(r1 - r0));
    f.stage_kind.push_back(1);
    for (int64_t r = r0; r < r1; ++r) {
      f.seg_val.push_back(dn_val[(size_t)r][0]);
      f.seg_col.push_back(dn_col[(size_t)r][0]);
      f.seg_len.push_back(dn_len[(size_t)r][0]);
      if (dn_len[(size_t)r][1] > 0) {
        f.seg_val.push_back(dn_val[(size_t)r][1]);
        f.seg_col.push_back(dn_col[(size_t)r][1]);
        f.seg_len.push_back(dn_len[(size_t)r][1]);
      }
      f.seg_ptr.push_back((int64_t)f.seg_val.size());
    }
    total_rows += r1 - r0;
  }
  return f;
}

struct Plan {  // ndsolver.FactorPlan
  std::vector<int64_t> nodes;  // [n][7]: level, front offset, nf, ni, voff, parent, slot
  std::vector<int64_t> level_ptr, a_src, a_dst, a_ptr, ext_off, ap_src, node_i0;
  std::vector<int> ext_p, Ap_rowptr, Ap_col;
  int64_t front_size = 0;
  int max_slots = 1;
};

// ndsolver.factor_plan: symbolic side of the device factorisation for the CSR pattern (original numbering)
inline Plan factor_plan(const Tree& t, const Factors& fac, const std::vector<int>& rowptr, const std::vector<int>& col,
                        const std::vector<unsigned char>* skip, const Keep& keep) {
  Plan p;
  const int N = fac.N;
  const int K = t.depth;
  // permuted pattern with the original value index, rows sorted by permuted column
  p.Ap_rowptr.assign((size_t)N + 1, 0);
  for (int i = 0; i < N; ++i) p.Ap_rowptr[i + 1] = p.Ap_rowptr[i] + (rowptr[t.perm[i] + 1] - rowptr[t.perm[i]]);
  const int64_t nnz = p.Ap_rowptr[N];
  p.Ap_col.resize((size_t)nnz);
  p.ap_src.resize((size_t)nnz);
  parallel_for(N, [&](int64_t i) {
    const int r = t.perm[(size_t)i];
    std::vector<std::pair<int, int64_t>> row;
    row.reserve((size_t)(rowptr[r + 1] - rowptr[r]));
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) row.emplace_back(t.iperm[col[k]], (int64_t)k);
    std::sort(row.begin(), row.end());
    for (size_t q = 0; q < row.size(); ++q) {
      p.Ap_col[(size_t)p.Ap_rowptr[(size_t)i] + q] = row[q].first;
      p.ap_src[(size_t)p.Ap_rowptr[(size_t)i] + q] = row[q].second;
    }
  });
  // kept tree nodes in elimination order
  std::vector<int64_t> lv, nn_, i0s, nis, nbs;
  for (int k = K; k >= 0; --k)
    for (int n = 0; n < t.nnodes(k); ++n) {
      if (keep && !keep(k, n)) continue;
      lv.push_back(k);
      nn_.push_back(n);
      i0s.push_back(t.node_ptr[k][n]);
      nis.push_back(t.node_ptr[k][n + 1] - t.node_ptr[k][n]);
      nbs.push_back((int64_t)t.bnd[k][n].size());
    }
  const size_t G = lv.size();
  std::map<std::pair<int, int>, int> gid;
  for (size_t g = 0; g < G; ++g) gid[{(int)lv[g], (int)nn_[g]}] = (int)g;
  std::vector<int64_t> front_off(G + 1, 0);
  for (size_t g = 0; g < G; ++g) front_off[g + 1] = front_off[g] + (nis[g] + nbs[g]) * (nis[g] + nbs[g]);
  std::vector<int64_t> voff(G, -1);
  for (size_t q = 0; q < fac.nodes.size(); q += 7) {
    auto it = gid.find({(int)fac.nodes[q], (int)fac.nodes[q + 1]});
    if (it != gid.end()) voff[(size_t)it->second] = fac.nodes[q + 5];
  }
  std::vector<int> owner((size_t)N, -1);
  for (size_t g = 0; g < G; ++g)
    for (int64_t r = 0; r < nis[g]; ++r) owner[(size_t)(i0s[g] + r)] = (int)g;
  // matrix entries -> front slots, grouped by owner node (stable)
  std::vector<unsigned char> sk;
  if (skip) {
    sk.resize(N);
    for (int i = 0; i < N; ++i) sk[i] = (*skip)[t.perm[i]];
  }
  // counting sort by owner node (stable: row-major order inside a node, as the specification's argsort(kind="stable"))
  auto owner_of = [&](int i, int k) -> int {
    const int c = p.Ap_col[(size_t)k];
    if (skip && (sk[(size_t)i] || sk[(size_t)c]) && i != c) return -1;
    return owner[(size_t)std::min(i, c)];  // < 0: a front another rank builds
  };
  std::vector<int64_t> beg(G + 1, 0);
  for (int i = 0; i < N; ++i)
    for (int k = p.Ap_rowptr[i]; k < p.Ap_rowptr[i + 1]; ++k) {
      const int own = owner_of(i, k);
      if (own >= 0) beg[(size_t)own + 1]++;
    }
  for (size_t g = 0; g < G; ++g) beg[g + 1] += beg[g];
  const int64_t n_ent = beg[G];
  struct Ent {
    int r, c;
  };
  std::vector<Ent> ents((size_t)n_ent);
  p.a_src.resize((size_t)n_ent);
  p.a_dst.resize((size_t)n_ent);
  {
    std::vector<int64_t> fill(beg.begin(), beg.end() - 1);
    for (int i = 0; i < N; ++i)
      for (int k = p.Ap_rowptr[i]; k < p.Ap_rowptr[i + 1]; ++k) {
        const int own = owner_of(i, k);
        if (own < 0) continue;
        const int64_t q = fill[(size_t)own]++;
        ents[(size_t)q] = {i, p.Ap_col[(size_t)k]};
        p.a_src[(size_t)q] = p.ap_src[(size_t)k];
      }
  }
  parallel_for((int64_t)G, [&](int64_t g) {
    const int64_t i0 = i0s[(size_t)g], ni = nis[(size_t)g], nf = ni + nbs[(size_t)g];
    const std::vector<int>& B = t.bnd[(size_t)lv[(size_t)g]][(size_t)nn_[(size_t)g]];
    auto pos = [&](int d) -> int64_t {
      if (d < i0 + ni) return d - i0;
      const auto it = std::lower_bound(B.begin(), B.end(), d);
      if (it == B.end() || *it != d) throw std::runtime_error("matrix entry outside the front: tree/boundary sets inconsistent");
      return ni + (it - B.begin());
    };
    for (int64_t q = beg[(size_t)g]; q < beg[(size_t)g + 1]; ++q)
      p.a_dst[(size_t)q] = front_off[(size_t)g] + pos(ents[(size_t)q].r) * nf + pos(ents[(size_t)q].c);
  });
  // per level ranges (nodes are level-sorted, deepest first)
  p.level_ptr.assign((size_t)K + 2, 0);
  for (int li = 0; li <= K + 1; ++li) {
    const int level = K - li;  // levels K, K-1, ..., 0, then -1 (end)
    size_t g = 0;
    while (g < G && lv[g] > level) ++g;
    p.level_ptr[(size_t)li] = (int64_t)g;
  }
  p.a_ptr.resize((size_t)K + 2);
  for (int li = 0; li <= K + 1; ++li) p.a_ptr[(size_t)li] = beg[(size_t)p.level_ptr[(size_t)li]];
  // extend-add lists
  std::vector<int64_t> parent(G, -1), slot(G, 0);
  p.ext_off.assign(G, -1);
  for (size_t g = 0; g < G; ++g) {
    const int k = (int)lv[g], n = (int)nn_[g];
    if (k == K) continue;
    const int b = t.child_bits(k);
    p.max_slots = std::max(p.max_slots, 1 << b);
    std::vector<int> idxs;
    idxs.reserve((size_t)(nis[g] + nbs[g]));
    for (int64_t r = 0; r < nis[g]; ++r) idxs.push_back((int)(i0s[g] + r));
    idxs.insert(idxs.end(), t.bnd[k][n].begin(), t.bnd[k][n].end());
    for (int c = 0; c < (1 << b); ++c) {
      const int chn = (n << b) + c;
      const auto it = gid.find({k + 1, chn});
      if (it == gid.end()) continue;  // another rank's sub-tree (or not kept)
      const std::vector<int>& cb = t.bnd[(size_t)k + 1][(size_t)chn];
      if (cb.empty()) continue;
      const int gc = it->second;
      parent[(size_t)gc] = (int64_t)g;
      slot[(size_t)gc] = c;
      p.ext_off[(size_t)gc] = (int64_t)p.ext_p.size();
      for (int d : cb) {
        // idxs = [i0 .. i0+ni) ++ sorted B: position by two binary searches
        int64_t pos;
        if (d >= i0s[g] && d < i0s[g] + nis[g])
          pos = d - i0s[g];
        else {
          const auto jt = std::lower_bound(t.bnd[k][n].begin(), t.bnd[k][n].end(), d);
          if (jt == t.bnd[k][n].end() || *jt != d) throw std::runtime_error("child boundary outside the parent front");
          pos = nis[g] + (jt - t.bnd[k][n].begin());
        }
        p.ext_p.push_back((int)pos);
      }
    }
  }
  if (p.ext_p.empty()) p.ext_p.push_back(0);
  p.nodes.resize(G * 7);
  for (size_t g = 0; g < G; ++g) {
    const int64_t row[7] = {lv[g], front_off[g], nis[g] + nbs[g], nis[g], voff[g], parent[g], slot[g]};
    std::copy(row, row + 7, p.nodes.begin() + (long)g * 7);
  }
  p.front_size = front_off[G];
  p.node_i0 = i0s;
  return p;
}

struct Partition {  // ndsolver.RankPartition (tables only)
  std::vector<unsigned char> rowkind;  // ORIGINAL numbering: 0 other rank, 1 owned, 2 root
  std::vector<int> local_cells;
  std::vector<int64_t> seg_ptr, seg_val, stage_begin;
  std::vector<int> seg_col, seg_len, stage_row0, stage_nrows, stage_kind;
  int ar_stage = -1, ar_row0 = 0, ar_n = 0, ar2_stage = -1, root_row0 = 0, root_nrows = 0;
};

inline Partition partition(const Tree& t, const Factors& fac, int rank, int world) {
  Partition P;
  const int N = fac.N;
  int p = 0;
  while ((1 << p) < world) ++p;
  if ((1 << p) != world) throw std::runtime_error("world size must be a power of two");
  if (world > 1 && (t.cum.size() < 2 || t.cum[1] != p)) throw std::runtime_error("tree was not built with top_bits = log2(world)");
  std::vector<int> rank_of((size_t)N, world == 1 ? 0 : -1);
  if (world > 1)
    for (int k = 1; k <= t.depth; ++k) {
      const int sh = t.cum[k] - p;
      for (int n = 0; n < t.nnodes(k); ++n)
        for (int64_t r = t.node_ptr[k][n]; r < t.node_ptr[k][n + 1]; ++r) rank_of[(size_t)r] = n >> sh;
    }
  P.rowkind.assign((size_t)N, 0);
  for (int i = 0; i < N; ++i) P.rowkind[(size_t)t.perm[i]] = rank_of[i] < 0 ? 2 : (rank_of[i] == rank ? 1 : 0);
  for (int c = 0; c < (int)t.leaf_of_cell.size(); ++c)
    if (world == 1 || (t.leaf_of_cell[c] >> (t.depth_bin - p)) == rank) P.local_cells.push_back(c);
  const int64_t root_lo = t.node_ptr[0].front(), root_hi = t.node_ptr[0].back();
  const int64_t blk = (root_hi - root_lo + world - 1) / world;
  const int64_t my_lo = std::min(root_hi, root_lo + rank * blk), my_hi = std::min(root_hi, my_lo + blk);
  P.seg_ptr.push_back(0);
  const int nst = (int)fac.stage_kind.size();
  int64_t rows_total = 0;
  for (int s = 0; s < nst; ++s) {
    const int k = s < t.depth ? (t.depth - 1 - s) : (s - t.depth);
    const int64_t g0 = fac.stage_begin[s];
    const int64_t gr0 = fac.stage_row0[s], gn = fac.stage_nrows[s];
    const int64_t* ptr = fac.seg_ptr.data() + g0;
    auto take_rows = [&](int64_t a, int64_t b, const std::function<bool(int64_t)>& keepseg) {
      for (int64_t r = a; r < b; ++r) {
        for (int64_t q = ptr[r]; q < ptr[r + 1]; ++q)
          if (!keepseg || keepseg(q)) {
            P.seg_val.push_back(fac.seg_val[(size_t)q]);
            P.seg_col.push_back(fac.seg_col[(size_t)q]);
            P.seg_len.push_back(fac.seg_len[(size_t)q]);
          }
        P.seg_ptr.push_back((int64_t)P.seg_val.size());
      }
    };
    P.stage_begin.push_back(rows_total);
    if (k >= 1 && world > 1) {
      const int sh = t.cum[k] - p;
      const int64_t r0 = t.node_ptr[k][(size_t)rank << sh], r1 = t.node_ptr[k][(size_t)(rank + 1) << sh];
      P.stage_row0.push_back((int)r0);
      P.stage_nrows.push_back((int)(r1 - r0));
      take_rows(r0 - gr0, r1 - gr0, nullptr);
      rows_total += r1 - r0;
    } else if (k == 0 && world > 1 && fac.stage_kind[s] == 0) {
      P.stage_row0.push_back((int)gr0);
      P.stage_nrows.push_back((int)gn);
      take_rows(0, gn, [&](int64_t q) { return rank_of[(size_t)fac.seg_col[(size_t)q]] == rank; });
      P.ar_stage = s;
      rows_total += gn;
    } else if (k == 0 && world > 1 && fac.stage_kind[s] != 0) {
      P.stage_row0.push_back((int)my_lo);
      P.stage_nrows.push_back((int)(my_hi - my_lo));
      take_rows(my_lo - gr0, my_hi - gr0, nullptr);
      P.ar2_stage = s;
      rows_total += my_hi - my_lo;
    } else {
      P.stage_row0.push_back((int)gr0);
      P.stage_nrows.push_back((int)gn);
      take_rows(0, gn, nullptr);
      rows_total += gn;
    }
    P.stage_kind.push_back(fac.stage_kind[s]);
  }
  P.ar_row0 = (int)root_lo;
  if (world > 1) {
    P.ar_n = (int)(root_hi - root_lo);
    P.root_row0 = (int)my_lo;
    P.root_nrows = (int)(my_hi - my_lo);
  } else {
    P.ar_stage = P.ar2_stage = -1;
    P.root_row0 = (int)root_lo;
    P.root_nrows = (int)(root_hi - root_lo);
  }
  return P;
}

struct Blocks {  // ndsolver.down_blocks
  std::vector<int64_t> begin, val;
  std::vector<int> count, lpr, row0, nrows, i0, ni, idx, nb;
};

inline Blocks down_blocks(const Tree& t, const Factors& fac, int rank, int world, int max_rows = 32, int target_blocks = 1024, int min_blocks = 512) {
  Blocks B;
  int p = 0;
  while ((1 << p) < world) ++p;
  const int nst = (int)fac.stage_kind.size();
  B.begin.assign((size_t)nst, 0);
  B.count.assign((size_t)nst, 0);
  B.lpr.assign((size_t)nst, 64);
  int64_t nblk = 0;
  const size_t nn = fac.nodes.size() / 7;
  for (int s = 0; s < nst; ++s) {
    B.begin[s] = nblk;
    if (fac.stage_kind[s] != 1) continue;
    const int k = s - t.depth;
    std::vector<size_t> sel;
    for (size_t q = 0; q < nn; ++q) {
      if (fac.nodes[q * 7] != k) continue;
      if (world > 1 && k >= 1 && (fac.nodes[q * 7 + 1] >> (t.cum[k] - p)) != rank) continue;
      sel.push_back(q);
    }
    std::sort(sel.begin(), sel.end(), [&](size_t a, size_t b) { return fac.nodes[a * 7 + 2] < fac.nodes[b * 7 + 2]; });
    int64_t lo = 0, hi = INT64_MAX;
    if (world > 1 && k == 0) {
      const int64_t r_lo = t.node_ptr[0].front(), r_hi = t.node_ptr[0].back();
      const int64_t bs = (r_hi - r_lo + world - 1) / world;
      lo = std::min(r_hi, r_lo + rank * bs);
      hi = std::min(r_hi, lo + bs);
    }
    double values = 0.0;
    int64_t rows = 0;
    for (size_t q : sel) {
      const int64_t i0 = fac.nodes[q * 7 + 2], ni = fac.nodes[q * 7 + 3], nb = fac.nodes[q * 7 + 4];
      const int64_t nr = std::max<int64_t>(0, std::min(i0 + ni, hi) - std::max(i0, lo));
      values += (double)(ni + nb) * (double)nr;
      rows += nr;
    }
    const double wd_mean = values / (double)std::max<int64_t>(rows, 1);
    B.lpr[s] = wd_mean <= 64 ? 16 : (wd_mean <= 128 ? 32 : 64);
    const int slots = 256 / B.lpr[s];
    if (rows / slots < min_blocks) continue;
    int rc = max_rows;
    while (rc > slots && rows / rc < target_blocks) rc /= 2;
    for (size_t q : sel) {
      const int64_t i0 = fac.nodes[q * 7 + 2], ni = fac.nodes[q * 7 + 3], nb = fac.nodes[q * 7 + 4], voff = fac.nodes[q * 7 + 5], ioff = fac.nodes[q * 7 + 6];
      const int64_t wd = ni + nb;
      const int64_t first = std::max<int64_t>(0, lo - i0), last = std::min<int64_t>(ni, hi - i0);
      const int64_t skipped = (k == 0 && fac.root_lo >= 0) ? fac.root_lo - i0 : 0;  // root rows before the stored block
      for (int64_t r0 = first; r0 < last; r0 += rc) {
        B.val.push_back(voff + (r0 - skipped) * wd);
        B.row0.push_back((int)(i0 + r0));
        B.nrows.push_back((int)std::min<int64_t>(rc, last - r0));
        B.i0.push_back((int)i0);
        B.ni.push_back((int)ni);
        B.idx.push_back((int)ioff);
        B.nb.push_back((int)nb);
        ++nblk;
      }
    }
    B.count[s] = (int)(nblk - B.begin[s]);
  }
  return B;
}

}  // namespace fcsym
