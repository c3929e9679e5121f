// Dense front elimination of the device-side numeric factorisation (fc_refactor): blocked Gauss-Jordan on fp64 matrix
// cores.  Included by fc_hip.hip only.
//
// A front  [F11 F12; F21 F22]  (nf x nf row-major, ni pivot columns) is swept IN PLACE into
//     [ F11^-1 ,  F11^-1 F12 ;  -F21 F11^-1 ,  F22 - F21 F11^-1 F12 ]
// (= pivot-block inverse D^-1, the U block, the -L block and the Schur complement the parent receives) by block
// steps of KB pivot columns K = [k0, k0 + kb) (KB = 32 on levels of small fronts, where the pivot inversions' dependent
// chain dominates; KB = 64 on levels whose largest front has order >= FC_FE_WIDE_NF: half the steps, twice the arithmetic per byte
// of front touched by the trailing update; 128 columns -- the fc_fe_*_huge kernels below -- where that traffic is the bound):
//     W        = A[K,K]^-1                      fc_fe_pivot   one workgroup per front: Gauss-Jordan in LDS with partial
//                                                              pivoting inside the block (the inverse of the block
//                                                              does not depend on the pivoting; it only needs it)
//     Cs       = A[:, K]   (copy)               fc_fe_panels
//     A[K, :]  = W A[K, :],  A[K, K] = W        fc_fe_panels
//     A[i, :]  = [j in K ? 0 : A[i, :]] - Cs[i, :] A[K, :]   for rows i not in K      fc_fe_update   v_mfma_f64_16x16x4_f64
// The velocity dofs of a node precede its pressure dofs, so by the time a pivot block reaches the pressure rows their
// diagonal block holds the (definite) Schur complement of the velocities: block-local pivoting is enough; every
// refactorisation is still accepted only after a probe solve (device.py).
// All fronts of one tree level are processed together: blockIdx.y = front, steps beyond a front's ni are no-ops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FC_FE_KB 32          // block step of the levels of small fronts
#define FC_FE_KB_WIDE 64     // ... of the levels of wide fronts
#ifndef FC_FE_WIDE_NF
#define FC_FE_WIDE_NF 3072   // a level is "wide" when its largest front has at least this order (below, the longer pivot chain costs more than the update gains)
#endif
#ifndef FC_FE_PIVOT_KEEP_LOG2
#define FC_FE_PIVOT_KEEP_LOG2 2u  // threshold pivoting: the diagonal is kept while its binary exponent is within this many of the largest candidate's (2: within 8x)
#endif
#ifndef FC_FE_GJ_WAVES
#define FC_FE_GJ_WAVES 4  // waves that invert a pivot block: 4 (columns split over the workgroup's waves, one barrier per column step) or 1
#endif
#ifndef FC_FE_PIVOT_UNROLL
#define FC_FE_PIVOT_UNROLL 8  // column steps per trip of the pivot-block loop: the row rotation costs one register move per entry and TRIP
#endif
#ifndef FC_FE_HUGE_MB
#define FC_FE_HUGE_MB 256.0  // a level takes 128-column steps (fc_fe_*_huge) when its fronts together hold at least this many MB: the update is then
#endif                       // bound by the traffic of the fronts (a level of one or few fronts, however wide -- the root --, is bound by the pivot chain instead)
#ifndef FC_FE_HUGE_NF
#define FC_FE_HUGE_NF (1 << 30)  // ... or when its largest front has at least this order (off; tests force the kernels on small meshes with it)
#endif
#define FC_FE_HUGE_MIN_NF 256  // ... and never below this one (the scratch of such levels is sized for 64-column steps)
#define FC_FE_KB_MAX 64      // scratch layout: W (KB_MAX x KB_MAX) then Cs (nf x KB_MAX), whatever KB a level uses

struct __attribute__((aligned(16))) FcFront {
  long long front;  // offset of the nf x nf row-major front
  long long voff;   // offset of the node's factor values
  int nf, ni;
  long long scratch;  // offset of this front's scratch: W (KB x KB) then Cs (nf x KB)
};

typedef double fc_d4 __attribute__((ext_vector_type(4)));

// Multi-GPU root front: of the swept front this handle exports only the pivot rows [keep0, keep1) (its block of the root's D^-1).  A
// row above the current pivot block (already eliminated) that is not one of them is DEAD: nothing reads it again, so the 64-row tiles
// made of such rows are skipped by the panel copy and the trailing update -- per rank (1 + 1 / world) n^3 instead of 2 n^3 flops for the
// replicated root elimination.  keep0 = 0, keep1 = INT_MAX: every row is kept (all other fronts).
__device__ __forceinline__ bool fc_fe_dead_rows(int i0, int k0, int keep0, int keep1) {
  return i0 + 64 <= k0 && (i0 + 64 <= keep0 || i0 >= keep1);
}

// W = A[K,K]^-1 by Gauss-Jordan with partial pivoting among the block's rows (ties -> smallest row: reproducible)
// (a: KB x (KB + 1) doubles and piv: KB ints of LDS, provided by the caller: the stand-alone kernel below for a front's
// first step, fc_fe_update for every later one)
//
// The KB dependent column steps of this inversion are the critical path of every block step (the trailing update hides
// behind it, not the other way round): they run on ONE wave out of registers — lane r owns row r (KB doubles), the pivot
// search is a wave reduction, the pivot row reaches the other lanes as scalar broadcasts (v_readlane), a row swap is a
// lane permute — with no workgroup barrier and no LDS round trip inside a column step (the LDS version spent 1.4 us per
// column: 44 us per 32-column block, 46 us measured for the stand-alone kernel).  LDS carries the
// block in (coalesced load by all threads) and out (the columns of the inverse permuted back).
// value of `v` in lane `l` (wave-uniform l) as a scalar broadcast
__device__ __forceinline__ double fc_readlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// max of v over the 16 lanes of a DPP row, left in every lane of the row (0 is the neutral element)
__device__ __forceinline__ unsigned fc_row_max_u32(unsigned v) {
  unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);  // quad_perm [1, 0, 3, 2]
  v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);  // quad_perm [2, 3, 0, 1]
  v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, true);  // row_ror:4
  v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);  // row_ror:8
  return o > v ? o : v;
}

// One wave inverts a KB x KB block held as lane r = row r (x[c] = its KB entries; lanes >= KB shadow the last row), in
// place; kb <= KB valid rows / columns, the rest identity padding.  On return x holds the rows of the swept block and
// piv[KB + c] (written by lane 0) the column of the swept block that is column c of the inverse (row exchanges undone).
template <int KB>
__device__ __forceinline__ void fc_fe_gj_wave(double (&x)[KB], int lane, int kb, int* piv) {
  // A COMPACT loop over the KB columns (the fully unrolled form is 64 KB of straight-line code: instruction fetch then
  // costs what the barriers cost before): every step works on register 0 and rotates the row by one position while it
  // updates it — after KB steps the columns are back in place.  Steps k >= kb meet the identity padding: no-ops.
  // One wave issues one instruction at a time, so a column step costs its instruction count:
  //  * pivot search on a 32-bit key (upper word of |x|: exponent + 20 mantissa bits, the lane in the low 6 bits) reduced
  //    with DPP row operations + one readlane per 16-lane row — no LDS permutes; the diagonal is kept whenever it is
  //    within 8x of the largest candidate (threshold pivoting: a row exchange is 2 KB permutes), ties -> smallest row;
  //  * 1 / pivot by v_rcp_f64 + two Newton steps;
  //  * the pivot row is NOT scaled inside the loop (its lane multiplies by its 1 / pivot once, at the end: a scaled
  //    row only ever acts on itself afterwards, so the scaling commutes with the later column steps): every entry
  //    of the rank-1 update is then two scalar broadcasts (v_readlane) and one v_fma_f64 with g = x[0] / pivot.
  double dsave = 1.0;
#pragma clang loop unroll_count(FC_FE_PIVOT_UNROLL)
  for (int k = 0; k < KB; ++k) {
    const unsigned hi = (unsigned)__double2hiint(x[0]) & 0x7fffffffu;
    const bool cand = k < kb ? (lane >= k && lane < kb) : lane == k;
    const unsigned key = cand ? ((hi & ~63u) | (unsigned)(63 - lane)) : 0u;
    unsigned m = fc_row_max_u32(key);
    unsigned mw = (unsigned)__builtin_amdgcn_readlane((int)m, 0);
#pragma unroll
    for (int rw = 16; rw < KB; rw += 16) {
      const unsigned o = (unsigned)__builtin_amdgcn_readlane((int)m, rw);
      mw = o > mw ? o : mw;
    }
    const unsigned kd = (unsigned)__builtin_amdgcn_readlane((int)key, k);
    const int p = (mw >> 20) <= (kd >> 20) + FC_FE_PIVOT_KEEP_LOG2 ? k : 63 - (int)(mw & 63u);
    if (lane == 0) piv[k] = p;
    if (p != k) {  // exchange rows k and p: a permute between two lanes
      const int partner = lane == k ? p : (lane == p ? k : lane);
#pragma unroll
      for (int c = 0; c < KB; ++c) x[c] = __shfl(x[c], partner, 64);
    }
    const double pv = fc_readlane(x[0], k);
    double d = __builtin_amdgcn_rcp(pv);
    d = __builtin_fma(__builtin_fma(-pv, d, 1.0), d, d);
    d = __builtin_fma(__builtin_fma(-pv, d, 1.0), d, d);
    const bool isk = lane == k;
    const double g = isk ? 0.0 : x[0] * d;  // this row's entry in the pivot column over the pivot
    if (isk) dsave = d;
#pragma unroll
    for (int c = 1; c < KB; ++c) x[c - 1] = __builtin_fma(-g, fc_readlane(x[c], k), x[c]);
    x[KB - 1] = isk ? 1.0 : -g;  // the swept pivot column takes the free slot at the end of the rotation
  }
#pragma unroll
  for (int c = 0; c < KB; ++c) x[c] *= dsave;
  if (lane == 0) {
    for (int c = 0; c < KB; ++c) piv[KB + c] = c;
    for (int k = kb - 1; k >= 0; --k) {
      const int p = piv[k];
      if (p != k) {
        const int u = piv[KB + k];
        piv[KB + k] = piv[KB + p];
        piv[KB + p] = u;
      }
    }
  }
}

// The same inversion on FOUR waves: wave w owns the columns [KB/4 w, KB/4 (w + 1)) of every row (lane r = row r, KB/4 entries in
// registers).  Per column step the owner of the column finds the pivot, forms g_r = x_r[k] / pivot for every row and leaves it in
// LDS; one workgroup barrier; then every wave updates its own columns (two v_readlane + one v_fmac_f64 per entry).  The rank-1
// update -- three quarters of a column step on one wave -- is spread over the four SIMDs of the CU; what stays serial is the
// search, the reciprocal and the barrier.  a: the block (LDS, row stride KB + 1), swept in place; gj: LDS scratch of
// 2 * 64 + KB doubles; piv: 2 * KB + 2 ints.  All 256 threads call it; ends with a barrier; piv[KB + c] as fc_fe_gj_wave.
template <int KB>
__device__ __forceinline__ void fc_fe_gj_block(double (*a)[KB + 1], int kb, double* gj, int* piv) {
  constexpr int CW = KB / 4;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int r = lane < KB ? lane : KB - 1;  // lanes beyond the block shadow its last row
  double* dsc = gj + 128;
  int* pcur = piv + 2 * KB;
  double x[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) x[c] = a[r][CW * wave + c];
  if (t < KB) dsc[t] = 1.0;
#pragma clang loop unroll(disable)
  for (int ow = 0; ow < 4; ++ow) {
#pragma unroll
    for (int kl = 0; kl < CW; ++kl) {
      const int k = CW * ow + kl;
      double* gb = gj + 64 * (k & 1);
      if (wave == ow) {
        const unsigned hi = (unsigned)__double2hiint(x[kl]) & 0x7fffffffu;
        const bool cand = k < kb ? (lane >= k && lane < kb) : lane == k;
        const unsigned key = cand ? ((hi & ~63u) | (unsigned)(63 - lane)) : 0u;
        unsigned m = fc_row_max_u32(key);
        unsigned mw = (unsigned)__builtin_amdgcn_readlane((int)m, 0);
#pragma unroll
        for (int rw = 16; rw < KB; rw += 16) {
          const unsigned o = (unsigned)__builtin_amdgcn_readlane((int)m, rw);
          mw = o > mw ? o : mw;
        }
        const unsigned kd = (unsigned)__builtin_amdgcn_readlane((int)key, k);
        const int p = (mw >> 20) <= (kd >> 20) + FC_FE_PIVOT_KEEP_LOG2 ? k : 63 - (int)(mw & 63u);
        if (p != k) {
          const int partner = lane == k ? p : (lane == p ? k : lane);
#pragma unroll
          for (int c = 0; c < CW; ++c) x[c] = __shfl(x[c], partner, 64);
        }
        const double pv = fc_readlane(x[kl], k);
        double d = __builtin_amdgcn_rcp(pv);
        d = __builtin_fma(__builtin_fma(-pv, d, 1.0), d, d);
        d = __builtin_fma(__builtin_fma(-pv, d, 1.0), d, d);
        const bool isk = lane == k;
        const double g = isk ? 0.0 : x[kl] * d;
        if (lane < KB) gb[lane] = g;
        if (isk) dsc[k] = d;
        if (lane == 0) {
          pcur[k & 1] = p;
          piv[k] = p;
        }
        x[kl] = isk ? 1.0 : -g;  // the swept pivot column
      }
      __syncthreads();
      const int p = pcur[k & 1];
      const double ng = -gb[r];
      if (wave == ow) {
#pragma unroll
        for (int c = 0; c < CW; ++c)
          if (c != kl) x[c] = __builtin_fma(ng, fc_readlane(x[c], k), x[c]);
      } else {
        if (p != k) {
          const int partner = lane == k ? p : (lane == p ? k : lane);
#pragma unroll
          for (int c = 0; c < CW; ++c) x[c] = __shfl(x[c], partner, 64);
        }
#pragma unroll
        for (int c = 0; c < CW; ++c) x[c] = __builtin_fma(ng, fc_readlane(x[c], k), x[c]);
      }
    }
  }
  __syncthreads();
  if (lane < KB) {
    const double ds = dsc[lane];
#pragma unroll
    for (int c = 0; c < CW; ++c) a[lane][CW * wave + c] = x[c] * ds;
  }
  if (t == 0) {
    for (int c = 0; c < KB; ++c) piv[KB + c] = c;
    for (int k = kb - 1; k >= 0; --k) {
      const int p = piv[k];
      if (p != k) {
        const int u = piv[KB + k];
        piv[KB + k] = piv[KB + p];
        piv[KB + p] = u;
      }
    }
  }
  __syncthreads();
}

template <int KB>
__device__ __forceinline__ void fc_fe_pivot_finish(const FcFront& nd, double* __restrict__ scratch, int kb, double (*a)[KB + 1], int* piv, double* gj);

template <int KB>
__device__ __forceinline__ void fc_fe_pivot_block(const FcFront& nd, const double* fronts, double* __restrict__ scratch, int step,
                                                  double (*a)[KB + 1], int* piv, double* gj) {
  const int k0 = step * KB;
  const int kb = nd.ni - k0 < KB ? nd.ni - k0 : KB;
  const int nf = nd.nf;
  const double* A = fronts + nd.front;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  for (int e = t; e < KB * KB; e += 256) {
    const int r = e / KB, c = e % KB;
    a[r][c] = (r < kb && c < kb) ? A[(size_t)(k0 + r) * nf + k0 + c] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  fc_fe_pivot_finish<KB>(nd, scratch, kb, a, piv, gj);
}

// a (LDS, identity-padded beyond kb, visible to the whole workgroup) -> its inverse -> W in the front's scratch
template <int KB>
__device__ __forceinline__ void fc_fe_pivot_finish(const FcFront& nd, double* __restrict__ scratch, int kb, double (*a)[KB + 1], int* piv, double* gj) {
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
#if FC_FE_GJ_WAVES == 4
  fc_fe_gj_block<KB>(a, kb, gj, piv);
#else
  if (wave == 0) {
    const int r = lane < KB ? lane : KB - 1;  // lanes beyond the block shadow its last row (KB = 32: half the wave)
    double x[KB];
#pragma unroll
    for (int c = 0; c < KB; ++c) x[c] = a[r][c];
    fc_fe_gj_wave<KB>(x, lane, kb, piv);
    // inverse back to LDS; the row swaps are undone on its columns when it is read
    if (lane < KB) {
#pragma unroll
      for (int c = 0; c < KB; ++c) a[lane][c] = x[c];
    }
  }
  __syncthreads();
#endif
  double* W = scratch + nd.scratch;
  for (int e = t; e < KB * KB; e += 256) {
    const int r = e / KB, c = e % KB;
    W[e] = (r < kb && c < kb) ? a[r][piv[KB + c]] : 0.0;
  }
}

template <int KB>
__global__ __launch_bounds__(256) void fc_fe_pivot(const FcFront* __restrict__ nodes, double* fronts, double* __restrict__ scratch, int step) {
  __shared__ double a[KB][KB + 1];
  __shared__ int piv[2 * KB + 2];
  __shared__ double gj[128 + KB];
  const FcFront nd = nodes[blockIdx.x];
  if (step * KB >= nd.ni) return;
  fc_fe_pivot_block<KB>(nd, fronts, scratch, step, a, piv, gj);
}

// blockIdx.x < ct: row panel, 64 columns per workgroup (two passes of 32):  A[K, j] = sum_c W[., c] A[k0 + c, j]  (j in K: = W)
// blockIdx.x >= ct: column panel copy, 64 rows per workgroup: Cs[i, c] = A[i, k0 + c]  (zero beyond kb)
template <int KB>
__global__ __launch_bounds__(256) void fc_fe_panels(const FcFront* __restrict__ nodes, double* fronts, double* __restrict__ scratch, int step,
                                                    int ct, int keep0 = 0, int keep1 = 0x7fffffff) {
  __shared__ double Ws[KB][KB];  // read as Ws[r][c] with r (nearly) uniform over a wave: broadcast, no padding needed
  __shared__ double Rs[KB][KB == 32 ? 2 * 33 : 32 + 1];
  const FcFront nd = nodes[blockIdx.y];
  const int k0 = step * KB;
  if (k0 >= nd.ni) return;
  const int kb = nd.ni - k0 < KB ? nd.ni - k0 : KB;
  const int nf = nd.nf;
  double* A = fronts + nd.front;
  const double* W = scratch + nd.scratch;
  double* Cs = scratch + nd.scratch + FC_FE_KB_MAX * FC_FE_KB_MAX;
  const int t = threadIdx.x;
  if ((int)blockIdx.x >= ct) {
    const int i0 = ((int)blockIdx.x - ct) * 64;
    if (i0 >= nf || fc_fe_dead_rows(i0, k0, keep0, keep1)) return;
    for (int e = t; e < 64 * KB; e += 256) {
      const int i = i0 + e / KB, c = e % KB;
      if (i < nf) Cs[((size_t)(c >> 2) * nf + i) * 4 + (c & 3)] = c < kb ? A[(size_t)i * nf + k0 + c] : 0.0;  // [k / 4][row][k % 4]: the order the update's MFMA A operands are loaded in
    }
    return;
  }
  const int j0 = blockIdx.x * 64;
  if (j0 >= nf) return;
  if constexpr (KB == 32) {
    // both halves at once: every load of the workgroup (W and the 32 x 64 old pivot rows) is in flight before the one barrier --
    // the panel kernel is on the dependent chain of every block step of the narrow levels, its latency is what counts
    double (*R2)[64 + 1] = reinterpret_cast<double (*)[64 + 1]>(&Rs[0][0]);  // (Rs is declared [KB][2 * 33] for KB = 32)
    double wq[4], rq[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) wq[q] = W[t + 256 * q];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = t + 256 * q, c = e / 64, j = j0 + e % 64;
      rq[q] = (c < kb && j < nf) ? A[(size_t)(k0 + c) * nf + j] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) Ws[(t + 256 * q) / KB][(t + 256 * q) % KB] = wq[q];
#pragma unroll
    for (int q = 0; q < 8; ++q) R2[(t + 256 * q) / 64][(t + 256 * q) % 64] = rq[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = t + 256 * q, r = e / 64, jj = e % 64, j = j0 + jj;
      if (r >= kb || j >= nf) continue;
      double sum;
      if (j >= k0 && j < k0 + kb) {
        sum = Ws[r][j - k0];
      } else {
        sum = 0.0;
#pragma unroll 8
        for (int c = 0; c < KB; ++c) sum += Ws[r][c] * R2[c][jj];
      }
      A[(size_t)(k0 + r) * nf + j] = sum;
    }
    return;
  }
  for (int e = t; e < KB * KB; e += 256) Ws[e / KB][e % KB] = W[e];
  for (int half = 0; half < 2; ++half) {
    const int jh = j0 + 32 * half;
    if (jh >= nf) break;
    __syncthreads();
    for (int e = t; e < KB * 32; e += 256) {
      const int c = e / 32, j = jh + e % 32;
      Rs[c][e % 32] = (c < kb && j < nf) ? A[(size_t)(k0 + c) * nf + j] : 0.0;
    }
    __syncthreads();
    for (int e = t; e < KB * 32; e += 256) {
      const int r = e / 32, jj = e % 32, j = jh + jj;
      if (r >= kb || j >= nf) continue;
      double sum;
      if (j >= k0 && j < k0 + kb) {
        sum = Ws[r][j - k0];
      } else {
        sum = 0.0;
#pragma unroll 8
        for (int c = 0; c < KB; ++c) sum += Ws[r][c] * Rs[c][jj];
      }
      A[(size_t)(k0 + r) * nf + j] = sum;
    }
  }
}

// 64 x 64 tile of the trailing update on the fp64 matrix cores: wave w owns rows [16 w, 16 w + 16) x 64 columns
// (four 16 x 16 accumulators).  A operand: Cs (rows of the tile, K = KB) in registers; B operand: the new pivot rows
// A[K, j0 .. j0 + 64), staged once per workgroup in LDS (the four waves share them).  Everything a wave needs from memory —
// its Cs rows, its share of the B panel, its 16 entries of the C tile — is requested before the first MFMA, so a tile pays
// the memory latency once.
//   v_mfma_f64_16x16x4_f64: lane l holds A[row l & 15][k = l >> 4], B[k = l >> 4][col l & 15];
//   D[row (l >> 4) + 4 r][col l & 15] in register r.
template <int KB>
__global__ __launch_bounds__(256) void fc_fe_update(const FcFront* __restrict__ nodes, double* fronts, double* scratch,
                                                    int step, int tiles_per_side, int keep0 = 0, int keep1 = 0x7fffffff) {
  __shared__ double smem[KB * (KB + 1) > KB * 64 ? KB * (KB + 1) : KB * 64];  // the B panel, then (one tile only) the next pivot block
  __shared__ int piv[2 * KB + 2];
  __shared__ double gj[128 + KB];
  double (*Bs)[64] = reinterpret_cast<double (*)[64]>(smem);
  const FcFront nd = nodes[blockIdx.y];
  const int k0 = step * KB;
  if (k0 >= nd.ni) return;
  const int kb = nd.ni - k0 < KB ? nd.ni - k0 : KB;
  const int nf = nd.nf;
  double* A = fronts + nd.front;
  const double* Cs = scratch + nd.scratch + FC_FE_KB_MAX * FC_FE_KB_MAX;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int lr = lane & 15, lk = lane >> 4;
  if (blockIdx.x == 0) {
    // look-ahead workgroup (scheduled first): it updates ONLY the next pivot block -- KB x KB entries, straight from the panels into
    // the matrix cores and on into LDS -- and inverts it right away, while the other workgroups of the launch update their tiles
    // (one of them computes the same entries again and stores them; this one stores only the inverse).  The next step then starts
    // with its panels; the dependent chain of a block step is this workgroup: launch, three loads, KB column steps, one store.
    const int k1 = k0 + KB;
    if (k1 >= nd.ni) return;
    const int kb1 = nd.ni - k1 < KB ? nd.ni - k1 : KB;
    double (*a)[KB + 1] = reinterpret_cast<double (*)[KB + 1]>(smem);
    constexpr int NT = KB / 16;
    for (int tile = wave; tile < NT * NT; tile += 4) {
      const int rt = tile / NT, ct = tile % NT;
      const int arow = k1 + 16 * rt + lr, bcol = k1 + 16 * ct + lr;
      double av[KB / 4], bv[KB / 4], cin[4];
#pragma unroll
      for (int s4 = 0; s4 < KB / 4; ++s4) {
        av[s4] = arow < nf ? Cs[((size_t)s4 * nf + arow) * 4 + lk] : 0.0;  // Cs[arow][4 s4 + lk], k-step major; zero beyond kb (fc_fe_panels)
        bv[s4] = (4 * s4 + lk < kb && bcol < nf) ? A[(size_t)(k0 + 4 * s4 + lk) * nf + bcol] : 0.0;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * rt + lk + 4 * r, col = 16 * ct + lr;
        cin[r] = (row < kb1 && col < kb1) ? A[(size_t)(k1 + row) * nf + k1 + col] : 0.0;
      }
      fc_d4 acc = fc_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s4 = 0; s4 < KB / 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], bv[s4], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * rt + lk + 4 * r, col = 16 * ct + lr;
        const bool in = row < kb1 && col < kb1;
        const double v = cin[r] - acc[r];
        a[row][col] = in ? v : (row == col ? 1.0 : 0.0);
        if (in) A[(size_t)(k1 + row) * nf + k1 + col] = v;  // these entries are this workgroup's alone (the tile that holds them skips them)
      }
    }
    __syncthreads();
    fc_fe_pivot_finish<KB>(nd, scratch, kb1, a, piv, gj);
    return;
  }
  const int k1 = k0 + KB, k1e = k1 < nd.ni ? (nd.ni < k1 + KB ? nd.ni : k1 + KB) : k1;  // [k1, k1e): the next pivot block (empty after the last step)
  const int tile = (int)blockIdx.x - 1;
  const int ti = tile / tiles_per_side, tj = tile % tiles_per_side;
  const int i0 = ti * 64, j0 = tj * 64;
  if (i0 >= nf || j0 >= nf || fc_fe_dead_rows(i0, k0, keep0, keep1)) return;
  // B panel: KB x 64, a row of 64 columns per 64 consecutive threads (coalesced)
  constexpr int BPT = KB * 64 / 256;
  double bq[BPT];
#pragma unroll
  for (int q = 0; q < BPT; ++q) {
    const int e = t + 256 * q, k = e / 64, col = j0 + e % 64;
    bq[q] = (k < kb && col < nf) ? A[(size_t)(k0 + k) * nf + col] : 0.0;
  }
  const int arow = i0 + 16 * wave + lr;
  double av[KB / 4];
#pragma unroll
  for (int s = 0; s < KB / 4; ++s) av[s] = arow < nf ? Cs[((size_t)s * nf + arow) * 4 + lk] : 0.0;  // Cs[arow][4 s + lk], k-step major; zero beyond kb (fc_fe_panels)
  // the wave's entries of the C tile (pivot rows and everything outside the front: not touched)
  double cv[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int col = j0 + 16 * c + lr;
    const bool inK = col >= k0 && col < k0 + kb;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + 16 * wave + lk + 4 * r;
      const bool live = col < nf && row < nf && !(row >= k0 && row < k0 + kb);
      cv[c][r] = (live && !inK) ? A[(size_t)row * nf + col] : 0.0;
    }
  }
#pragma unroll
  for (int q = 0; q < BPT; ++q) {
    const int e = t + 256 * q;
    Bs[e / 64][e % 64] = bq[q];
  }
  __syncthreads();
  fc_d4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    acc[c] = fc_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KB / 4; ++s) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], Bs[4 * s + lk][16 * c + lr], acc[c], 0, 0, 0);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int col = j0 + 16 * c + lr;
    if (col >= nf) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + 16 * wave + lk + 4 * r;
      if (row >= nf || (row >= k0 && row < k0 + kb)) continue;  // the pivot rows are final (fc_fe_panels)
      if (row >= k1 && row < k1e && col >= k1 && col < k1e) continue;  // the next pivot block: read and written by the look-ahead workgroup only
      A[(size_t)row * nf + col] = cv[c][r] - acc[c][r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 128-column block steps for the levels of WIDE fronts, where the trailing update is bound by the HBM traffic of the front
// (read + written once per block step: 4 flop/B at KB = 32, 8 at 64, 16 at 128): half the passes over the front of the
// 64-column steps.  The pivot block (128 x 128) is inverted by ONE workgroup per front entirely in LDS -- Gauss-Jordan in
// four sub-steps of 32 columns: the 32 x 32 diagonal sub-block on one wave out of registers (fc_fe_gj_wave), its row panel
// and the rank-32 update of the other 96 rows by all four waves out of LDS -- so the dependent chain of a 128-column step
// is one launch.  Scratch of a front: W (128 x 128) then Cs (nf x 128), both stored [k / 4][row][k % 4].
// ---------------------------------------------------------------------------------------------------------------------
#define FC_FE_KH 128
#define FC_FE_KH_LD (FC_FE_KH + 1)
#define FC_FE_KH_LDS_BYTES ((FC_FE_KH * FC_FE_KH_LD + 2 * 32 * 33 + 160) * 8 + 66 * 4)

__global__ __launch_bounds__(256) void fc_fe_pivot_huge(const FcFront* __restrict__ nodes, const double* fronts, double* __restrict__ scratch, int step) {
  extern __shared__ double fc_fe_lds[];
  constexpr int KH = FC_FE_KH, LD = FC_FE_KH_LD;
  double* a = fc_fe_lds;                                                   // [KH][LD] the pivot block, swept in place
  double (*sub)[33] = reinterpret_cast<double (*)[33]>(a + KH * LD);       // rows of the swept 32 x 32 sub-block
  double (*subp)[33] = sub + 32;                                           // its inverse (columns permuted back)
  double* gj = reinterpret_cast<double*>(subp + 32);                       // 128 + 32 doubles: scratch of fc_fe_gj_block
  int* piv = reinterpret_cast<int*>(gj + 160);                            // 66 ints
  const FcFront nd = nodes[blockIdx.x];
  const int k0 = step * KH;
  if (k0 >= nd.ni) return;
  const int kb = nd.ni - k0 < KH ? nd.ni - k0 : KH;
  const int nf = nd.nf;
  const double* A = fronts + nd.front;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  for (int e = t; e < KH * KH; e += 256) {
    const int r = e / KH, c = e % KH;
    a[r * LD + c] = (r < kb && c < kb) ? A[(size_t)(k0 + r) * nf + k0 + c] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  for (int c0 = 0; c0 < kb; c0 += 32) {  // (sub-steps beyond kb meet the identity padding: nothing to do)
    const int kbs = kb - c0 < 32 ? kb - c0 : 32;
    // the 32 x 32 diagonal sub-block, inverted by all four waves (fc_fe_gj_block)
    for (int e = t; e < 32 * 32; e += 256) sub[e / 32][e % 32] = a[(c0 + e / 32) * LD + c0 + e % 32];
    __syncthreads();
    fc_fe_gj_block<32>(sub, kbs, gj, piv);
    for (int e = t; e < 32 * 32; e += 256) subp[e / 32][e % 32] = sub[e / 32][piv[32 + e % 32]];
    __syncthreads();
    const int lr = lane & 15, lk = lane >> 4;
    {
      // row panel on the matrix cores: a[c0 + r, j] = sum_k subp[r, k] a[c0 + k, j]  (j in the sub-block's columns: = subp).  Wave w owns
      // the columns [32 w, 32 w + 32): it reads and writes only those, so the panel is swept in place without a barrier.
      fc_d4 acc[2][2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) acc[rt][cc] = fc_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s4 = 0; s4 < 8; ++s4) {
        const double b0 = a[(c0 + 4 * s4 + lk) * LD + 32 * wave + lr], b1 = a[(c0 + 4 * s4 + lk) * LD + 32 * wave + 16 + lr];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const double av = subp[16 * rt + lr][4 * s4 + lk];
          acc[rt][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b0, acc[rt][0], 0, 0, 0);
          acc[rt][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b1, acc[rt][1], 0, 0, 0);
        }
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const int j = 32 * wave + 16 * cc + lr;
          const bool inK = j >= c0 && j < c0 + 32;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * rt + lk + 4 * r;
            a[(c0 + row) * LD + j] = inK ? subp[row][j - c0] : acc[rt][cc][r];
          }
        }
    }
    __syncthreads();
    {
      // the other rows on the matrix cores: a[i, j] = [j in the sub-block's columns ? 0 : a[i, j]] - a[i, c0 .. c0 + 32) a[c0 .. c0 + 32, j].
      // Wave w owns the row tiles w and w + 4 (16 rows each; the two tiles of the pivot rows are skipped): a row tile is read
      // and written by its owner only, its old pivot-column entries go to registers before any of it is overwritten.
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int rt = wave + 4 * q;
        if (16 * rt >= c0 && 16 * rt < c0 + 32) continue;  // (wave-uniform)
        double cs[8];
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) cs[s4] = a[(16 * rt + lr) * LD + c0 + 4 * s4 + lk];
#pragma clang loop unroll_count(2)
        for (int ct = 0; ct < 8; ++ct) {
          const int j = 16 * ct + lr;
          const bool inK = j >= c0 && j < c0 + 32;
          fc_d4 acc = fc_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s4 = 0; s4 < 8; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cs[s4], a[(c0 + 4 * s4 + lk) * LD + j], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double* dst = a + (16 * rt + lk + 4 * r) * LD + j;
            *dst = (inK ? 0.0 : *dst) - acc[r];
          }
        }
      }
    }
    __syncthreads();
  }
  // W goes out in the order fc_fe_panels_huge reads it as MFMA A operands: [k / 4][row][k % 4] (a wave's load of one k-step is then
  // 16 rows x 32 B back to back instead of sixteen 32-B pieces a row apart)
  double* W = scratch + nd.scratch;
  for (int e = t; e < KH * KH; e += 256) {
    const int r = (e / 4) % KH, c = 4 * (e / (4 * KH)) + e % 4;
    W[e] = (r < kb && c < kb) ? a[r * LD + c] : 0.0;
  }
}

// blockIdx.x < ct: row panel of 64 columns on the matrix cores:  A[K, j] = sum_c W[., c] A[k0 + c, j]  (j in K: = W); wave w computes
//                  rows [32 w, 32 w + 32) (its W rows in registers, the 128 x 64 panel of A staged once in LDS)
// blockIdx.x >= ct: column panel copy, 64 rows per workgroup: Cs[i, c] = A[i, k0 + c]  (zero beyond kb)
#define FC_FE_KH_PANEL_LDS_BYTES (FC_FE_KH * 64 * 8)
__global__ __launch_bounds__(256) void fc_fe_panels_huge(const FcFront* __restrict__ nodes, double* fronts, double* __restrict__ scratch, int step, int ct,
                                                         int keep0 = 0, int keep1 = 0x7fffffff) {
  extern __shared__ double fc_fe_lds[];
  constexpr int KH = FC_FE_KH;
  const FcFront nd = nodes[blockIdx.y];
  const int k0 = step * KH;
  if (k0 >= nd.ni) return;
  const int kb = nd.ni - k0 < KH ? nd.ni - k0 : KH;
  const int nf = nd.nf;
  double* A = fronts + nd.front;
  const double* W = scratch + nd.scratch;
  double* Cs = scratch + nd.scratch + KH * KH;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  if ((int)blockIdx.x >= ct) {
    const int i0 = ((int)blockIdx.x - ct) * 64;
    if (i0 >= nf || fc_fe_dead_rows(i0, k0, keep0, keep1)) return;
    // Cs is stored [k / 4][row][k % 4]: the order fc_fe_update_huge reads it as MFMA A operands (16 rows x 32 B back to back per load)
    for (int e = t; e < 64 * KH; e += 256) {
      const int i = i0 + e / KH, c = e % KH;
      if (i < nf) Cs[((size_t)(c >> 2) * nf + i) * 4 + (c & 3)] = c < kb ? A[(size_t)i * nf + k0 + c] : 0.0;
    }
    return;
  }
  const int j0 = blockIdx.x * 64;
  if (j0 >= nf) return;
  double (*Bs)[64] = reinterpret_cast<double (*)[64]>(fc_fe_lds);  // [KH][64] the old pivot rows of these columns
  for (int e = t; e < KH * 64; e += 256) {
    const int c = e / 64, j = j0 + e % 64;
    Bs[c][e % 64] = (c < kb && j < nf) ? A[(size_t)(k0 + c) * nf + j] : 0.0;
  }
  const int lr = lane & 15, lk = lane >> 4;
  // the wave's W rows as MFMA A operands: av[rt][s] = W[32 wave + 16 rt + lr][4 s + lk]
  double av[2][KH / 4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int s2 = 0; s2 < KH / 4; ++s2) av[rt][s2] = W[((size_t)s2 * KH + 32 * wave + 16 * rt + lr) * 4 + lk];  // W[row][4 s2 + lk], k-step major (fc_fe_pivot_huge)
  __syncthreads();
  fc_d4 acc[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[rt][c] = fc_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s2 = 0; s2 < KH / 4; ++s2) acc[rt][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[rt][s2], Bs[4 * s2 + lk][16 * c + lr], acc[rt][c], 0, 0, 0);
    }
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = j0 + 16 * c + lr;
      if (j >= nf) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 32 * wave + 16 * rt + lk + 4 * r;  // row of the pivot block
        if (row >= kb) continue;
        A[(size_t)(k0 + row) * nf + j] = (j >= k0 && j < k0 + kb) ? W[((size_t)((j - k0) >> 2) * KH + row) * 4 + ((j - k0) & 3)] : acc[rt][c][r];
      }
    }
}

// 64 x 64 tile of the trailing update with 128 pivot columns (fc_fe_update without the look-ahead): the B panel (128 x 64) goes
// through LDS in two halves of 64 pivot rows, the tile's Cs rows stay in registers.
__global__ __launch_bounds__(256) void fc_fe_update_huge(const FcFront* __restrict__ nodes, double* fronts, const double* __restrict__ scratch,
                                                         int step, int tiles_per_side, int keep0 = 0, int keep1 = 0x7fffffff) {
  __shared__ double Bs[64][64];
  constexpr int KH = FC_FE_KH;
  const FcFront nd = nodes[blockIdx.y];
  const int k0 = step * KH;
  if (k0 >= nd.ni) return;
  const int kb = nd.ni - k0 < KH ? nd.ni - k0 : KH;
  const int nf = nd.nf;
  const int ti = (int)blockIdx.x / tiles_per_side, tj = (int)blockIdx.x % tiles_per_side;
  const int i0 = ti * 64, j0 = tj * 64;
  if (i0 >= nf || j0 >= nf || fc_fe_dead_rows(i0, k0, keep0, keep1)) return;
  double* A = fronts + nd.front;
  const double* Cs = scratch + nd.scratch + KH * KH;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int lr = lane & 15, lk = lane >> 4;
  const int arow = i0 + 16 * wave + lr;
  double av[KH / 4];
#pragma unroll
  for (int s2 = 0; s2 < KH / 4; ++s2) av[s2] = arow < nf ? Cs[((size_t)s2 * nf + arow) * 4 + lk] : 0.0;  // Cs[arow][4 s2 + lk], k-step major; zero beyond kb (fc_fe_panels_huge)
  double cv[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int col = j0 + 16 * c + lr;
    const bool inK = col >= k0 && col < k0 + kb;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + 16 * wave + lk + 4 * r;
      const bool live = col < nf && row < nf && !(row >= k0 && row < k0 + kb);
      cv[c][r] = (live && !inK) ? A[(size_t)row * nf + col] : 0.0;
    }
  }
  fc_d4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = fc_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    double bq[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = t + 256 * q, k = 64 * half + e / 64, col = j0 + e % 64;
      bq[q] = (k < kb && col < nf) ? A[(size_t)(k0 + k) * nf + col] : 0.0;
    }
    if (half) __syncthreads();  // the first half's MFMAs have read Bs
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = t + 256 * q;
      Bs[e / 64][e % 64] = bq[q];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[16 * half + s2], Bs[4 * s2 + lk][16 * c + lr], acc[c], 0, 0, 0);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int col = j0 + 16 * c + lr;
    if (col >= nf) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + 16 * wave + lk + 4 * r;
      if (row >= nf || (row >= k0 && row < k0 + kb)) continue;  // the pivot rows are final (fc_fe_panels_huge)
      A[(size_t)row * nf + col] = cv[c][r] - acc[c][r];
    }
  }
}

// factor rows [D^-1 | -U] (stride nf) and the -L block (nb x ni, stride ni) into the layout the sweeps read, in the
// storage type of the slot (double, or rounded once to float / bfloat16: compressed factors, fc_kernels.hip.h).
// ONE launch for all fronts of all levels at the end of the factorisation (the fronts stay intact: an extend-add only reads a
// child) over a flat work list.  Of the pivot rows of front `root_front` only [xr0_root, xr1_root) are written, row xr0_root first
// (the root of a multi-GPU layout: the handle's block, fc_set_root_rows; root_front = -1: every front is written in full).
// work list of the export: one entry per 16 rows of a front
struct __attribute__((aligned(8))) FcExpItem {
  int front;  // index into the front table
  int i0;     // first row
};
template <typename VT>
__global__ __launch_bounds__(256) void fc_fe_export(const FcFront* __restrict__ nodes, const FcExpItem* __restrict__ items, const double* __restrict__ fronts,
                                                    VT* __restrict__ fvals, int root_front, int xr0_root, int xr1_root) {
  const FcExpItem it = items[blockIdx.x];
  const FcFront nd = nodes[it.front];
  const int nf = nd.nf, ni = nd.ni;
  const int i0 = it.i0;
  const int xr0 = it.front == root_front ? xr0_root : 0, xr1 = it.front == root_front ? xr1_root : INT_MAX;
  const double* A = fronts + nd.front;
  VT* dv = fvals + nd.voff;
  const int stored = (xr1 < ni ? xr1 : ni) - xr0;  // pivot rows with storage
  VT* mw = dv + (size_t)stored * nf;
  for (int i = i0; i < i0 + 16 && i < nf; ++i) {
    if (i < ni) {
      if (i < xr0 || i >= xr1) continue;
      for (int j = threadIdx.x; j < nf; j += 256) dv[(size_t)(i - xr0) * nf + j] = fc_pack<VT>(j < ni ? A[(size_t)i * nf + j] : -A[(size_t)i * nf + j]);
    } else {
      for (int j = threadIdx.x; j < ni; j += 256) mw[(size_t)(i - ni) * ni + j] = fc_pack<VT>(A[(size_t)i * nf + j]);
    }
  }
}
