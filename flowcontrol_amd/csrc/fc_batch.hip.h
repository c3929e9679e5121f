// Shared-operator batched stepping: k lock-step simulations on ONE handle (gfx950 / CDNA4, wave64).
// Included by fc_hip.hip only.
//
// The reference's outer workloads are k runs of the SAME operator that differ in initial condition, control
// input or controller (IC sweeps examples/lidcavity/batch_run_lidcavity.py:197-215, controller optimisation
// utils/optim.py:95-102): the factors (O1: 176 MB) are the same for all of them.  A single run re-reads them
// every step to advance ONE right-hand side; here every vector becomes a row-major matrix [row][KB]
// (KB = 4, 8, 16 or 32 simulations side by side, simulation index fastest) and every level of the factor sweep a
// dense block product  (factor block) x (operand rows x KB)  on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64): the factor bytes are streamed once per KB simulated steps.
//
//   fc_nd_block_b   one workgroup per task = a 16-row tile of ONE dense factor block of one tree node
//                   ([D^-1 | -U] rows in the down-sweep, -L rows in the up-sweep) times the node's operand rows;
//                   CG waves split the tile's columns, their accumulators are summed through LDS in a fixed order
//   fc_nd_fold_b    up-sweep: a tree node's -L block writes its products to a private scratch row per
//                   (node, boundary row); the rows of the next level add the scratch rows of their descendants in
//                   a fixed order (bit-reproducible, no fp64 atomics)
//   fc_rhs_elem_b / fc_rhs_gather_b / fc_tail_b / fc_final_b   the step kernels of fc_kernels.hip.h with the
//                   simulation index as the fastest-running one (cell tables, gather lists, matrix rows and sensor
//                   rows are read once for all KB simulations)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct __attribute__((aligned(16))) FcBTask {
  long long src;  // offset of the tile's first value in the factor values as the single-simulation sweeps store them (row-major, stride ld)
  long long val;  // offset of the tile in the TILED copy the batched kernel streams (fc_b_repack)
  int ld;         // row stride of the block in the row-major layout
  int nrows;      // rows of this tile (<= 16)
  int ncols;      // columns of the block = operand rows of the node
  int op;         // offset of the node's operand row list (buffer row of every column; multiple of 8)
  int dst;        // first destination buffer row
  int split;      // 0: the task is a whole tile.  Else part | parts << 8: the task covers a run of the tile's 32-column chunks (src / val / op / ncols
                  // describe that run), its products are a PARTIAL sum: the tile's last-arriving part adds the parts up in part order
  int pslot;      // first of the tile's `parts` partial slots (16 x KB doubles each); its arrival counter is ticket[pslot]
  int pad;
};

typedef int fc_i2 __attribute__((ext_vector_type(2)));
typedef double fc_d2 __attribute__((ext_vector_type(2)));

// Tiled copy of the factor values for the batched kernel: a tile (16 rows of one dense block) is a sequence of 32-column
// chunks of 4 KB, every chunk stored in the order the matrix instructions consume it — group u of 8 columns, then lane
// (row = lane & 15, q = lane >> 4), then the value pair at columns 32 c + 8 u + 2 q, + 1 — so that one wave instruction
// (16 B per lane) reads 1 KB of consecutive memory and a chunk 4 KB.  Rows past the tile's last one and columns past the
// block's last one hold zeros.  One workgroup per tile; re-run after every numeric factorisation (fc_refactor).
__global__ __launch_bounds__(256) void fc_b_repack(const FcBTask* __restrict__ tasks, const double* __restrict__ val, double* __restrict__ tiled) {
  const FcBTask tk = tasks[blockIdx.x];
  const int nchunk = (tk.ncols + 31) >> 5;
  for (int e = threadIdx.x; e < nchunk * 256; e += 256) {  // one value pair per trip
    const int c = e >> 8, u = (e >> 6) & 3, lane = e & 63;
    const int row = lane & 15, col = 32 * c + 8 * u + 2 * (lane >> 4);
    fc_d2 v = {0.0, 0.0};
    if (row < tk.nrows) {
      const double* __restrict__ p = val + tk.src + (long long)row * tk.ld + col;
      if (col < tk.ncols) v.x = p[0];
      if (col + 1 < tk.ncols) v.y = p[1];
    }
    *reinterpret_cast<fc_d2*>(tiled + tk.val + 2 * (long long)e) = v;
  }
}

// One workgroup per task = one 16-row tile of ONE dense factor block of a tree node, CG <= 16 waves (blockDim = 64 CG): wave g
// takes the 32-column chunks g, g + CG, ... of the tile, the CG accumulators are summed through LDS in the order
// of the groups (reproducible).  No barrier before that sum: every wave streams on its own through a three-deep register
// pipeline  operand-row indices (chunk c + 3) | values + operand rows (chunk c + 2, c + 1 in flight) | matrix
// instructions (chunk c).  The factor values come from HBM / Infinity Cache (each exactly once per apply, 4 KB of
// consecutive memory per chunk), the operand rows [row][KB] from L2 (a node's rows are shared by all of its tiles).
//
// v_mfma_f64_16x16x4_f64: lane l holds A[row l & 15][k = l >> 4], B[k = l >> 4][col l & 15];
// D[row (l >> 4) + 4 r][col l & 15] in register r.  The k index of an instruction may stand for any four columns as
// long as A and B agree: lane (row, q) holds the value PAIR at columns 8 u + 2 q, 8 u + 2 q + 1 of its row and feeds the
// pair to two consecutive instructions.
#ifndef FC_B_WPE
#define FC_B_WPE 4  // waves per SIMD the register allocation aims at: 128 VGPRs keep two chunks of a wave in flight
#endif
template <int KB, bool NT = false>  // NT: the tiled factors exceed the Infinity Cache and are streamed with nontemporal loads (fc_ld in fc_kernels.hip.h)
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(FC_B_WPE, FC_B_WPE))) void fc_nd_block_b(
    const FcBTask* __restrict__ tasks, const int* __restrict__ olist, const double* __restrict__ tiled, double* __restrict__ buf, int CG,
    const unsigned char* __restrict__ velrow = nullptr, int N = 0, int* __restrict__ flag = nullptr, double* __restrict__ part = nullptr,
    unsigned* __restrict__ ticket = nullptr) {
  // part / ticket (split tiles, FcBTask::split): the few wide tiles of the levels near the root are cut into runs of chunks, one workgroup
  // each, so that those launches fill all compute units instead of one per tile (O1 root: 143 tiles of 72 chunks)
  // velrow (overlapped tail): a tile that writes solution rows (buffer rows N .. 2N: the down-sweep) tests what it writes for finiteness,
  // velocity rows only, and raises flag[simulation] -- the reference's test (flowsolver.py:731,816-819) without a pass of its own
  // KB = 32: the matrix instruction's tile is 16 simulations wide, so a wave keeps TWO accumulators (even / odd simulations) and feeds
  // every value pair it loaded to both: the factor values -- and the launch floors -- are shared by 32 simulated steps
  constexpr int NH = KB > 16 ? 2 : 1;
  extern __shared__ double fc_b_red[];
  const FcBTask tk = tasks[blockIdx.x];
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const double* __restrict__ tv = tiled + tk.val + 2 * lane;
  const int* __restrict__ orow = olist + 2 * lq;
  const int ls = lr < KB ? lr : 0;  // lanes beyond the batch width shadow simulation 0 (their output columns are never stored)
  fc_d4 acc[NH];
#pragma unroll
  for (int hh = 0; hh < NH; ++hh) acc[hh] = fc_d4{0.0, 0.0, 0.0, 0.0};
  // chunk c of this wave is chunk grp + CG c of the tile.  The pipeline below has NO branch and no per-lane predicate in
  // its body (a branch makes the compiler drain the load queue at every join): the tiled values are zero past the block's
  // edges, a node's operand list is padded to a multiple of 32 with the index of a buffer row that is always zero, and the
  // up to two chunks by which a wave's chunk count is rounded up to the pipeline's period read chunk 0 of the tile (one
  // valid 4 KB) against the "null group" olist[0..8) (the zero row again).  Everything that depends on the chunk index
  // only is wave-uniform (scalar).
  const int nchunk = (tk.ncols + 31) >> 5;
  const int nw = nchunk > grp ? (nchunk - grp + CG - 1) / CG : 0;  // chunks of this wave
  auto I = [&](int c, fc_i2 (&x)[4]) {  // operand-row indices of chunk c
    const int ch = grp + CG * c;
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const fc_i2*>(orow + (ch < nchunk ? tk.op + 32 * ch + 8 * u : 0));
  };
  auto L = [&](int c, const fc_i2 (&x)[4], double (&a)[8], double (&b)[8 * NH]) {  // values and operand rows of chunk c
    const int ch = grp + CG * c;
    const double* __restrict__ p = tv + 512 * (long long)(ch < nchunk ? ch : 0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const fc_d2 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const fc_d2*>(p + 128 * u)) : *reinterpret_cast<const fc_d2*>(p + 128 * u);
      a[2 * u] = v.x;
      a[2 * u + 1] = v.y;
      if constexpr (NH == 1) {
        b[2 * u] = buf[(size_t)x[u].x * KB + ls];
        b[2 * u + 1] = buf[(size_t)x[u].y * KB + ls];
      } else {
        // accumulator hh of lane column lr stands for simulation 2 lr + hh: both operand values of a row come with ONE 16-byte load
        const fc_d2 p0 = *reinterpret_cast<const fc_d2*>(buf + (size_t)x[u].x * KB + 2 * lr);
        const fc_d2 p1 = *reinterpret_cast<const fc_d2*>(buf + (size_t)x[u].y * KB + 2 * lr);
        b[2 * u] = p0.x;
        b[8 + 2 * u] = p0.y;
        b[2 * u + 1] = p1.x;
        b[8 + 2 * u + 1] = p1.y;
      }
    }
  };
  auto M = [&](const double (&a)[8], const double (&b)[8 * NH]) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) acc[hh] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[8 * hh + u], acc[hh], 0, 0, 0);
  };
  if constexpr (NH == 1) {
    fc_i2 x0[4], x1[4], x2[4];
    double a0[8], b0[8], a1[8], b1[8], a2[8], b2[8];
    if (nw == 1) {  // the small blocks near the leaves: one round trip per stage, nothing to overlap
      I(0, x0);
      L(0, x0, a0, b0);
      M(a0, b0);
    } else if (nw == 2) {
      I(0, x0);
      I(1, x1);
      L(0, x0, a0, b0);
      L(1, x1, a1, b1);
      M(a0, b0);
      M(a1, b1);
    } else if (nw > 2) {
      I(0, x0);
      I(1, x1);
      I(2, x2);
      L(0, x0, a0, b0);
      L(1, x1, a1, b1);
      for (int c = 0; c < nw; c += 3) {
        I(c + 3, x0);
        L(c + 2, x2, a2, b2);
        M(a0, b0);
        I(c + 4, x1);
        L(c + 3, x0, a0, b0);
        M(a1, b1);
        I(c + 5, x2);
        L(c + 4, x1, a1, b1);
        M(a2, b2);
      }
    }
  } else {
    // two accumulators and twice the operand rows per chunk: ONE chunk of values and operands in registers at a time (the register file of
    // a 1024-thread workgroup allows 128 per lane), the next chunk's row indices fetched ahead; the other waves of the SIMD cover the round trips
    fc_i2 x0[4], x1[4];
    double a0[8], b0[8 * NH];
    if (nw > 0) I(0, x0);
    for (int c = 0; c < nw; c += 2) {
      I(c + 1, x1);
      L(c, x0, a0, b0);
      M(a0, b0);
      I(c + 2, x0);
      L(c + 1, x1, a0, b0);
      M(a0, b0);
    }
  }
  if (CG > 1) {
    const int wave = (int)threadIdx.x >> 6;
#pragma unroll
    for (int hh = 0; hh < NH; ++hh)
#pragma unroll
      for (int r = 0; r < 4; ++r) fc_b_red[((wave * NH + hh) * 4 + r) * 64 + lane] = acc[hh][r];
    __syncthreads();
    if (wave != 0) return;
    for (int g = 1; g < CG; ++g)
#pragma unroll
      for (int hh = 0; hh < NH; ++hh)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[hh][r] += fc_b_red[((g * NH + hh) * 4 + r) * 64 + lane];
  }
  if (tk.split) {
    // (wave 0 only from here.)  The partial goes to the coherence point (sc1 stores, drained), then the part arrives on the tile's counter;
    // the last arriver reads all parts back -- its own too -- and adds them in part order: the sum does not depend on who arrives when
    const int me = tk.split & 255, np = tk.split >> 8;
    double* __restrict__ mine = part + (size_t)(tk.pslot + me) * (NH * 256);
#pragma unroll
    for (int hh = 0; hh < NH; ++hh)
#pragma unroll
      for (int r = 0; r < 4; ++r) fc_st_sc1(mine + (hh * 4 + r) * 64 + lane, acc[hh][r]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int last = 0;
    if (lane == 0) {
      last = __hip_atomic_fetch_add(ticket + tk.pslot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)np - 1u ? 1 : 0;
      if (last) __hip_atomic_store(ticket + tk.pslot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!__shfl(last, 0)) return;
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) acc[hh] = fc_d4{0.0, 0.0, 0.0, 0.0};
    for (int g0 = 0; g0 < np; g0 += 4) {  // four parts' loads in flight at a time (a part per round trip would cost np round trips)
      double v[4][NH * 4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double* __restrict__ src = part + (size_t)(tk.pslot + (g0 + u < np ? g0 + u : np - 1)) * (NH * 256);
#pragma unroll
        for (int q = 0; q < NH * 4; ++q) v[u][q] = fc_ld_sc1(src + q * 64 + lane);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (g0 + u < np) {
#pragma unroll
          for (int hh = 0; hh < NH; ++hh)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[hh][r] += v[u][hh * 4 + r];
        }
    }
  }
#pragma unroll
  for (int hh = 0; hh < NH; ++hh) {
    const int sim = NH == 1 ? lr : 2 * lr + hh;
    if (sim < KB) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = lq + 4 * r;
        if (o < tk.nrows) {
          buf[(size_t)(tk.dst + o) * KB + sim] = acc[hh][r];
          const int d = tk.dst + o - N;
          if (velrow && d >= 0 && d < N && velrow[d] && !isfinite(acc[hh][r])) atomicOr(flag + sim, 1);
        }
      }
    }
  }
}

// rows [row0, row0 + nrows) of the buffer half at dst_off: (+)= the sum of their source rows, in list order
template <int KB>
__global__ __launch_bounds__(256) void fc_nd_fold_b(int nrows, int row0, const int* __restrict__ fptr, const int* __restrict__ fsrc,
                                                    double* __restrict__ buf, int dst_off, int accumulate) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int r = t / KB, s = t % KB;
  if (r >= nrows) return;
  const int i = row0 + r;
  double* d = buf + (size_t)(dst_off + i) * KB + s;
  const int q0 = fptr[i], q1 = fptr[i + 1];
  double acc = accumulate ? *d : 0.0;
  for (int base = q0; base < q1; base += 4) {
    int id[4];
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) id[u] = base + u < q1 ? fsrc[base + u] : -1;
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = id[u] >= 0 ? buf[(size_t)id[u] * KB + s] : 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (id[u] >= 0) acc += v[u];
  }
  *d = acc;
}

// ---------------------------------------------------------------------------------------------
// step kernels, simulation index fastest.  State u_n / u_nn [2 nn][KB], p_n [nv][KB], controls
// uctrl[s * ustride + a] (host-mapped record of simulation s).
// ---------------------------------------------------------------------------------------------
// fc_rhs_elem for KB simulations, thread = (cell, lane8, simulation s) with s fastest.  The element loop is bound by the
// number of vector-memory instructions, not by bytes: a cell's nodal values are therefore loaded ONCE, by thread
// (cell, node a = lane8 < 6, s) — KB contiguous doubles per node and field — and shared through LDS together with the
// basis tables; thread (cell, Radon point q = lane8 < 7, s) evaluates the integrand at its point from LDS, and thread
// (cell, a < 6, s) tests the weighted point values with phi_a in the fixed order q = 0..6.
// ev[(slot * nc + cell) * KB + s]
// (un / unn: the batched state in the solver's permuted numbering -- solution halves of the batched work buffers, "state ring";
//  cnp[a][c] / cnp[6 + a][c] = permuted position of the x- / y-velocity dof of node a of cell c; cn only addresses fprof)
template <int KB>
__global__ __launch_bounds__(256) void fc_rhs_elem_b(int nc, int nn, const int* __restrict__ cn, const int* __restrict__ cnp,
                                                     const double* __restrict__ geom,
                                                     const double* __restrict__ un, const double* __restrict__ unn,
                                                     const double* __restrict__ fprof, int n_act, const double* __restrict__ uforce,
                                                     int ustride, double cm_n, double cm_nn, double cc_n, double cc_nn,
                                                     double* __restrict__ ev) {
  // thread = (cell, lane8, simulation PAIR): every nodal load and every store moves 16 B (the loop is bound by the number of vector-memory
  // instructions); the two simulations of a pair go through the same operations in the same order as a single run's
  typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
  constexpr int HP = KB / 2;            // simulation pairs
  constexpr int CPB = 256 / (8 * HP);   // cells per workgroup
  __shared__ d2 nod[6][256];            // fields (ux, uy of u_n; ux, uy of u_nn; fx, fy) x [cell][node 8][pair]
  __shared__ d2 gs[2][256];
  __shared__ double tph[FC_NQ * 6], tdx[FC_NQ * 6], tde[FC_NQ * 6];
  const int t = threadIdx.x;
  const int sp = t % HP, lane = (t / HP) % 8, cw = t / (8 * HP);
  const int s = 2 * sp;
  const int cl = blockIdx.x * CPB + cw;
  const bool active = cl < nc;
  const int c = active ? cl : 0;
  if (t < FC_NQ * 6) {
    tph[t] = c_phi2[t];
    tdx[t] = c_dphi2[2 * t];
    tde[t] = c_dphi2[2 * t + 1];
  }
  {
    const int a = lane < 6 ? lane : 0;
    const int ix = cnp[a * nc + c], iy = cnp[(6 + a) * nc + c];
    d2 fx = {0.0, 0.0}, fy = {0.0, 0.0};
    if (n_act > 0) {
      const int n = cn[a * nc + c];
      for (int k = 0; k < n_act; ++k) {
        const d2 uk = {uforce[s * ustride + k], uforce[(s + 1) * ustride + k]};
        fx += uk * fprof[(size_t)k * 2 * nn + n];
        fy += uk * fprof[(size_t)k * 2 * nn + nn + n];
      }
    }
    nod[0][t] = *reinterpret_cast<const d2*>(un + (size_t)ix * KB + s);
    nod[1][t] = *reinterpret_cast<const d2*>(un + (size_t)iy * KB + s);
    nod[2][t] = *reinterpret_cast<const d2*>(unn + (size_t)ix * KB + s);
    nod[3][t] = *reinterpret_cast<const d2*>(unn + (size_t)iy * KB + s);
    nod[4][t] = fx;
    nod[5][t] = fy;
  }
  const int q = lane < FC_NQ ? lane : FC_NQ - 1;
  const double j00 = geom[c], j01 = geom[nc + c], j10 = geom[2 * nc + c], j11 = geom[3 * nc + c];
  const double wq = lane < FC_NQ ? c_qw[q] * 0.5 * geom[4 * nc + c] : 0.0;
  __syncthreads();
  const d2 z = {0.0, 0.0};
  d2 ux = z, uy = z, uxi = z, uet = z, vxi = z, vet = z;
  d2 wx = z, wy = z, wxi = z, wet = z, zxi = z, zet = z;
  d2 gx = z, gy = z;
  const int nb = cw * 8 * HP + sp;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const d2 ax = nod[0][nb + a * HP], ay = nod[1][nb + a * HP], bx = nod[2][nb + a * HP], by = nod[3][nb + a * HP];
    const double ph = tph[q * 6 + a], dx = tdx[q * 6 + a], de = tde[q * 6 + a];
    ux += ph * ax;
    uy += ph * ay;
    uxi += dx * ax;
    uet += de * ax;
    vxi += dx * ay;
    vet += de * ay;
    wx += ph * bx;
    wy += ph * by;
    wxi += dx * bx;
    wet += de * bx;
    zxi += dx * by;
    zet += de * by;
    gx += ph * nod[4][nb + a * HP];
    gy += ph * nod[5][nb + a * HP];
  }
  const d2 ux_x = uxi * j00 + uet * j10, ux_y = uxi * j01 + uet * j11;
  const d2 uy_x = vxi * j00 + vet * j10, uy_y = vxi * j01 + vet * j11;
  const d2 wx_x = wxi * j00 + wet * j10, wx_y = wxi * j01 + wet * j11;
  const d2 wy_x = zxi * j00 + zet * j10, wy_y = zxi * j01 + zet * j11;
  gx += cm_n * ux + cm_nn * wx + cc_n * (ux * ux_x + uy * ux_y) + cc_nn * (wx * wx_x + wy * wx_y);
  gy += cm_n * uy + cm_nn * wy + cc_n * (ux * uy_x + uy * uy_y) + cc_nn * (wx * wy_x + wy * wy_y);
  gs[0][t] = gx * wq;
  gs[1][t] = gy * wq;
  __syncthreads();
  if (active && lane < 6) {
    d2 accx = z, accy = z;
#pragma unroll
    for (int p = 0; p < FC_NQ; ++p) {
      const double pa = tph[p * 6 + lane];
      accx += pa * gs[0][nb + p * HP];
      accy += pa * gs[1][nb + p * HP];
    }
    *reinterpret_cast<d2*>(ev + ((size_t)lane * nc + c) * KB + s) = accx;
    *reinterpret_cast<d2*>(ev + ((size_t)(6 + lane) * nc + c) * KB + s) = accy;
  }
}

// The same element loop with ALL of a cell's work on one thread = (cell, simulation): the 24 nodal values of the thread's simulation sit in
// registers, the basis tables come through the scalar unit (compile-time indices into constant memory), there is no LDS and no barrier.
// fc_rhs_elem_b above shares the nodal values of a cell between its 7 point threads through LDS -- every thread reads all of them back
// (75 LDS instructions per thread, 8 clocks each at 16 B per lane): at KB = 32 that is ~20 us of LDS issue alone.  Here the cost is the
// ~870 fp64 FMAs per (cell, simulation), i.e. 8.7 us at KB = 32 on full-rate vector units.  Same sums in the same order (a = 0..5 inside a
// point, points 0..6 into the test functions).  FORCE: body-force profiles present (their nodal values take 12 more registers).
template <int KB, bool FORCE>
__global__ __launch_bounds__(256) void fc_rhs_elem_breg(int nc, int nn, const int* __restrict__ cn, const int* __restrict__ cnp,
                                                        const double* __restrict__ geom,
                                                        const double* __restrict__ un, const double* __restrict__ unn,
                                                        const double* __restrict__ fprof, int n_act, const double* __restrict__ uforce,
                                                        int ustride, double cm_n, double cm_nn, double cc_n, double cc_nn,
                                                        double* __restrict__ ev) {
  constexpr int CPB = 256 / KB;
  const int s = (int)threadIdx.x % KB, c = blockIdx.x * CPB + (int)threadIdx.x / KB;
  if (c >= nc) return;
  double ax[6], ay[6], bx[6], by[6], fx[FORCE ? 6 : 1], fy[FORCE ? 6 : 1];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const int ix = cnp[a * nc + c], iy = cnp[(6 + a) * nc + c];
    ax[a] = un[(size_t)ix * KB + s];
    ay[a] = un[(size_t)iy * KB + s];
    bx[a] = unn[(size_t)ix * KB + s];
    by[a] = unn[(size_t)iy * KB + s];
    if constexpr (FORCE) {
      const int n = cn[a * nc + c];
      double f0 = 0.0, f1 = 0.0;
      for (int k = 0; k < n_act; ++k) {
        const double uk = uforce[s * ustride + k];
        f0 += uk * fprof[(size_t)k * 2 * nn + n];
        f1 += uk * fprof[(size_t)k * 2 * nn + nn + n];
      }
      fx[a] = f0, fy[a] = f1;
    }
  }
  const double j00 = geom[c], j01 = geom[nc + c], j10 = geom[2 * nc + c], j11 = geom[3 * nc + c], hdet = 0.5 * geom[4 * nc + c];
  double accx[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, accy[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < FC_NQ; ++q) {
    double ux = 0.0, uy = 0.0, uxi = 0.0, uet = 0.0, vxi = 0.0, vet = 0.0;
    double wx = 0.0, wy = 0.0, wxi = 0.0, wet = 0.0, zxi = 0.0, zet = 0.0;
    double gx = 0.0, gy = 0.0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const double ph = c_phi2[q * 6 + a], dx = c_dphi2[2 * (q * 6 + a)], de = c_dphi2[2 * (q * 6 + a) + 1];
      ux += ph * ax[a];
      uy += ph * ay[a];
      uxi += dx * ax[a];
      uet += de * ax[a];
      vxi += dx * ay[a];
      vet += de * ay[a];
      wx += ph * bx[a];
      wy += ph * by[a];
      wxi += dx * bx[a];
      wet += de * bx[a];
      zxi += dx * by[a];
      zet += de * by[a];
      if constexpr (FORCE) {
        gx += ph * fx[a];
        gy += ph * fy[a];
      }
    }
    const double ux_x = uxi * j00 + uet * j10, ux_y = uxi * j01 + uet * j11;
    const double uy_x = vxi * j00 + vet * j10, uy_y = vxi * j01 + vet * j11;
    const double wx_x = wxi * j00 + wet * j10, wx_y = wxi * j01 + wet * j11;
    const double wy_x = zxi * j00 + zet * j10, wy_y = zxi * j01 + zet * j11;
    gx += cm_n * ux + cm_nn * wx + cc_n * (ux * ux_x + uy * ux_y) + cc_nn * (wx * wx_x + wy * wx_y);
    gy += cm_n * uy + cm_nn * wy + cc_n * (ux * uy_x + uy * uy_y) + cc_nn * (wx * wy_x + wy * wy_y);
    const double wq = c_qw[q] * hdet;
    gx *= wq;
    gy *= wq;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      accx[a] += c_phi2[q * 6 + a] * gx;
      accy[a] += c_phi2[q * 6 + a] * gy;
    }
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    ev[((size_t)a * nc + c) * KB + s] = accx[a];
    ev[((size_t)(6 + a) * nc + c) * KB + s] = accy[a];
  }
}

// fc_rhs_gather, one thread per (permuted row, simulation PAIR): every load moves 16 B
template <int KB>
__global__ __launch_bounds__(256) void fc_rhs_gather_b(int N, const int* __restrict__ gptr, const int* __restrict__ gidx,
                                                       const double* __restrict__ ev, const int* __restrict__ bcslot,
                                                       const double* __restrict__ bcprof, const double* __restrict__ lift, int n_act,
                                                       const double* __restrict__ uctrl, int ustride, double* __restrict__ b,
                                                       double* __restrict__ y, const int* __restrict__ c_rowptr,
                                                       const int* __restrict__ c_col, const double* __restrict__ c_val,
                                                       const double* __restrict__ un, int with_ctrl,
                                                       unsigned long long* __restrict__ solved = nullptr, const double* __restrict__ seq_in = nullptr) {
  // solved (the gather that runs AHEAD for the next step, behind fc_early_b and the element loop): this launch opens the side stream's
  // gate instead of fc_early_b -- the late tail then runs beside this gather and the next step's first (latency-bound) sweep launches
  // rather than beside the element loop, which is compute-bound and was slowed from 16.5 to 30 us by it (k = 16, O1).  Correctness
  // needs the gate no earlier than fc_early_b; when it opens afterwards is a matter of speed only.
  if (solved && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(solved, (unsigned long long)seq_in[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  // with_ctrl = 0: the part of the right-hand side that does not depend on the controls -- what the previous step's launches can
  // already assemble for this one while the host still computes u_ctrl (fc_rhs_ctrl_b adds the rest on the few rows it touches)
  typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
  constexpr int HP = KB / 2;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t / HP, s = 2 * (t % HP);
  if (i >= N) return;
  double a0 = 0.0, a1 = 0.0;
  const int bs = bcslot[i];
  if (bs >= 0 && !with_ctrl) {
  } else if (bs >= 0) {
    for (int k = 0; k < n_act; ++k) {
      const double p = bcprof[(size_t)bs * n_act + k];
      a0 += uctrl[s * ustride + k] * p;
      a1 += uctrl[(s + 1) * ustride + k] * p;
    }
  } else {
    const int k0 = gptr[i], k1 = gptr[i + 1];
    for (int base = k0; base < k1; base += 8) {
      int id[8];
      d2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) id[u] = gidx[base + u < k1 ? base + u : k1 - 1];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const d2*>(ev + (size_t)id[u] * KB + s);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (base + u < k1) {
          a0 += v[u].x;
          a1 += v[u].y;
        }
    }
    if (with_ctrl)
      for (int k = 0; k < n_act; ++k) {
        const double l = lift[(size_t)k * N + i];
        a0 -= uctrl[s * ustride + k] * l;
        a1 -= uctrl[(s + 1) * ustride + k] * l;
      }
    if (c_rowptr)
      for (int k = c_rowptr[i]; k < c_rowptr[i + 1]; ++k) {
        const d2 u2 = *reinterpret_cast<const d2*>(un + (size_t)c_col[k] * KB + s);
        a0 -= c_val[k] * u2.x;
        a1 -= c_val[k] * u2.y;
      }
  }
  const d2 out = {a0, a1};
  *reinterpret_cast<d2*>(b + (size_t)i * KB + s) = out;
  *reinterpret_cast<d2*>(y + (size_t)i * KB + s) = out;
}

// the control-dependent rest of a right-hand side that fc_rhs_gather_b assembled with_ctrl = 0: the Dirichlet rows (value = profile . u_ctrl)
// and the rows next to them (- lifting vector . u_ctrl), listed once per slot (rows[]); same operations in the same order as the
// full gather, so the result is bit-identical.  thread = (listed row, simulation pair)
template <int KB>
__global__ __launch_bounds__(256) void fc_rhs_ctrl_b(int n_rows, const int* __restrict__ rows, int N, const int* __restrict__ bcslot,
                                                     const double* __restrict__ bcprof, const double* __restrict__ lift, int n_act,
                                                     const double* __restrict__ uctrl, int ustride, double* __restrict__ b, double* __restrict__ y) {
  typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
  constexpr int HP = KB / 2;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int q = t / HP, s = 2 * (t % HP);
  if (q >= n_rows) return;
  const int i = rows[q];
  const int bs = bcslot[i];
  double a0 = 0.0, a1 = 0.0;
  if (bs >= 0) {
    for (int k = 0; k < n_act; ++k) {
      const double p = bcprof[(size_t)bs * n_act + k];
      a0 += uctrl[s * ustride + k] * p;
      a1 += uctrl[(s + 1) * ustride + k] * p;
    }
  } else {
    const d2 v = *reinterpret_cast<const d2*>(b + (size_t)i * KB + s);
    a0 = v.x, a1 = v.y;
    for (int k = 0; k < n_act; ++k) {
      const double l = lift[(size_t)k * N + i];
      a0 -= uctrl[s * ustride + k] * l;
      a1 -= uctrl[(s + 1) * ustride + k] * l;
    }
  }
  const d2 out = {a0, a1};
  *reinterpret_cast<d2*>(b + (size_t)i * KB + s) = out;
  *reinterpret_cast<d2*>(y + (size_t)i * KB + s) = out;
}

// fc_tail for KB simulations: residual monitor r = b - A x of every simulation, non-finite flags, energy.  The new state needs
// no work -- the solution half of the batched work buffer IS the new state (permuted numbering, rotating buffers: fc_hip.hip
// "state ring"); rounds up to 3 scattered it to W-layout arrays and shifted three time levels here, 20 of the kernel's 50 us at
// KB = 16.  The matrix rows in nested-dissection order come in runs that touch the same few solution
// rows (a tree node's rows couple to the node and its boundary), so a ROW BLOCK — up to 16 consecutive permuted rows with
// at most tb_cols (128 or 256: build_batch_tables) distinct columns, tabulated once per pattern — brings its distinct solution rows [col][KB] to LDS
// once (coalesced KB-wide rows, instead of one 8 KB-byte gather per matrix entry and simulation pair) and evaluates the
// 16 rows from there: 16 lanes per row = (j: 16 / HP) x (simulation pair: HP = KB / 2), each lane walks every
// (16 / HP)-th entry of its row with the entry's LOCAL column (uint16) and reads two simulations (16 B) from LDS.
// partial[(s * 3 + w) * G + block], w = 0: sum r^2, 1: sum b^2 (row blocks), 2: sum e (cell blocks)  (fixed order: reproducible).
#define FC_TB_ROWS 16
#define FC_TB_COLS 128  // width of a row block's column set when the tail runs alone (fc_ctx::Batch::tb_cols; dynamic LDS: tb_cols * KB doubles)
#define FC_TB_NNZ 768   // matrix entries of a row block at most (staged in LDS: 10 B each); build_tail_blocks cuts the blocks accordingly
inline size_t fc_tail_b_lds(int tb_cols, int KB) { return (size_t)tb_cols * KB * sizeof(double) + (size_t)FC_TB_NNZ * (sizeof(double) + sizeof(unsigned short)); }
typedef double fc_d2u __attribute__((ext_vector_type(2), aligned(8)));
struct __attribute__((aligned(16))) FcTBlock {
  int row0, nrows;  // its rows: trowd[row0 .. row0 + nrows) = (permuted row, first matrix entry, length | velocity row << 16, offset in the staged range)
  int col0, ncols;  // its distinct columns: bcols[col0 .. col0 + ncols)
};
// energy of the new velocity, 1/2 int |u|^2 element by element (flowsolver.py:827-829), for KB simulations.  thread = (cell, simulation):
// 256 / KB cells side by side, FC_EB_TRIPS of them one after the other per workgroup; a thread reads the twelve nodal velocities of
// its cell for its simulation (a KB-wide row of x per node: coalesced over the simulations) and evaluates the seven Radon points
// itself -- no LDS staging, no barrier but the final fold.  (Rounds 3-4 spread a cell over 8 lanes, one Radon point each: at KB = 32
// that was ONE cell per 256-thread workgroup, 12 284 workgroups with three barriers each on O1 -- most of the 114 us of fc_tail_b<32>.)
// The cell workgroups are the FIRST n_cell_blocks of fc_tail_b's grid.  partial[(s * 3 + 2) * G + first + block]
#define FC_EB_TRIPS 4
template <int KB>
__device__ __forceinline__ void fc_energy_b_block(int cb, int nc, const int* __restrict__ cnp, const double* __restrict__ geom,
                                                  const double* __restrict__ x, double* __restrict__ partial,
                                                  int G, int first, double (&red)[2][256]) {
  constexpr int CW = 256 / KB;  // cells side by side
  const int t = threadIdx.x;
  const int s = t % KB, cw = t / KB;
  double e = 0.0;
#pragma unroll
  for (int trip = 0; trip < FC_EB_TRIPS; ++trip) {
    const int c = (cb * FC_EB_TRIPS + trip) * CW + cw;
    if (c < nc) {
      double ux[6], uy[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        ux[a] = x[(size_t)cnp[(size_t)a * nc + c] * KB + s];
        uy[a] = x[(size_t)cnp[(size_t)(6 + a) * nc + c] * KB + s];
      }
      double acc = 0.0;
#pragma unroll
      for (int q = 0; q < FC_NQ; ++q) {
        double vx = 0.0, vy = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const double ph = c_phi2[q * 6 + a];
          vx += ph * ux[a];
          vy += ph * uy[a];
        }
        acc += c_qw[q] * (vx * vx + vy * vy);
      }
      e += 0.5 * geom[4 * (size_t)nc + c] * acc;
    }
  }
  red[0][t] = e;
  __syncthreads();
  if (t < KB) {
    double se = 0.0;
#pragma unroll
    for (int g = 0; g < CW; ++g) se += red[0][g * KB + t];
    partial[((size_t)t * 3 + 2) * G + first + cb] = se;
  }
}

template <int KB>
__global__ __launch_bounds__(256) void fc_tail_b(int N, const unsigned char* __restrict__ velrow, const double* __restrict__ x,
                                                 const double* __restrict__ b, const FcTBlock* __restrict__ blocks,
                                                 const int* __restrict__ bcols, const int4* __restrict__ trowd,
                                                 const unsigned short* __restrict__ a_lidx, const double* __restrict__ a_val,
                                                 int* __restrict__ flag, double* __restrict__ partial, int G,
                                                 int n_cell_blocks, int nc, const int* __restrict__ cnp, const double* __restrict__ geom, int tb_cols) {
  constexpr int HP = KB / 2;   // simulation pairs
  constexpr int JL = 16 / HP;  // lanes of a row that split its entries
  const int t = threadIdx.x;
  extern __shared__ double xs[];  // [tb_cols][KB] | values [FC_TB_NNZ] | local columns [FC_TB_NNZ] (launch: dynamic LDS, fc_tail_b_lds)
  __shared__ double red[2][256];
  if ((int)blockIdx.x < n_cell_blocks) {
    fc_energy_b_block<KB>((int)blockIdx.x, nc, cnp, geom, x, partial, G, G - n_cell_blocks, red);
    return;
  }
  const int rbk = (int)blockIdx.x - n_cell_blocks;  // row block of this workgroup
  {
    // A row block is a short chain of dependent memory round trips and little else, so the chain is what is laid out here:
    //   trip 1 (scalar)  the block descriptor
    //   trip 2           row descriptors (row, first entry, length, offset in the staged range -- one int4 each, tabulated on the host:
    //                    no row -> row pointer -> entries chain) and the column list
    //   trip 3           everything else at once: solution rows -> LDS, the block's matrix entries -> LDS, the x / b values of the
    //                    finishing pass -> registers
    // then LDS only.  (Rounds 3-4 walked ~8 dependent trips per block: 114 us for fc_tail_b<32> on O1.)
    const FcTBlock bk = blocks[rbk];
    double* vs = xs + (size_t)tb_cols * KB;
    unsigned short* ls = reinterpret_cast<unsigned short*>(vs + FC_TB_NNZ);
    const int rl = t / 16, q16 = t % 16, j = q16 / HP, sp = t % HP;
    const bool live = rl < bk.nrows;
    constexpr int NT = (16 * KB + 255) / 256;  // trips of the finishing pass (thread = (row, simulation))
    constexpr int CT = 12;                     // trips of the staging loop at most (tb_cols * KB / 2 <= 3072 lanes' worth)
    // every load below is UNCONDITIONAL at a clamped (always valid) address and lands in a register; what is conditional is the use.
    // (Written as predicated loads straight into LDS the compiler emitted load - s_waitcnt vmcnt(0) - ds_write per element: nine
    // dependent round trips for the solution rows alone.)
    const int rlc = rl < bk.nrows ? rl : bk.nrows - 1;
    const int4 rd = trowd[bk.row0 + rlc];
    int fi[NT], fvel[NT], cid[CT];
    bool fok[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int r_l = (t + 256 * u) / KB;
      fok[u] = r_l < bk.nrows;
      const int4 d = trowd[bk.row0 + (fok[u] ? r_l : bk.nrows - 1)];
      fi[u] = d.x, fvel[u] = d.z >> 16;
    }
#pragma unroll
    for (int u = 0; u < CT; ++u) {
      const int c = t / HP + u * (256 / HP);
      cid[u] = bcols[bk.col0 + (c < bk.ncols ? c : bk.ncols - 1)];
    }
    const int g0 = rd.y, len = live ? (rd.z & 0xFFFF) : 0, k0 = rd.w, k1 = k0 + len;
    // trip 3: solution rows, the block's matrix entries, the finishing pass's x / b
    fc_d2u stage[CT];
#pragma unroll
    for (int u = 0; u < CT; ++u) stage[u] = *reinterpret_cast<const fc_d2u*>(x + (size_t)cid[u] * KB + 2 * sp);
    constexpr int ET = 4;  // 16 lanes per row, four 128-byte pieces: rows of up to 64 entries in one go
    double ev[ET];
    unsigned short el[ET];
    const int lenc = (rd.z & 0xFFFF) > 0 ? (rd.z & 0xFFFF) : 1;
#pragma unroll
    for (int u = 0; u < ET; ++u) {
      const int e = q16 + 16 * u;
      const int ec = e < lenc ? e : lenc - 1;
      ev[u] = a_val[g0 + ec];
      el[u] = a_lidx[g0 + ec];
    }
    double xv[NT], bv[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int s = (t + 256 * u) % KB;
      xv[u] = x[(size_t)fi[u] * KB + s];
      bv[u] = b[(size_t)fi[u] * KB + s];
    }
#pragma unroll
    for (int u = 0; u < CT; ++u)
      if (t / HP + u * (256 / HP) < bk.ncols) *reinterpret_cast<fc_d2u*>(xs + (t / HP + u * (256 / HP)) * KB + 2 * sp) = stage[u];
#pragma unroll
    for (int u = 0; u < ET; ++u)
      if (q16 + 16 * u < len) vs[k0 + q16 + 16 * u] = ev[u], ls[k0 + q16 + 16 * u] = el[u];
    for (int e = q16 + 16 * ET; e < len; e += 16) {  // (longer rows: the rest, piece by piece)
      vs[k0 + e] = a_val[g0 + e];
      ls[k0 + e] = a_lidx[g0 + e];
    }
    __syncthreads();
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 4
    for (int k = k0 + j; k < k1; k += JL) {
      const double v = vs[k];
      const fc_d2u xa = *reinterpret_cast<const fc_d2u*>(xs + (int)ls[k] * KB + 2 * sp);
      s0 += v * xa.x;
      s1 += v * xa.y;
    }
#pragma unroll
    for (int off = (JL / 2) * HP; off >= HP; off >>= 1) {
      s0 += __shfl_down(s0, off, 16);
      s1 += __shfl_down(s1, off, 16);
    }
    // (A x)[row][s] -> LDS, then thread = (row, simulation) finishes the row
    __syncthreads();  // xs is free
    if (j == 0) {
      xs[rl * KB + 2 * sp] = s0;
      xs[rl * KB + 2 * sp + 1] = s1;
    }
    __syncthreads();
    double r2 = 0.0, b2 = 0.0;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      if (!fok[u]) continue;
      const int e = t + 256 * u, r_l = e / KB, s = e % KB;
      const double res = bv[u] - xs[r_l * KB + s];
      r2 += res * res;
      b2 += bv[u] * bv[u];
      if (fvel[u] && !isfinite(xv[u])) atomicOr(flag + s, 1);  // (reference flowsolver.py:731,816-819: the velocity is tested)
    }
    // threads e = r_l * KB + s: simulation = t % KB for every trip (256 is a multiple of KB)
    red[0][t] = r2;
    red[1][t] = b2;
    __syncthreads();
    if (t < KB) {
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int g = 0; g < 256 / KB; ++g) {
        a0 += red[0][g * KB + t];
        a1 += red[1][g * KB + t];
      }
      partial[((size_t)t * 3 + 0) * G + rbk] = a0;
      partial[((size_t)t * 3 + 1) * G + rbk] = a1;
    }
  }
}

// fc_final for simulation s = blockIdx.x: folds its partials, evaluates the sensor rows on its column of the new solution x
// (permuted numbering; s_idxp = the sensor dofs' permuted positions), publishes its record (fc_publish: checksummed, the host
// polls it) and clears its non-finite flag for the next step (the flags are per step: one diverged run does not mark the later
// steps of the batch)
template <int KB>
__global__ __launch_bounds__(1024) void fc_final_b(int G, int n_row_blocks, const double* __restrict__ partial, int n_sens,
                                                   const int* __restrict__ s_rowptr, const int* __restrict__ s_idx,
                                                   const double* __restrict__ s_w, const double* __restrict__ up,
                                                   int* __restrict__ flag, double* __restrict__ rec, int rstride,
                                                   const double* __restrict__ seq_in, int compute_energy) {
  // one workgroup of 1024 threads per simulation: the kernel is a chain of dependent round trips over 3 G partial sums, so
  // what counts is how few trips there are (sixteen waves, eight loads per array and trip: 8 192 partials per trip)
  const int s = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  __shared__ double red[3][16];
  __shared__ double ysh[64];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  const double* __restrict__ ps = partial + (size_t)s * 3 * G;  // [w][block], contiguous per simulation
  for (int base = 0; base < G; base += 8 * nt) {
    double v0[8], v1[8], v2[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * nt + t;
      v0[u] = i < n_row_blocks ? ps[i] : 0.0;  // row blocks [0, n_row_blocks): residual sums; cell blocks behind them: energy
      v1[u] = i < n_row_blocks ? ps[(size_t)G + i] : 0.0;
      v2[u] = (i >= n_row_blocks && i < G) ? ps[2 * (size_t)G + i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 += v0[u];
      a1 += v1[u];
      a2 += v2[u];
    }
  }
  const int wave = t >> 6, lane = t & 63, nw = nt >> 6;
  for (int q = wave; q < n_sens; q += nw) {
    double acc = 0.0;
    for (int k = s_rowptr[q] + lane; k < s_rowptr[q + 1]; k += 64) acc += s_w[k] * up[(size_t)s_idx[k] * KB + s];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) ysh[q] = acc;
  }
  // fixed order: lanes of a wave (shuffle tree), then the waves in index order
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a0 += __shfl_down(a0, off, 64);
    a1 += __shfl_down(a1, off, 64);
    a2 += __shfl_down(a2, off, 64);
  }
  if (lane == 0) {
    red[0][wave] = a0;
    red[1][wave] = a1;
    red[2][wave] = a2;
  }
  __syncthreads();
  if (t == 0) {
    double r0 = 0.0, r1 = 0.0, r2 = 0.0;
    for (int w = 0; w < nw; ++w) {
      r0 += red[0][w];
      r1 += red[1][w];
      r2 += red[2][w];
    }
    double* r = rec + (size_t)s * rstride;
    const int fl = flag[s] & 1;
    flag[s] = 0;
    fc_publish(ysh, n_sens, compute_energy ? 0.5 * r2 : 0.0, r0, r1, (double)fl, r + 64, r + 128, r + 129, r + 136, r + 137, seq_in[0]);
  }
}

// Overlapped tail of the batched step (single GPU, cache-resident factors): fc_early_b is what the host waits for -- sensors of every
// simulation on the new solution, its non-finite flag (raised by the down-sweep tiles as they write; cleared here: per step), one
// checksummed record per simulation -- and releases the side stream's gate; fc_tail_b and fc_final_late_b (residual norms, energy ->
// a late record per simulation and step parity) then run on the side stream while the host and the next step go on.
template <int KB>
__global__ __launch_bounds__(256) void fc_early_b(int n_sens, const int* __restrict__ s_rowptr, const int* __restrict__ s_idxp,
                                                  const double* __restrict__ s_w, const double* __restrict__ x, int* __restrict__ flag,
                                                  double* __restrict__ rec, int rstride, const double* __restrict__ seq_in,
                                                  unsigned long long* __restrict__ solved /* null: a later launch opens the gate */) {
  const int s = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
  __shared__ double ysh[64];
  for (int q = wave; q < n_sens; q += 4) {
    double acc = 0.0;
    for (int k = s_rowptr[q] + lane; k < s_rowptr[q + 1]; k += 64) acc += s_w[k] * x[(size_t)s_idxp[k] * KB + s];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) ysh[q] = acc;
  }
  __syncthreads();
  if (t == 0) {
    const int fl = flag[s] & 1;
    flag[s] = 0;
    const double seq = seq_in[0];
    double* r = rec + (size_t)s * rstride;
    fc_publish(ysh, n_sens, 0.0, 0.0, 0.0, (double)fl, r + 64, r + 128, r + 129, r + 136, r + 137, seq);
    if (s == 0 && solved) __hip_atomic_store(solved, (unsigned long long)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// gate of the batched side stream: as fc_wait_solved, the sequence number read from the step's slot of the mapped record (graph replay:
// no per-step kernel argument)
__global__ void fc_wait_solved_b(const unsigned long long* __restrict__ solved, const double* __restrict__ seq_in, int* __restrict__ gave_up, long max_spin) {
  const unsigned long long seq = (unsigned long long)seq_in[0];
  for (long spin = 0; spin < max_spin; ++spin) {
    if (__hip_atomic_load(solved, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= seq) {
      *gave_up = 0;
      return;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  *gave_up = 1;
}
// late record of simulation s = blockIdx.x: rec[0] = E, [1] = sum r^2, [2] = sum b^2, [3] = seq, [4], [5] = checksums, [6] = the gate gave up
template <int KB>
__global__ __launch_bounds__(1024) void fc_final_late_b(int G, int n_row_blocks, const double* __restrict__ partial, double* __restrict__ rec, int rstride,
                                                        int rec_off, const double* __restrict__ seq_in, int compute_energy, const int* __restrict__ gave_up) {
  const int s = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  __shared__ double red[3][16];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  const double* __restrict__ ps = partial + (size_t)s * 3 * G;
  for (int base = 0; base < G; base += 8 * nt) {
    double v0[8], v1[8], v2[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * nt + t;
      v0[u] = i < n_row_blocks ? ps[i] : 0.0;
      v1[u] = i < n_row_blocks ? ps[(size_t)G + i] : 0.0;
      v2[u] = (i >= n_row_blocks && i < G) ? ps[2 * (size_t)G + i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 += v0[u];
      a1 += v1[u];
      a2 += v2[u];
    }
  }
  const int wave = t >> 6, lane = t & 63, nw = nt >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a0 += __shfl_down(a0, off, 64);
    a1 += __shfl_down(a1, off, 64);
    a2 += __shfl_down(a2, off, 64);
  }
  if (lane == 0) {
    red[0][wave] = a0;
    red[1][wave] = a1;
    red[2][wave] = a2;
  }
  __syncthreads();
  if (t == 0) {
    double r0 = 0.0, r1 = 0.0, r2 = 0.0;
    for (int w = 0; w < nw; ++w) {
      r0 += red[0][w];
      r1 += red[1][w];
      r2 += red[2][w];
    }
    typedef unsigned long long u64;
    const double seq = seq_in[0];
    const double v[4] = {compute_energy ? 0.5 * r2 : 0.0, r0, r1, (gave_up && *gave_up) ? 1.0 : 0.0};
    double* r = rec + (size_t)s * rstride + rec_off;
    u64 x = (u64)__double_as_longlong(seq), w = x, k = 3;
    for (int i = 0; i < 4; ++i, k += 2) {
      r[i < 3 ? i : 6] = v[i];
      x ^= (u64)__double_as_longlong(v[i]);
      w += k * (u64)__double_as_longlong(v[i]);
    }
    r[4] = __longlong_as_double((long long)x);
    r[5] = __longlong_as_double((long long)w);
    r[3] = seq;
  }
}

// host <-> device layout change of a state block: dev[i * KB + s] <-> host[s * n + p(i)], p = perm (device rows are in the
// solver's permuted numbering, host vectors in the W layout) or the identity (perm == nullptr)
template <int KB>
__global__ void fc_b_interleave(int n, int k, const double* __restrict__ src, double* __restrict__ dst, int to_device,
                                const int* __restrict__ perm) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t / KB, s = t % KB;
  if (i >= n) return;
  const int j = perm ? perm[i] : i;
  if (to_device)
    dst[(size_t)i * KB + s] = s < k ? src[(size_t)s * n + j] : 0.0;
  else if (s < k)
    dst[(size_t)s * n + j] = src[(size_t)i * KB + s];
}

// one simulation's column of a [rows][KB] block <- 0 (fc_reset_sim_batch: a diverged run is taken out of the batch)
template <int KB>
__global__ void fc_b_zero_column(int n, int s, double* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[(size_t)i * KB + s] = 0.0;
}
