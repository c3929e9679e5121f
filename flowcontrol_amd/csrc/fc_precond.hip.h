// fc_precond.hip.h -- device kernels of the factorisation-free preconditioner (fc_setup_krylov; host side: fc_precond.hpp).
//
// Every step of the SIMPLE / AMG apply is one instance of ONE fused CSR kernel
//
//     out[opos(i)] = base[i] + scale * dinv[i] * ( rhs[rpos(i)] - sum_j M[i][j] x[j] )
//
// with any of base / dinv / rhs / M absent (0 / 1 / 0 / 0) and optional gather (rpos) and scatter (opos) index lists:
//   velocity sweeps      u  = K_F in                           (two damped-Jacobi sweeps folded into ONE product on the host,
//                                                               K_F = Wd (2 I - F Wd), its columns address the Krylov vector)
//   further sweeps       u' = u + w D^-1 (r_u - F u)
//   Schur right-hand side   = B u - r_p                        (scale = -1)
//   AMG level, down      r_c = G r,  G = R (I - A Wd)          (pre-smoothing + residual + restriction folded)
//   AMG level, up        z = U [r ; z_c],  U = [Wd (2 I - A Wd) | (I - Wd A) P]   (prolongation + post-smoothing folded)
// plus fc_pc_final (velocity update z_u = u - D^-1 Bt z_p and the scatter of (z_u, z_p) into the Krylov vector) and fc_pc_dense.
// fp64 throughout; LANES lanes per row with the predicated 4-deep issue of fc_spmv_csr.  The matrices are small (the
// pressure Schur complement has nv rows, its AMG levels shrink by ~8 each) and every launch is latency-bound: what
// matters is the number of launches, not their bytes -- DESIGN.md section 4.1.
#pragma once
#include <hip/hip_runtime.h>

struct FcPcArgs {
  int n;
  const int* rp;      // CSR of M (null: no matrix term)
  const int* ci;
  const double* v;
  const double* x;    // operand of M
  const double* rhs;  // null: 0
  const int* rpos;    // gather positions of rhs (null: i)
  const double* dinv; // null: 1
  const double* base; // null: 0
  double scale;
  double* out;
  const int* opos;    // scatter positions of out (null: i)
};

template <int LANES>
__global__ __launch_bounds__(256) void fc_pc_csr(FcPcArgs a) {
  constexpr int RPB = 256 / LANES;
  const int lane = threadIdx.x % LANES;
  const int row = blockIdx.x * RPB + threadIdx.x / LANES;
  double s = 0.0;
  if (row < a.n && a.rp) {
    const int k0 = a.rp[row], k1 = a.rp[row + 1];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int b = k0; b < k1; b += 4 * LANES) {
      const int j0 = b + lane, j1 = j0 + LANES, j2 = j1 + LANES, j3 = j2 + LANES;
      const int c0 = j0 < k1 ? a.ci[j0] : 0, c1 = j1 < k1 ? a.ci[j1] : 0;
      const int c2 = j2 < k1 ? a.ci[j2] : 0, c3 = j3 < k1 ? a.ci[j3] : 0;
      const double v0 = j0 < k1 ? a.v[j0] : 0.0, v1 = j1 < k1 ? a.v[j1] : 0.0;
      const double v2 = j2 < k1 ? a.v[j2] : 0.0, v3 = j3 < k1 ? a.v[j3] : 0.0;
      s0 += v0 * a.x[c0];
      s1 += v1 * a.x[c1];
      s2 += v2 * a.x[c2];
      s3 += v3 * a.x[c3];
    }
    s = (s0 + s1) + (s2 + s3);
  }
#pragma unroll
  for (int off = LANES / 2; off > 0; off >>= 1) s += __shfl_down(s, off, LANES);
  if (row < a.n && lane == 0) {
    double t = (a.rhs ? a.rhs[a.rpos ? a.rpos[row] : row] : 0.0) - s;
    if (a.dinv) t *= a.dinv[row];
    const double o = (a.base ? a.base[row] : 0.0) + a.scale * t;
    a.out[a.opos ? a.opos[row] : row] = o;
  }
}

// coarsest AMG level: x = Ainv r with the dense inverse (n <= a few hundred), one wave per row
__global__ __launch_bounds__(256) void fc_pc_dense(int n, const double* __restrict__ Ainv, const double* __restrict__ r, double* __restrict__ x) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const double* a = Ainv + (size_t)row * n;
  double s = 0.0;
  for (int j = lane; j < n; j += 64) s += a[j] * r[j];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) x[row] = s;
}

// last launch of an apply: out[vpos[i]] = u[i] - dinv[i] (Bt zp)[i] for the velocity rows, out[ppos[j]] = zp[j] for the pressure rows
// (4 lanes per row: a row of Bt has 2-4 entries)
__global__ __launch_bounds__(256) void fc_pc_final(int nu, int np, const int* __restrict__ rp, const int* __restrict__ ci, const double* __restrict__ v,
                                                   const double* __restrict__ zp, const double* __restrict__ dinv, const double* __restrict__ u,
                                                   const int* __restrict__ vpos, const int* __restrict__ ppos, double* __restrict__ out) {
  const int lane = threadIdx.x & 3;
  const int row = blockIdx.x * 64 + (threadIdx.x >> 2);
  if (row >= nu + np) return;
  if (row >= nu) {
    if (lane == 0) out[ppos[row - nu]] = zp[row - nu];
    return;
  }
  double s = 0.0;
  for (int k = rp[row] + lane; k < rp[row + 1]; k += 4) s += v[k] * zp[ci[k]];
  s += __shfl_down(s, 2, 4);
  s += __shfl_down(s, 1, 4);
  if (lane == 0) out[vpos[row]] = u[row] - dinv[row] * s;
}
