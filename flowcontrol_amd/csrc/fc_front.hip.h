// Dense front elimination of the device-side numeric factorisation (fc_refactor): blocked Gauss-Jordan on fp64 matrix
// cores.  Included by fc_hip.hip only.
//
// A front  [F11 F12; F21 F22]  (nf x nf row-major, ni pivot columns) is swept IN PLACE into
//     [ F11^-1 ,  F11^-1 F12 ;  -F21 F11^-1 ,  F22 - F21 F11^-1 F12 ]
// (= pivot-block inverse D^-1, the U block, the -L block and the Schur complement the parent receives) by block
// steps of FC_FE_KB = 32 pivot columns K = [k0, k0 + kb):
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

#define FC_FE_KB 32

struct __attribute__((aligned(16))) FcFront {
  long long front;  // offset of the nf x nf row-major front
  long long voff;   // offset of the node's factor values
  int nf, ni;
  long long scratch;  // offset of this front's scratch: W (KB x KB) then Cs (nf x KB)
};

typedef double fc_d4 __attribute__((ext_vector_type(4)));

// W = A[K,K]^-1 by Gauss-Jordan with partial pivoting among the block's rows (ties -> smallest row: reproducible)
__global__ __launch_bounds__(256) void fc_fe_pivot(const FcFront* __restrict__ nodes, double* fronts, double* __restrict__ scratch, int step) {
  __shared__ double a[FC_FE_KB][FC_FE_KB + 1];
  __shared__ int piv[FC_FE_KB];
  __shared__ double wv[4];
  __shared__ int wi[4];
  const FcFront nd = nodes[blockIdx.x];
  const int k0 = step * FC_FE_KB;
  if (k0 >= nd.ni) return;
  const int kb = nd.ni - k0 < FC_FE_KB ? nd.ni - k0 : FC_FE_KB;
  const int nf = nd.nf;
  const double* A = fronts + nd.front;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  for (int e = t; e < FC_FE_KB * FC_FE_KB; e += 256) {
    const int r = e / FC_FE_KB, c = e % FC_FE_KB;
    a[r][c] = (r < kb && c < kb) ? A[(size_t)(k0 + r) * nf + k0 + c] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  for (int k = 0; k < kb; ++k) {
    // pivot search in column k, rows k .. kb-1 (one wave is enough)
    if (wave == 0) {
      double best = -1.0;
      int bi = k;
      if (lane >= k && lane < kb) {
        best = fabs(a[lane][k]);
        bi = lane;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(best, off, 64);
        const int oi = __shfl_down(bi, off, 64);
        if (ov > best || (ov == best && oi < bi)) {
          best = ov;
          bi = oi;
        }
      }
      if (lane == 0) piv[k] = bi;
    }
    __syncthreads();
    const int p = piv[k];
    if (p != k && t < FC_FE_KB) {
      const double u = a[k][t], v = a[p][t];
      a[k][t] = v;
      a[p][t] = u;
    }
    __syncthreads();
    const double d = 1.0 / a[k][k];
    // every thread updates its entries: row k scaled, other rows eliminated; column k takes the swept values
    double nv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = t + 256 * q, r = e / FC_FE_KB, c = e % FC_FE_KB;
      const double ark = a[r][k], akc = a[k][c];
      double v;
      if (r == k)
        v = c == k ? d : akc * d;
      else
        v = c == k ? -ark * d : a[r][c] - ark * (akc * d);
      nv[q] = v;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = t + 256 * q;
      a[e / FC_FE_KB][e % FC_FE_KB] = nv[q];
    }
    __syncthreads();
  }
  // the row swaps act on the columns of the inverse, in reverse order
  for (int k = kb - 1; k >= 0; --k) {
    const int p = piv[k];
    if (p != k && t < FC_FE_KB) {
      const double u = a[t][k], v = a[t][p];
      a[t][k] = v;
      a[t][p] = u;
    }
    __syncthreads();
  }
  double* W = scratch + nd.scratch;
  for (int e = t; e < FC_FE_KB * FC_FE_KB; e += 256) {
    const int r = e / FC_FE_KB, c = e % FC_FE_KB;
    W[e] = (r < kb && c < kb) ? a[r][c] : 0.0;
  }
  (void)wv;
  (void)wi;
}

// blockIdx.x < ct: row panel, 64 columns per workgroup:  A[K, j] = sum_c W[., c] A[k0 + c, j]  (j in K: = W)
// blockIdx.x >= ct: column panel copy, 64 rows per workgroup: Cs[i, c] = A[i, k0 + c]  (zero beyond kb)
__global__ __launch_bounds__(256) void fc_fe_panels(const FcFront* __restrict__ nodes, double* fronts, double* __restrict__ scratch, int step,
                                                    int ct) {
  __shared__ double Ws[FC_FE_KB][FC_FE_KB + 1];
  __shared__ double Rs[FC_FE_KB][64 + 1];
  const FcFront nd = nodes[blockIdx.y];
  const int k0 = step * FC_FE_KB;
  if (k0 >= nd.ni) return;
  const int kb = nd.ni - k0 < FC_FE_KB ? nd.ni - k0 : FC_FE_KB;
  const int nf = nd.nf;
  double* A = fronts + nd.front;
  const double* W = scratch + nd.scratch;
  double* Cs = scratch + nd.scratch + FC_FE_KB * FC_FE_KB;
  const int t = threadIdx.x;
  if ((int)blockIdx.x >= ct) {
    const int i0 = ((int)blockIdx.x - ct) * 64;
    if (i0 >= nf) return;
    for (int e = t; e < 64 * FC_FE_KB; e += 256) {
      const int i = i0 + e / FC_FE_KB, c = e % FC_FE_KB;
      if (i < nf) Cs[(size_t)i * FC_FE_KB + c] = c < kb ? A[(size_t)i * nf + k0 + c] : 0.0;
    }
    return;
  }
  const int j0 = blockIdx.x * 64;
  if (j0 >= nf) return;
  for (int e = t; e < FC_FE_KB * FC_FE_KB; e += 256) Ws[e / FC_FE_KB][e % FC_FE_KB] = W[e];
  for (int e = t; e < FC_FE_KB * 64; e += 256) {
    const int c = e / 64, j = j0 + e % 64;
    Rs[c][e % 64] = (c < kb && j < nf) ? A[(size_t)(k0 + c) * nf + j] : 0.0;
  }
  __syncthreads();
  for (int e = t; e < FC_FE_KB * 64; e += 256) {
    const int r = e / 64, jj = e % 64, j = j0 + jj;
    if (r >= kb || j >= nf) continue;
    double s;
    if (j >= k0 && j < k0 + kb) {
      s = Ws[r][j - k0];
    } else {
      s = 0.0;
#pragma unroll 8
      for (int c = 0; c < FC_FE_KB; ++c) s += Ws[r][c] * Rs[c][jj];
    }
    A[(size_t)(k0 + r) * nf + j] = s;
  }
}

// 64 x 64 tile of the trailing update on the fp64 matrix cores: wave w owns rows [16 w, 16 w + 16) x 64 columns
// (four 16 x 16 accumulators).  A operand: Cs (rows of the tile, K = 32); B operand: the new pivot rows A[K, :].
//   v_mfma_f64_16x16x4_f64: lane l holds A[row l & 15][k = l >> 4], B[k = l >> 4][col l & 15];
//   D[row (l >> 4) + 4 r][col l & 15] in register r.
__global__ __launch_bounds__(256) void fc_fe_update(const FcFront* __restrict__ nodes, double* fronts, const double* __restrict__ scratch,
                                                    int step, int tiles_per_side) {
  const FcFront nd = nodes[blockIdx.y];
  const int k0 = step * FC_FE_KB;
  if (k0 >= nd.ni) return;
  const int kb = nd.ni - k0 < FC_FE_KB ? nd.ni - k0 : FC_FE_KB;
  const int nf = nd.nf;
  const int ti = blockIdx.x / tiles_per_side, tj = blockIdx.x % tiles_per_side;
  const int i0 = ti * 64, j0 = tj * 64;
  if (i0 >= nf || j0 >= nf) return;
  double* A = fronts + nd.front;
  const double* Cs = scratch + nd.scratch + FC_FE_KB * FC_FE_KB;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int lr = lane & 15, lk = lane >> 4;
  const int arow = i0 + 16 * wave + lr;
  double av[FC_FE_KB / 4];
#pragma unroll
  for (int s = 0; s < FC_FE_KB / 4; ++s) av[s] = arow < nf ? Cs[(size_t)arow * FC_FE_KB + 4 * s + lk] : 0.0;  // zero beyond kb (fc_fe_panels)
  fc_d4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    acc[c] = fc_d4{0.0, 0.0, 0.0, 0.0};
    const int col = j0 + 16 * c + lr;
#pragma unroll
    for (int s = 0; s < FC_FE_KB / 4; ++s) {
      const int k = 4 * s + lk;
      const double bv = (k < kb && col < nf) ? A[(size_t)(k0 + k) * nf + col] : 0.0;
      acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv, acc[c], 0, 0, 0);
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int col = j0 + 16 * c + lr;
    if (col >= nf) continue;
    const bool inK = col >= k0 && col < k0 + kb;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + 16 * wave + lk + 4 * r;
      if (row >= nf || (row >= k0 && row < k0 + kb)) continue;  // the pivot rows are final (fc_fe_panels)
      double* p = A + (size_t)row * nf + col;
      *p = (inK ? 0.0 : *p) - acc[c][r];
    }
  }
}

// factor rows [D^-1 | -U] (stride nf) and the -L block (nb x ni, stride ni) into the layout the sweeps read
__global__ __launch_bounds__(256) void fc_fe_export(const FcFront* __restrict__ nodes, const double* __restrict__ fronts, double* __restrict__ fvals) {
  const FcFront nd = nodes[blockIdx.y];
  const int nf = nd.nf, ni = nd.ni;
  const int i0 = blockIdx.x * 16;
  if (i0 >= nf || ni == 0) return;
  const double* A = fronts + nd.front;
  double* dv = fvals + nd.voff;
  double* mw = dv + (size_t)ni * nf;
  for (int i = i0; i < i0 + 16 && i < nf; ++i) {
    if (i < ni) {
      for (int j = threadIdx.x; j < nf; j += 256) dv[(size_t)i * nf + j] = j < ni ? A[(size_t)i * nf + j] : -A[(size_t)i * nf + j];
    } else {
      for (int j = threadIdx.x; j < ni; j += 256) mw[(size_t)(i - ni) * ni + j] = A[(size_t)i * nf + j];
    }
  }
}
