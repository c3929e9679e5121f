// Device kernels of libfc_hip.so (gfx950 / CDNA4, wave64).  Included by fc_hip.hip only.
//
// All arithmetic is fp64 (the reference computes in fp64 end to end).  Every kernel here is
// bandwidth/latency bound: there is no dense contraction, so no MFMA; what matters
// is coalesced SoA access (thread-per-cell element loops read/write [slot][cell] arrays),
// deterministic gathers instead of atomics (bit-reproducible sums), and enough loads in flight
// per lane in the CSR kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FC_NQ 7

__constant__ double c_phi2[FC_NQ * 6];       // P2 basis at the 7 Radon points
__constant__ double c_dphi2[FC_NQ * 6 * 2];  // reference gradients d/d(xi,eta)
__constant__ double c_phi1[FC_NQ * 3];       // P1 basis
__constant__ double c_qw[FC_NQ];             // weights (sum = 1; times |detJ|/2)

// ---------------------------------------------------------------------------------------------
// RHS element loop: the `rhs` of NSForms._order1/_order2 (reference nsforms.py:238-305)
//   g = cm_n u_n + cm_nn u_nn + cc_n (u_n.grad)u_n + cc_nn (u_nn.grad)u_nn + f ;  L_e[a,j] = ∫ g_j φ_a
// EIGHT lanes per cell: lane q < 7 evaluates g at Radon point q (the nodal values of the cell come
// through LDS, fetched once by lanes a < 6), the weighted point
// values are exchanged inside the group and lane a < 6 sums the contribution to test function a in
// the fixed order q = 0..6.  With a thread per cell the launch had 12 k threads (< 1 wave per CU) and
// ran ~1 000 dependent FMAs each; this way it is two memory round trips and ~150 FMAs deep.
// Element vectors go to ev[slot][cell] (slot = a + 6 j), summed per dof by fc_rhs_gather
// (wavefront-independent, deterministic).
// ---------------------------------------------------------------------------------------------
// un / unn: the velocity-pressure state vectors in the solver's PERMUTED numbering (they are solution halves of the
// sweep work buffer, never copied: fc_hip.hip "state ring"); cnp[a][c] / cnp[6 + a][c] = permuted position of the x- / y-
// velocity dof of node a of cell c.  cn (node ids) is only needed to address a body-force profile (fprof, W layout).
__global__ __launch_bounds__(256) void fc_rhs_elem(int nc, int nn, const int* __restrict__ cn, const int* __restrict__ cnp,
                                                   const double* __restrict__ geom,
                                                   const double* __restrict__ un,
                                                   const double* __restrict__ unn,
                                                   const double* __restrict__ fprof, int n_act,
                                                   const double* __restrict__ uctrl, double cm_n,
                                                   double cm_nn, double cc_n, double cc_nn,
                                                   double* __restrict__ ev,
                                                   const int* __restrict__ cell_list, int ncl) {
  // cell_list != nullptr: this rank's share of the cells (multi-GPU partition), else all nc cells
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int cl = t >> 3, lane = t & 7;
  const bool active = cl < ncl;
  const int c = active ? (cell_list ? cell_list[cl] : cl) : 0;
  // The loop is bound by the number of vector-memory instructions, not by bytes (the nodal values sit in the caches): lane
  // a < 6 fetches node a of its cell ONCE -- its two velocity fields and the force profile -- and the eight lanes of the cell
  // read the six nodes back from LDS (broadcast reads; row stride 9 keeps the eight cells of a wave on different banks).
  __shared__ double sh[6][32 * 9];
  const int cb = (threadIdx.x >> 3) * 9;
  if (lane < 6) {
    const int ix = cnp[lane * nc + c], iy = cnp[(6 + lane) * nc + c];
    double fx = 0.0, fy = 0.0;
    if (n_act > 0) {
      const int n = cn[lane * nc + c];
      for (int k = 0; k < n_act; ++k) {
        const double uk = uctrl[k];
        fx += uk * fprof[(size_t)k * 2 * nn + n];
        fy += uk * fprof[(size_t)k * 2 * nn + nn + n];
      }
    }
    sh[0][cb + lane] = un[ix];
    sh[1][cb + lane] = un[iy];
    sh[2][cb + lane] = unn[ix];
    sh[3][cb + lane] = unn[iy];
    sh[4][cb + lane] = fx;
    sh[5][cb + lane] = fy;
  }
  const int q = lane < FC_NQ ? lane : FC_NQ - 1;  // lane 7 shadows point 6 with zero weight
  const double j00 = geom[c], j01 = geom[nc + c], j10 = geom[2 * nc + c], j11 = geom[3 * nc + c];
  const double wq = lane < FC_NQ ? c_qw[q] * 0.5 * geom[4 * nc + c] : 0.0;
  double ux = 0, uy = 0, uxi = 0, uet = 0, vxi = 0, vet = 0;  // field n: value, d/dxi, d/deta of (ux, uy)
  double wx = 0, wy = 0, wxi = 0, wet = 0, zxi = 0, zet = 0;  // field nn
  double gx = 0, gy = 0;
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const double ax = sh[0][cb + a], ay = sh[1][cb + a], bx = sh[2][cb + a], by = sh[3][cb + a];
    const double fx = sh[4][cb + a], fy = sh[5][cb + a];
    const double ph = c_phi2[q * 6 + a], dx = c_dphi2[(q * 6 + a) * 2], de = c_dphi2[(q * 6 + a) * 2 + 1];
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
    gx += ph * fx;
    gy += ph * fy;
  }
  // physical gradients: d/dx = d/dxi*j00 + d/deta*j10 ; d/dy = d/dxi*j01 + d/deta*j11
  const double ux_x = uxi * j00 + uet * j10, ux_y = uxi * j01 + uet * j11;
  const double uy_x = vxi * j00 + vet * j10, uy_y = vxi * j01 + vet * j11;
  const double wx_x = wxi * j00 + wet * j10, wx_y = wxi * j01 + wet * j11;
  const double wy_x = zxi * j00 + zet * j10, wy_y = zxi * j01 + zet * j11;
  gx += cm_n * ux + cm_nn * wx + cc_n * (ux * ux_x + uy * ux_y) + cc_nn * (wx * wx_x + wy * wx_y);
  gy += cm_n * uy + cm_nn * wy + cc_n * (ux * uy_x + uy * uy_y) + cc_nn * (wx * wy_x + wy * wy_y);
  gx *= wq;
  gy *= wq;
  // lane a < 6 gathers the weighted point values and tests them with phi_a
  const int a = lane < 6 ? lane : 5;
  double accx = 0.0, accy = 0.0;
#pragma unroll
  for (int p = 0; p < FC_NQ; ++p) {
    const double pa = c_phi2[p * 6 + a];
    accx += pa * __shfl(gx, p, 8);
    accy += pa * __shfl(gy, p, 8);
  }
  if (active && lane < 6) {
    ev[(size_t)lane * nc + c] = accx;
    ev[(size_t)(6 + lane) * nc + c] = accy;
  }
}

// The same loop with a cell's whole work on ONE thread (nodal values in registers, basis tables through the scalar unit, no LDS, no
// shuffles; the sums run in the same order).  ~870 dependent-ish FMAs per thread: a loss on O1 (12 k cells = less than a wave per SIMD,
// see above), a gain once the mesh brings >= ~1 wave per SIMD -- the eight-lane form then pays 8 x the threads for the LDS round trip
// (pinball, 66.7 k cells: 14 us).  No body-force profiles (the time steps use pre-assembled load vectors).
__global__ __launch_bounds__(256) void fc_rhs_elem_reg(int nc, const int* __restrict__ cnp, const double* __restrict__ geom,
                                                       const double* __restrict__ un, const double* __restrict__ unn, double cm_n,
                                                       double cm_nn, double cc_n, double cc_nn, double* __restrict__ ev,
                                                       const int* __restrict__ cell_list, int ncl) {
  const int cl = blockIdx.x * blockDim.x + threadIdx.x;
  if (cl >= ncl) return;
  const int c = cell_list ? cell_list[cl] : cl;
  double ax[6], ay[6], bx[6], by[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const int ix = cnp[a * nc + c], iy = cnp[(6 + a) * nc + c];
    ax[a] = un[ix];
    ay[a] = un[iy];
    bx[a] = unn[ix];
    by[a] = unn[iy];
  }
  const double j00 = geom[c], j01 = geom[nc + c], j10 = geom[2 * nc + c], j11 = geom[3 * nc + c], hdet = 0.5 * geom[4 * nc + c];
  double accx[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, accy[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < FC_NQ; ++q) {
    double ux = 0, uy = 0, uxi = 0, uet = 0, vxi = 0, vet = 0;
    double wx = 0, wy = 0, wxi = 0, wet = 0, zxi = 0, zet = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const double ph = c_phi2[q * 6 + a], dx = c_dphi2[(q * 6 + a) * 2], de = c_dphi2[(q * 6 + a) * 2 + 1];
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
    }
    const double ux_x = uxi * j00 + uet * j10, ux_y = uxi * j01 + uet * j11;
    const double uy_x = vxi * j00 + vet * j10, uy_y = vxi * j01 + vet * j11;
    const double wx_x = wxi * j00 + wet * j10, wx_y = wxi * j01 + wet * j11;
    const double wy_x = zxi * j00 + zet * j10, wy_y = zxi * j01 + zet * j11;
    double gx = 0.0, gy = 0.0;
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
    ev[(size_t)a * nc + c] = accx[a];
    ev[(size_t)(6 + a) * nc + c] = accy[a];
  }
}

// per (permuted) row: sum the element contributions, lift and impose the Dirichlet data.
__global__ __launch_bounds__(256) void fc_rhs_gather(int N, const int* __restrict__ gptr,
                                                     const int* __restrict__ gidx,
                                                     const double* __restrict__ ev,
                                                     const int* __restrict__ bcslot,
                                                     const double* __restrict__ bcprof,
                                                     const double* __restrict__ lift, int n_act,
                                                     const double* __restrict__ uctrl,
                                                     double* __restrict__ b, double* __restrict__ y,
                                                     const unsigned char* __restrict__ rowkind, int lead,
                                                     const int* __restrict__ c_rowptr,
                                                     const int* __restrict__ c_col,
                                                     const double* __restrict__ c_val,
                                                     const double* __restrict__ un,
                                                     const unsigned char* __restrict__ colkind = nullptr,
                                                     const double* __restrict__ fvec = nullptr,
                                                     const double* __restrict__ uforce = nullptr) {
  // rowkind (multi-GPU): 0 = another rank's row, 1 = owned, 2 = root separator shared by all ranks
  // (every rank adds its cells' share; the BC value / lifting is added once, by the lead rank; of an explicit operator's
  // root rows every rank takes the columns it accounts for -- colkind: the dof kinds in the numbering of `un`, the permuted one)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double s = 0.0;
  const int kind = rowkind ? rowkind[i] : 1;
  const bool once = kind == 1 || (kind == 2 && lead);
  const int bs = bcslot[i];
  if (kind == 0) {
    s = 0.0;
  } else if (bs >= 0) {
    if (once)
      for (int k = 0; k < n_act; ++k) s += uctrl[k] * bcprof[(size_t)bs * n_act + k];
  } else {
    // eight element contributions per trip: the index loads, then the value loads, are all in flight
    // together (a vertex dof has 6-8 cells, an edge dof 2); summed in list order (reproducible)
    const int k0 = gptr[i], k1 = gptr[i + 1];
    for (int base = k0; base < k1; base += 8) {
      int id[8];
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) id[u] = base + u < k1 ? gidx[base + u] : -1;
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = id[u] >= 0 ? ev[id[u]] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (id[u] >= 0) s += v[u];
    }
    if (once)
      for (int k = 0; k < n_act; ++k) s -= uctrl[k] * lift[(size_t)k * N + i];
    // body-force actuators: + sum_k u_k F_k[row], F_k the assembled load vector of actuator k's profile (fc_force_rows: once per
    // profile -- the element loop then depends on the state only and can run ahead of u_ctrl); every rank adds its cells' share
    if (fvec)
      for (int k = 0; k < n_act; ++k) s += uforce[k] * fvec[(size_t)k * N + i];
    // explicit half of the linear terms of Crank-Nicolson (nsforms.py:212-216): -(C u_n)[row]
    if (c_rowptr)
      for (int k = c_rowptr[i]; k < c_rowptr[i + 1]; ++k) {
        const int j = c_col[k];
        if (kind == 2 && colkind) {
          const int ck = colkind[j];
          if (!(ck == 1 || (ck == 2 && lead))) continue;
        }
        s -= c_val[k] * un[j];
      }
  }
  b[i] = s;
  y[i] = s;  // y-half of the solver work buffer: the first factor sweep starts from b
}

// load vector of ONE body-force profile from its element vectors (fc_rhs_elem with the state terms switched off and a unit
// amplitude on that actuator): out[row] = sum of the row's element contributions, 0 on Dirichlet rows and on other ranks' rows
__global__ __launch_bounds__(256) void fc_force_rows(int N, const int* __restrict__ gptr, const int* __restrict__ gidx,
                                                     const double* __restrict__ ev, const int* __restrict__ bcslot,
                                                     const unsigned char* __restrict__ rowkind, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double s = 0.0;
  if (bcslot[i] < 0 && !(rowkind && rowkind[i] == 0))
    for (int k = gptr[i]; k < gptr[i + 1]; ++k) s += ev[gidx[k]];
  out[i] = s;
}

// ---------------------------------------------------------------------------------------------
// Bilinear-form element loop (lhs of the transient forms, Picard operator, steady Jacobian).
// One thread per (cell, test node a): rows a (ux) and 6+a (uy) of the 15x15 element matrix and
// the matching divergence columns.  em[(i*15+j)][cell].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fc_mat_elem(int nc, int nn, const int* __restrict__ cn,
                                                   const double* __restrict__ geom, double mass,
                                                   double nu, const double* __restrict__ adv,
                                                   double adv_scale, const double* __restrict__ lin,
                                                   double lin_scale, double pressure, double divergence,
                                                   double* __restrict__ em, const int* __restrict__ cell_list = nullptr,
                                                   int ncl = 0) {
  // (multi-GPU: cell_list = the cells whose element matrices this rank needs -- its own and those touching a root dof;
  //  the element matrices of the others keep their zeros)
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int a = blockIdx.y;  // wave-uniform test node
  if (tid >= (cell_list ? ncl : nc)) return;
  const int c = cell_list ? cell_list[tid] : tid;
  const double j00 = geom[c], j01 = geom[nc + c], j10 = geom[2 * nc + c], j11 = geom[3 * nc + c];
  const double hdet = 0.5 * geom[4 * nc + c];
  double Ux[6], Uy[6], Lx[6], Ly[6];
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    const int n = cn[b * nc + c];
    Ux[b] = adv ? adv[n] : 0.0;
    Uy[b] = adv ? adv[nn + n] : 0.0;
    Lx[b] = lin ? lin[n] : 0.0;
    Ly[b] = lin ? lin[nn + n] : 0.0;
  }
  double blk[6], l00[6], l01[6], l10[6], l11[6], P0[3], P1[3];
#pragma unroll
  for (int b = 0; b < 6; ++b) blk[b] = l00[b] = l01[b] = l10[b] = l11[b] = 0.0;
  P0[0] = P0[1] = P0[2] = P1[0] = P1[1] = P1[2] = 0.0;
#pragma unroll
  for (int q = 0; q < FC_NQ; ++q) {
    const double w = c_qw[q] * hdet;
    double gbx[6], gby[6];
    double uq = 0, vq = 0, lx_xi = 0, lx_et = 0, ly_xi = 0, ly_et = 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const double ph = c_phi2[q * 6 + b], dx = c_dphi2[(q * 6 + b) * 2], de = c_dphi2[(q * 6 + b) * 2 + 1];
      gbx[b] = dx * j00 + de * j10;
      gby[b] = dx * j01 + de * j11;
      uq += ph * Ux[b];
      vq += ph * Uy[b];
      lx_xi += dx * Lx[b];
      lx_et += de * Lx[b];
      ly_xi += dx * Ly[b];
      ly_et += de * Ly[b];
    }
    // GU[k][j] = d_k U_j of the `lin` field
    const double g00 = lx_xi * j00 + lx_et * j10;  // d_x Ux
    const double g10 = lx_xi * j01 + lx_et * j11;  // d_y Ux
    const double g01 = ly_xi * j00 + ly_et * j10;  // d_x Uy
    const double g11 = ly_xi * j01 + ly_et * j11;  // d_y Uy
    const double pa = c_phi2[q * 6 + a];
    const double dxa = c_dphi2[(q * 6 + a) * 2], dea = c_dphi2[(q * 6 + a) * 2 + 1];
    const double gax = dxa * j00 + dea * j10, gay = dxa * j01 + dea * j11;
    const double wpa = w * pa;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const double pb = c_phi2[q * 6 + b];
      const double mab = wpa * pb;
      blk[b] += mass * mab + nu * w * (gax * gbx[b] + gay * gby[b]) + adv_scale * wpa * (uq * gbx[b] + vq * gby[b]);
      // (row comp j, col comp k): ∫ φa φb d_k U_j
      l00[b] += lin_scale * mab * g00;  // j=0,k=0: d_x Ux
      l01[b] += lin_scale * mab * g10;  // j=0,k=1: d_y Ux
      l10[b] += lin_scale * mab * g01;  // j=1,k=0: d_x Uy
      l11[b] += lin_scale * mab * g11;  // j=1,k=1: d_y Uy
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const double ps = w * c_phi1[q * 3 + m];
      P0[m] += ps * gax;
      P1[m] += ps * gay;
    }
  }
  const size_t snc = (size_t)nc;
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    em[(size_t)(a * 15 + b) * snc + c] = blk[b] + l00[b];
    em[(size_t)(a * 15 + 6 + b) * snc + c] = l01[b];
    em[(size_t)((6 + a) * 15 + b) * snc + c] = l10[b];
    em[(size_t)((6 + a) * 15 + 6 + b) * snc + c] = blk[b] + l11[b];
  }
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    em[(size_t)(a * 15 + 12 + m) * snc + c] = pressure * P0[m];
    em[(size_t)((6 + a) * 15 + 12 + m) * snc + c] = pressure * P1[m];
    em[(size_t)((12 + m) * 15 + a) * snc + c] = divergence * P0[m];
    em[(size_t)((12 + m) * 15 + 6 + a) * snc + c] = divergence * P1[m];
  }
}

// per CSR slot: sum element-matrix contributions (inverted index, deterministic order)
__global__ __launch_bounds__(256) void fc_mat_gather(int64_t nnz, const int* __restrict__ mptr,
                                                     const int* __restrict__ midx,
                                                     const double* __restrict__ em,
                                                     double* __restrict__ vals) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nnz) return;
  double v = 0.0;
  for (int k = mptr[s]; k < mptr[s + 1]; ++k) v += em[midx[k]];
  vals[s] = v;
}

// SystemAssembler-style symmetric Dirichlet elimination on CSR values
__global__ __launch_bounds__(256) void fc_apply_bc_rows(int N, const int* __restrict__ rowptr,
                                                        const int* __restrict__ col,
                                                        const unsigned char* __restrict__ isbc,
                                                        double* __restrict__ vals) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  const bool rb = isbc[r] != 0;
  for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
    const int cidx = col[k];
    if (rb)
      vals[k] = (cidx == r) ? 1.0 : 0.0;
    else if (isbc[cidx])
      vals[k] = 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// CSR SpMV, LANES lanes per row (wave64 sub-groups), fp64 values / int32 columns.
//   MODE 0: y = A x         MODE 1: y = b - A x, xsave = x (optional), partial |y|^2 per block
// Algorithmic bytes: nnz*12 + N*16 + (N+1)*4  (SURVEY §8d).
// ---------------------------------------------------------------------------------------------
template <int LANES, int MODE>
__global__ __launch_bounds__(256) void fc_spmv_csr(int nrows, const int* __restrict__ rowptr,
                                                   const int* __restrict__ col,
                                                   const double* __restrict__ val,
                                                   const double* __restrict__ x,
                                                   const double* __restrict__ b,
                                                   double* __restrict__ y, double* __restrict__ xsave,
                                                   double* __restrict__ partial,
                                                   const unsigned char* __restrict__ rowmask) {
  // rowmask (multi-GPU residual monitor): only rows with mask == 1 (owned) are evaluated
  constexpr int RPB = 256 / LANES;
  const int lane = threadIdx.x % LANES;
  const int row = blockIdx.x * RPB + threadIdx.x / LANES;
  double s = 0.0;
  const bool active = row < nrows && (!rowmask || rowmask[row] == 1);
  if (active) {
    const int k0 = rowptr[row], k1 = rowptr[row + 1];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    // predicated 4-deep issue: the 8 streaming loads (values + columns) of a trip are in flight
    // before the first gather of x; a 29-nnz Taylor-Hood row is one trip of 8 lanes
    for (int base = k0; base < k1; base += 4 * LANES) {
      const int j0 = base + lane, j1 = j0 + LANES, j2 = j1 + LANES, j3 = j2 + LANES;
      const int c0 = j0 < k1 ? col[j0] : 0, c1 = j1 < k1 ? col[j1] : 0;
      const int c2 = j2 < k1 ? col[j2] : 0, c3 = j3 < k1 ? col[j3] : 0;
      const double v0 = j0 < k1 ? val[j0] : 0.0, v1 = j1 < k1 ? val[j1] : 0.0;
      const double v2 = j2 < k1 ? val[j2] : 0.0, v3 = j3 < k1 ? val[j3] : 0.0;
      s0 += v0 * x[c0];
      s1 += v1 * x[c1];
      s2 += v2 * x[c2];
      s3 += v3 * x[c3];
    }
    s = (s0 + s1) + (s2 + s3);
  }
#pragma unroll
  for (int off = LANES / 2; off > 0; off >>= 1) s += __shfl_down(s, off, LANES);
  double r2 = 0.0, b2 = 0.0;
  if (row < nrows && lane == 0 && !active) {
    y[row] = 0.0;
  } else if (row < nrows && lane == 0) {
    if (MODE == 0) {
      y[row] = s;
    } else {
      const double bb = b[row];
      const double r = bb - s;
      y[row] = r;
      if (xsave) xsave[row] = x[row];
      r2 = r * r;
      b2 = bb * bb;
    }
  }
  if (MODE == 1 && partial) {  // partial[0..grid) = sum r^2, partial[grid..2 grid) = sum b^2
    __shared__ double red[256], redb[256];
    red[threadIdx.x] = r2;
    redb[threadIdx.x] = b2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) {
        red[threadIdx.x] += red[threadIdx.x + st];
        redb[threadIdx.x] += redb[threadIdx.x + st];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      partial[blockIdx.x] = red[0];
      partial[gridDim.x + blockIdx.x] = redb[0];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Nested-dissection factor sweep: one level-wide block mat-vec in "segment list" form.
// Each factor row is a list of segments (val_off, col, len): len consecutive fp64 values times
//   col >= 0 : buf[col .. col+len)             (contiguous slice: a pivot-block / L-block row)
//   col <  0 : buf[idx[-(col+1) + j]]          (node-shared index list: a U-block row)
//   kind 0 (up):   buf[dest0 + r] += sum        (segments hold -L rows; reads deeper levels of y)
//   kind 1 (down): buf[dest0 + r]  = sum        ([D^-1 | -U] rows; reads y and shallower x)
// Rows of one level never read what the same launch writes, so a level is one launch.
// Values stream at 8 B/nnz; column information is O(1/len) bytes per value.
// ---------------------------------------------------------------------------------------------
// Storage type of the factor values.  double: the exact selected inverse (applied directly).  float / FcBf16 (bfloat16: the
// upper 16 bits of a float, full fp32 range): COMPRESSED factors at 50 % / 25 % of the memory — rounded once after the fp64
// elimination, accumulated in fp64 — which are a preconditioner for the device Krylov solvers (GMRES: 2 / ~9 iterations to
// 1e-12 on the cylinder operator).
struct FcBf16 {
  unsigned short u;
};
// value stream of the factor sweeps: every value is read exactly once per apply.  NT = nontemporal loads, for factors that do
// not fit the 256 MiB Infinity Cache anyway: streamed past the caches they leave the state vectors, the matrix of the residual
// monitor and the operand rows resident (refined O1 / pinball: +6 % steps/s; O1, whose 176 MB of factors DO stay in the Infinity
// Cache from step to step: -12 %, so the handle decides -- fc_solver_setup, FC_NT_BYTES)
template <bool NT>
__device__ __forceinline__ double fc_ld(const double* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  return *p;
}
template <bool NT>
__device__ __forceinline__ float fc_ld(const float* p) {
  return *p;
}
template <bool NT>
__device__ __forceinline__ FcBf16 fc_ld(const FcBf16* p) {
  return *p;
}
__device__ __forceinline__ double fc_val(double v) { return v; }
__device__ __forceinline__ double fc_val(float v) { return (double)v; }
__device__ __forceinline__ double fc_val(FcBf16 v) { return (double)__uint_as_float((unsigned)v.u << 16); }
template <typename VT>
__device__ __forceinline__ VT fc_pack(double v);
template <>
__device__ __forceinline__ double fc_pack<double>(double v) {
  return v;
}
template <>
__device__ __forceinline__ float fc_pack<float>(double v) {
  return (float)v;
}
template <>
__device__ __forceinline__ FcBf16 fc_pack<FcBf16>(double v) {
  const unsigned b = __float_as_uint((float)v);
  return FcBf16{(unsigned short)((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16)};  // round to nearest even
}

struct __attribute__((aligned(16))) FcSeg {
  long long val;  // offset of the first value
  int col;        // >= 0: first buffer index; < 0: -(offset into idx) - 1
  int len;
};

// LANES lanes cooperate on one row (8..64: sub-wave groups, RPB rows per 256-thread block; 256: the
// whole workgroup on one row — few, very long rows near the root).  Inside a row group, sub-groups
// of SUB lanes each take every (LANES/SUB)-th segment, so several segments of the row are in flight
// at once: a row is a *chain* of short dense slices and one slice per memory round trip would leave
// the launch latency-bound (bytes in flight = rows x slice length) far below the HBM rate.
template <int LANES, int SUB, typename VT = double, bool NT = false>
__global__ __launch_bounds__(256) void fc_nd_sweep(int nrows, const int64_t* __restrict__ seg_ptr,
                                                   const FcSeg* __restrict__ seg,
                                                   const int* __restrict__ idx,
                                                   const VT* __restrict__ val,
                                                   double* __restrict__ buf, int dest0, int accumulate,
                                                   const int* __restrict__ wg_order = nullptr,
                                                   const unsigned char* __restrict__ velrow = nullptr, int* __restrict__ flag = nullptr) {
  // wg_order: launch position -> row group, by decreasing work (the long rows start first instead of forming the launch's tail)
  // velrow (down stages of a time step, overlapped tail): the stage's slice of the velocity-row mask; a non-finite solution entry on
  // such a row raises *flag -- every solution row is written by exactly one down-stage launch, so the reference's finiteness test
  // (flowsolver.py:731,816-819) costs no pass of its own
  const int wg = wg_order ? wg_order[blockIdx.x] : (int)blockIdx.x;
  constexpr int RPB = 256 / LANES;
  constexpr int SW = LANES < 64 ? LANES : 64;  // shuffle width (descriptor broadcast, reduction)
  constexpr int G = LANES / SUB;               // segments processed concurrently per row
  const int lane = threadIdx.x % LANES;
  const int sl = threadIdx.x % SW;             // lane inside the shuffle group
  const int g = lane / SUB;                    // sub-group of this lane
  const int l2 = lane % SUB;
  const int row = wg * RPB + threadIdx.x / LANES;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  // rows beyond nrows still take part in the shuffles below (q0 == q1: zero trips)
  const int64_t q0 = row < nrows ? seg_ptr[row] : 0, q1 = row < nrows ? seg_ptr[row + 1] : 0;
  for (int64_t qb = q0; qb < q1; qb += SW) {
    // the lanes of a (sub-)wave fetch up to SW segment descriptors with one coalesced 16-B load
    // each, then broadcast them: no dependent descriptor -> value load chain per segment
    FcSeg mine = {0, 0, 0};
    if (qb + sl < q1) mine = seg[qb + sl];
    const int cnt = (int)((q1 - qb) < SW ? (q1 - qb) : SW);
    // sub-group g of wave w (LANES = 256: four waves) takes segments g, g + G, ...
    for (int sbase = 0; sbase < cnt; sbase += G) {
      const int sidx = sbase + g;
      const int src = sidx < cnt ? (sidx % SW) : 0;
      const long long vo = __shfl(mine.val, src, SW);
      const int c = __shfl(mine.col, src, SW);
      int len = __shfl(mine.len, src, SW);
      if (sidx >= cnt) len = 0;  // this sub-group has no segment in the last round
      const VT* __restrict__ v = val + vo;
      const VT vz = fc_pack<VT>(0.0);
      if (c >= 0) {
        const double* __restrict__ x = buf + c;
        // predicated 4-deep issue: all eight loads of a trip are in flight before the first FMA
        for (int base = 0; base < len; base += 4 * SUB) {
          const int j0 = base + l2, j1 = j0 + SUB, j2 = j1 + SUB, j3 = j2 + SUB;
          const double v0 = fc_val(j0 < len ? fc_ld<NT>(v + j0) : vz), v1 = fc_val(j1 < len ? fc_ld<NT>(v + j1) : vz);
          const double v2 = fc_val(j2 < len ? fc_ld<NT>(v + j2) : vz), v3 = fc_val(j3 < len ? fc_ld<NT>(v + j3) : vz);
          const double x0 = j0 < len ? x[j0] : 0.0, x1 = j1 < len ? x[j1] : 0.0;
          const double x2 = j2 < len ? x[j2] : 0.0, x3 = j3 < len ? x[j3] : 0.0;
          s0 += v0 * x0;
          s1 += v1 * x1;
          s2 += v2 * x2;
          s3 += v3 * x3;
        }
      } else {
        const int* __restrict__ ix = idx + (-(c + 1));
        for (int base = 0; base < len; base += 4 * SUB) {
          const int j0 = base + l2, j1 = j0 + SUB, j2 = j1 + SUB, j3 = j2 + SUB;
          const int i0 = j0 < len ? ix[j0] : 0, i1 = j1 < len ? ix[j1] : 0;
          const int i2 = j2 < len ? ix[j2] : 0, i3 = j3 < len ? ix[j3] : 0;
          const double v0 = fc_val(j0 < len ? fc_ld<NT>(v + j0) : vz), v1 = fc_val(j1 < len ? fc_ld<NT>(v + j1) : vz);
          const double v2 = fc_val(j2 < len ? fc_ld<NT>(v + j2) : vz), v3 = fc_val(j3 < len ? fc_ld<NT>(v + j3) : vz);
          s0 += v0 * buf[i0];
          s1 += v1 * buf[i1];
          s2 += v2 * buf[i2];
          s3 += v3 * buf[i3];
        }
      }
    }
  }
  double s = (s0 + s1) + (s2 + s3);
#pragma unroll
  for (int off = SW / 2; off > 0; off >>= 1) s += __shfl_down(s, off, SW);
  if (LANES == 256) {
    __shared__ double part[4];
    if (sl == 0) part[threadIdx.x / 64] = s;
    __syncthreads();
    if (threadIdx.x == 0 && row < nrows) {
      const double t = (part[0] + part[1]) + (part[2] + part[3]);
      const int d = dest0 + row;
      const double o = accumulate ? buf[d] + t : t;
      buf[d] = o;
      if (velrow && velrow[row] && !isfinite(o)) atomicOr(flag, 1);
    }
  } else if (row < nrows && lane == 0) {
    const int d = dest0 + row;
    const double o = accumulate ? buf[d] + s : s;
    buf[d] = o;
    if (velrow && velrow[row] && !isfinite(o)) atomicOr(flag, 1);
  }
}

// ---------------------------------------------------------------------------------------------
// Down-sweep as batched dense block mat-vec with the operand staged in LDS.
// All rows of a tree node share one operand vector  [ y[i0 .. i0+ni) | x[idx[0 .. nb)] ]  (pivot block
// D^-1 and coupling block -U side by side, row-major, stride wd = ni + nb).  A workgroup takes up
// to `nrows` rows of one node: the operand is gathered once per workgroup into LDS (coalesced /
// index-list gather), after which the lanes only stream the fp64 values from HBM (8 B/nnz, 512 B per
// wave instruction) and read the operand with conflict-free ds_read_b64 — the global-load queue is
// left entirely to the value stream.  LPR lanes cooperate on a row (256/LPR rows in flight per
// workgroup); fixed summation order, no atomics.
// ---------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) FcBlk {
  long long val;  // offset of the first row's values
  int row0;       // first destination row (permuted numbering)
  int nrows;
  int i0, ni;     // pivot part: y[i0 .. i0+ni)
  int idx, nb;    // coupling part: x[idx_list[idx .. idx+nb)]  (entries already offset by N)
};

#define FC_BLK_TILE 2048

// RPS = rows per row slot: a workgroup covers up to RPS * (256 / LPR) rows of its node.  The first
// trip of every row's value stream (4 x LPR values) depends only on the block descriptor, so it is
// issued BEFORE the operand gather: descriptor -> {values | index list -> operand} is a chain of
// three memory round trips instead of five (descriptor, indices, operand, LDS, values).  On the
// small nodes near the leaves (row width <= 4 x LPR) that first trip is the whole row.
template <int LPR, int RPS, typename VT = double, bool NT = false>
__global__ __launch_bounds__(256) void fc_nd_down_block(const FcBlk* __restrict__ blk,
                                                        const int* __restrict__ idxlist,
                                                        const VT* __restrict__ val,
                                                        double* __restrict__ buf, int N,
                                                        const unsigned char* __restrict__ velrow = nullptr, int* __restrict__ flag = nullptr,
                                                        double* __restrict__ out = nullptr) {
  // velrow / flag: as in fc_nd_sweep (here the whole mask, indexed by the permuted row)
  // out (column form of the up-sweep, fc_hip.hip build_up_column): the block is a node's -L block (nb x ni, operand = the node's own y rows,
  // no index part) and its products go to out[row0 + r] -- a scratch slot per (node, boundary row), folded per level by fc_nd_fold1
  __shared__ double xs[FC_BLK_TILE];
  constexpr int SLOTS = 256 / LPR;  // rows in flight per workgroup
  const FcBlk b = blk[blockIdx.x];
  const int slot = threadIdx.x / LPR, l = threadIdx.x % LPR;
  const int wd = b.ni + b.nb;
  const int tl0 = wd < FC_BLK_TILE ? wd : FC_BLK_TILE;
  double acc[RPS], pv[RPS][4];
#pragma unroll
  for (int k = 0; k < RPS; ++k) {
    acc[k] = 0.0;
    const int r = slot + k * SLOTS;
    const VT* __restrict__ v = val + b.val + (long long)r * wd;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = l + u * LPR;
      pv[k][u] = (r < b.nrows && j < tl0) ? fc_val(fc_ld<NT>(v + j)) : 0.0;
    }
  }
  for (int t0 = 0; t0 < wd; t0 += FC_BLK_TILE) {
    const int tl = wd - t0 < FC_BLK_TILE ? wd - t0 : FC_BLK_TILE;
    for (int j = threadIdx.x; j < tl; j += 256) {
      const int col = t0 + j;
      xs[j] = col < b.ni ? buf[b.i0 + col] : buf[idxlist[b.idx + (col - b.ni)]];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RPS; ++k) {
      const int r = slot + k * SLOTS;
      if (r < b.nrows) {
        const VT* __restrict__ v = val + b.val + (long long)r * wd + t0;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int base = 0;
        if (t0 == 0) {
          const int j0 = l, j1 = j0 + LPR, j2 = j1 + LPR, j3 = j2 + LPR;
          s0 = pv[k][0] * (j0 < tl ? xs[j0] : 0.0);
          s1 = pv[k][1] * (j1 < tl ? xs[j1] : 0.0);
          s2 = pv[k][2] * (j2 < tl ? xs[j2] : 0.0);
          s3 = pv[k][3] * (j3 < tl ? xs[j3] : 0.0);
          base = 4 * LPR;
        }
        for (; base < tl; base += 4 * LPR) {
          const int j0 = base + l, j1 = j0 + LPR, j2 = j1 + LPR, j3 = j2 + LPR;
          const double v0 = j0 < tl ? fc_val(fc_ld<NT>(v + j0)) : 0.0, v1 = j1 < tl ? fc_val(fc_ld<NT>(v + j1)) : 0.0;
          const double v2 = j2 < tl ? fc_val(fc_ld<NT>(v + j2)) : 0.0, v3 = j3 < tl ? fc_val(fc_ld<NT>(v + j3)) : 0.0;
          s0 += v0 * (j0 < tl ? xs[j0] : 0.0);
          s1 += v1 * (j1 < tl ? xs[j1] : 0.0);
          s2 += v2 * (j2 < tl ? xs[j2] : 0.0);
          s3 += v3 * (j3 < tl ? xs[j3] : 0.0);
        }
        acc[k] += (s0 + s1) + (s2 + s3);
      }
    }
    if (t0 + FC_BLK_TILE < wd) __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < RPS; ++k) {
    double s = acc[k];
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) s += __shfl_down(s, off, LPR);
    const int r = slot + k * SLOTS;
    if (l == 0 && r < b.nrows) {
      if (out) {
        out[b.row0 + r] = s;
      } else {
        buf[N + b.row0 + r] = s;
        if (velrow && velrow[b.row0 + r] && !isfinite(s)) atomicOr(flag, 1);
      }
    }
  }
}

// The same block product for the levels of SMALL nodes (rows of a few dozen values: the leaves and their parents).  A tile (FcBlk: nrows
// whole rows, row-major, contiguous) is read as ONE flat stream -- thread t takes values t, t + 256, ...: every wave instruction is 512
// consecutive bytes, all of a thread's U loads are issued back to back, whatever the row width -- into LDS, next to the operand; then 8 lanes
// per row form the row sums from LDS in a fixed order.  fc_nd_down_block puts LPR lanes on a row: with rows of 30-90 values a wave
// instruction there touches 8 separate 64-byte pieces and most of a row is a single trip (3.3-4.2 TB/s on the leaf levels of the
// HBM-streaming meshes where the wide levels reach 4.6-5.4).  Tiles hold <= FC_FLAT_CAP values, rows <= FC_FLAT_WD wide.
#define FC_FLAT_CAP 4096
#define FC_FLAT_WD 512
template <int U, bool NT>
__global__ __launch_bounds__(256) void fc_nd_flat_block(const FcBlk* __restrict__ blk, const int* __restrict__ idxlist,
                                                        const double* __restrict__ val, double* __restrict__ buf, int N,
                                                        const unsigned char* __restrict__ velrow, int* __restrict__ flag,
                                                        double* __restrict__ out) {
  __shared__ double xs[FC_FLAT_WD];
  __shared__ double vs[256 * U];
  const FcBlk b = blk[blockIdx.x];
  const int wd = b.ni + b.nb, n = b.nrows * wd, t = threadIdx.x;
  const double* __restrict__ v = val + b.val;
  double pv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {  // unconditional, clamped: a predicated load makes the compiler wait for each one in turn
    const int e = t + 256 * u;
    pv[u] = fc_ld<NT>(v + (e < n ? e : 0));
  }
  for (int j = t; j < wd; j += 256) xs[j] = j < b.ni ? buf[b.i0 + j] : buf[idxlist[b.idx + (j - b.ni)]];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = t + 256 * u;
    if (e < n) vs[e] = pv[u];
  }
  __syncthreads();
  const int l = t & 7;
  for (int r = t >> 3; r < b.nrows; r += 32) {
    const double* __restrict__ row = vs + r * wd;
    double s0 = 0.0, s1 = 0.0;
    for (int j = l; j < wd; j += 16) {
      s0 += row[j] * xs[j];
      if (j + 8 < wd) s1 += row[j + 8] * xs[j + 8];
    }
    double s = s0 + s1;
    s += __shfl_down(s, 4, 8);
    s += __shfl_down(s, 2, 8);
    s += __shfl_down(s, 1, 8);
    if (l == 0) {
      if (out) {
        out[b.row0 + r] = s;
      } else {
        buf[N + b.row0 + r] = s;
        if (velrow && velrow[b.row0 + r] && !isfinite(s)) atomicOr(flag, 1);
      }
    }
  }
}

// column form of the up-sweep: y[row0 + i] += the scratch slots of the row's descendants, in list order (deepest node first: reproducible)
__global__ __launch_bounds__(256) void fc_nd_fold1(int nrows, int row0, const int* __restrict__ fptr, const int* __restrict__ fsrc,
                                                   const double* __restrict__ scratch, double* __restrict__ y) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nrows) return;
  const int i = row0 + t;
  const int q0 = fptr[i], q1 = fptr[i + 1];
  double acc = y[i];
  for (int base = q0; base < q1; base += 8) {
    int id[8];
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) id[u] = base + u < q1 ? fsrc[base + u] : -1;
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = id[u] >= 0 ? scratch[id[u]] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (id[u] >= 0) acc += v[u];
  }
  y[i] = acc;
}

// ---------------------------------------------------------------------------------------------
// Device-side numeric factorisation (fc_refactor): index kernels around the dense front elimination (fc_front.hip.h).
// ---------------------------------------------------------------------------------------------
// fronts[a_dst[k]] = vals[a_src[k]]: every matrix entry has exactly one slot in exactly one front
// (threads beyond n: the permuted copy of the matrix for the residual monitor, out2[k2] = vals[src2[k2]] -- the same gather from the
//  same values, riding in this launch instead of one of its own)
__global__ void fc_front_scatter(int64_t n, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                 const double* __restrict__ vals, double* __restrict__ fronts, int64_t n2 = 0,
                                 const int64_t* __restrict__ src2 = nullptr, double* __restrict__ out2 = nullptr) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n)
    fronts[dst[k]] = vals[src[k]];
  else if (k - n < n2)
    out2[k - n] = vals[src2[k - n]];
}
// fronts[slot[k]] += value[k]: diagonal shifts applied after the scatter (pressure pin of enclosed flows)
// (multi-GPU: slots in [skip0, skip1) — the root front, which is summed over the ranks — are left to the lead rank)
__global__ void fc_front_shift(int n, const int64_t* __restrict__ slot, const double* __restrict__ value,
                               double* __restrict__ fronts, int64_t skip0, int64_t skip1) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n && !(slot[k] >= skip0 && slot[k] < skip1)) fronts[slot[k]] += value[k];
}
__global__ void fc_gather64(int64_t n, const int64_t* __restrict__ src, const double* __restrict__ vals,
                            double* __restrict__ out) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = vals[src[k]];
}
// out[k] = vals[src[k]], 0 where src[k] < 0 (padding of the one-launch apply's row tiles)
__global__ void fc_gather64_pad(int64_t n, const int64_t* __restrict__ src, const double* __restrict__ vals,
                                double* __restrict__ out) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = src[k] >= 0 ? vals[src[k]] : 0.0;
}
// extend-add: parent[p[i], p[j]] += child update block [i, j].  One launch covers the c-th child of
// every parent of a level, so no two workgroups of a launch touch the same parent entry.
struct __attribute__((aligned(16))) FcExt {
  long long src;  // child front: offset of its update block (row stride nfc)
  long long dst;  // parent front offset (row stride nfp)
  int nfc, nfp;
  int nbc;   // order of the update block
  int poff;  // offset of the child's position list
};
#define FC_EXT_ROWS 8
__global__ __launch_bounds__(256) void fc_extend_add(const FcExt* __restrict__ ext, const int* __restrict__ p,
                                                     double* __restrict__ fronts) {
  const FcExt d = ext[blockIdx.y];
  const int i0 = blockIdx.x * FC_EXT_ROWS;
  if (i0 >= d.nbc) return;
  const int* __restrict__ pp = p + d.poff;
  for (int i = i0; i < i0 + FC_EXT_ROWS && i < d.nbc; ++i) {
    const double* __restrict__ srow = fronts + d.src + (long long)i * d.nfc;
    double* __restrict__ drow = fronts + d.dst + (long long)pp[i] * d.nfp;
    for (int j = threadIdx.x; j < d.nbc; j += 256) drow[pp[j]] += srow[j];
  }
}

// The same sums in ONE launch per level: a workgroup owns `rows` (16, 8 or 4) rows of one PARENT front and adds the update blocks of
// all its children in slot order -- wave w takes the parent rows r with r % 4 == w, so every parent entry is touched by one wave
// only, children one after the other (deterministic, no atomics).  A child's position list is ascending; a wave finds the
// child rows that land in its parent rows with one coalesced pass over the list (ballot), the list itself sits in LDS
// for the column scatter.  par[q] = {first FcExt of parent q, number of children} (children of a parent are contiguous, in
// slot order).
#define FC_EXTP_ROWS 16
#define FC_EXTP_LDS 8192  // position-list entries held in LDS (longer lists are read from global memory)
struct __attribute__((aligned(8))) FcExtPar {
  int first, count;
};
__global__ __launch_bounds__(256) void fc_extend_add_parents(const FcExtPar* __restrict__ par, const FcExt* __restrict__ ext,
                                                             const int* __restrict__ p, double* __restrict__ fronts, int rows) {
  __shared__ int pl[FC_EXTP_LDS];
  const FcExtPar pq = par[blockIdx.y];
  const int r0 = blockIdx.x * rows;  // rows: parent rows per workgroup (a multiple of 4; fewer on levels of few, large parents)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int c = 0; c < pq.count; ++c) {
    const FcExt d = ext[pq.first + c];
    if (r0 >= d.nfp) return;  // (all children of a parent carry the parent's order: uniform)
    const int* __restrict__ pp = p + d.poff;
    // does any of this child's rows land in [r0, r0 + ROWS)?  The list is ascending: compare its ends first.
    if (pp[0] >= r0 + rows || pp[d.nbc - 1] < r0) continue;  // uniform
    const bool in_lds = d.nbc <= FC_EXTP_LDS;
    __syncthreads();  // the previous child's list is no longer read
    if (in_lds)
      for (int j = threadIdx.x; j < d.nbc; j += 256) pl[j] = pp[j];
    __syncthreads();
    for (int base = 0; base < d.nbc; base += 64) {
      const int i = base + lane;
      const int row = i < d.nbc ? (in_lds ? pl[i] : pp[i]) : -1;
      if (__builtin_amdgcn_readfirstlane(row) >= r0 + rows) break;  // ascending list: nothing further down lands here
      unsigned long long mask = __ballot(row >= r0 && row < r0 + rows && ((row - r0) & 3) == wave);
      while (mask) {
        const int b = __builtin_ctzll(mask);
        mask &= mask - 1;
        const int prow = __builtin_amdgcn_readlane(row, b);
        const double* __restrict__ srow = fronts + d.src + (long long)(base + b) * d.nfc;
        double* __restrict__ drow = fronts + d.dst + (long long)prow * d.nfp;
        if (in_lds)
          for (int j = lane; j < d.nbc; j += 64) drow[pl[j]] += srow[j];
        else
          for (int j = lane; j < d.nbc; j += 64) drow[pp[j]] += srow[j];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// small vector kernels
// ---------------------------------------------------------------------------------------------
__global__ void fc_copy(int n, const double* __restrict__ a, double* __restrict__ b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = a[i];
}
__global__ void fc_axpy(int n, double alpha, const double* __restrict__ a, double* __restrict__ b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] += alpha * a[i];
}
__global__ void fc_gather_perm(int n, const int* __restrict__ perm, const double* __restrict__ src,
                               double* __restrict__ dst) {  // dst[i] = src[perm[i]]
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[perm[i]];
}
__global__ void fc_scatter_perm(int n, const int* __restrict__ perm, const double* __restrict__ src,
                                const double* __restrict__ add, double* __restrict__ dst) {  // dst[perm[i]] = src[i] (+ add[i])
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[perm[i]] = add ? src[i] + add[i] : src[i];
}

// Krylov building blocks (fc_solve with FC_METHOD_BICGSTAB): two dot products per pass with a fixed
// workgroup -> partial -> fc_reduce_final order (reproducible), and a three-term linear combination.
// (multi-GPU: rowkind != nullptr restricts the sums to the rows this rank accounts for -- its own, the root's on the lead
// rank; the caller sums the results over the ranks)
__global__ __launch_bounds__(256) void fc_dots2(int n, const double* __restrict__ a, const double* __restrict__ b,
                                                const double* __restrict__ c, const double* __restrict__ d,
                                                double* __restrict__ partial, const unsigned char* __restrict__ rowkind = nullptr,
                                                int lead = 1) {
  double s0 = 0.0, s1 = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    if (rowkind) {
      const int kind = rowkind[i];
      if (!(kind == 1 || (kind == 2 && lead))) continue;
    }
    s0 += a[i] * b[i];
    s1 += c[i] * d[i];
  }
  __shared__ double r0[256], r1[256];
  r0[threadIdx.x] = s0;
  r1[threadIdx.x] = s1;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      r0[threadIdx.x] += r0[threadIdx.x + st];
      r1[threadIdx.x] += r1[threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = r0[0];
    partial[gridDim.x + blockIdx.x] = r1[0];
  }
}
// out = c0 v0 + c1 v1 (+ c2 v2); out may alias any input
__global__ void fc_lin3(int n, double* out, double c0, const double* v0, double c1, const double* v1, double c2,
                        const double* v2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    double s = c0 * v0[i] + c1 * v1[i];
    if (v2) s += c2 * v2[i];
    out[i] = s;
  }
}

// The un-fused end of a step (iterative refinement, Krylov solves): x is the new (v, p) in the permuted numbering, already in
// place in the state ring (nothing is copied or shifted: fc_hip.hip "state ring").  Non-finite flag over the velocity rows
// (reference flowsolver.py:730-731,816-819) and, fused, the per-row share of the perturbation energy 1/2 u^T M u
// (flowsolver.py:827-829) with the velocity mass matrix in permuted numbering (Mp rows of pressure dofs are empty) -> one
// partial per block, summed by fc_final.
__global__ __launch_bounds__(256) void fc_finish(int N, const unsigned char* __restrict__ velrow, const double* __restrict__ x,
                                                 int* __restrict__ flag, const int* __restrict__ m_rowptr,
                                                 const int* __restrict__ m_col, const double* __restrict__ m_val,
                                                 double* __restrict__ e_partial, const unsigned char* __restrict__ rowkind) {
  // 8 lanes per (permuted) row: they share the mass-matrix row of the energy term (coalesced 8 x 12 B per trip)
  constexpr int LANES = 8, RPB = 256 / LANES;
  const int lane = threadIdx.x % LANES;
  const int i = blockIdx.x * RPB + threadIdx.x / LANES;
  double e = 0.0;
  if (i < N && velrow[i] && (!rowkind || rowkind[i] != 0)) {
    const double v = x[i];
    if (m_rowptr) {
      double s = 0.0;
      for (int k = m_rowptr[i] + lane; k < m_rowptr[i + 1]; k += LANES) s += m_val[k] * x[m_col[k]];
      e = v * s;
    }
    if (lane == 0 && !isfinite(v)) atomicOr(flag, 1);
  }
  if (e_partial) {
    __shared__ double red[256];
    red[threadIdx.x] = e;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
      __syncthreads();
    }
    if (threadIdx.x == 0) e_partial[blockIdx.x] = red[0];
  }
}

// agent-scope relaxed atomic accesses (sc1: served at the memory-side coherence point, no L2 write-back or invalidate
// involved): how workgroups of ONE launch on different XCDs hand data to each other (fc_tail's last arriver, fc_nd_dag)
typedef unsigned long long fc_u64;
__device__ __forceinline__ double fc_ld_sc1(const double* p) {
  const fc_u64 v = __hip_atomic_load(reinterpret_cast<const fc_u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __longlong_as_double((long long)v);
}
__device__ __forceinline__ void fc_st_sc1(double* p, double x) {
  __hip_atomic_store(reinterpret_cast<fc_u64*>(p), (fc_u64)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The step record (y[n_sens], E, |r|^2, |b|^2, flag) may live in host-mapped memory that the host polls
// instead of synchronising the stream.  ONE thread writes the whole record, then two checksums over the bit
// patterns of every word and of the sequence number (an XOR, and a position-weighted sum modulo 2^64 with odd
// weights: two torn words that cancel in the XOR do not cancel in the weighted sum) and the sequence number
// itself: the host accepts a record only when the sequence number matches AND both checksums agree with what it
// reads, so it is immune to the order in which the individual writes become visible across PCIe.
__device__ inline void fc_publish(const double* ysrc, int n_sens, double E, double r0, double r1, double fl,
                                  double* __restrict__ y, double* __restrict__ E_out, double* __restrict__ r_out,
                                  double* __restrict__ flag_out, double* __restrict__ seq_out, double seq) {
  typedef unsigned long long u64;
  u64 x = (u64)__double_as_longlong(seq);
  u64 w = x;  // weights 1, 3, 5, ... (odd: invertible modulo 2^64)
  u64 k = 3;
  for (int s = 0; s < n_sens; ++s, k += 2) {
    const double v = ysrc[s];
    if (y) y[s] = v;
    x ^= (u64)__double_as_longlong(v);
    w += k * (u64)__double_as_longlong(v);
  }
  if (E_out) E_out[0] = E;
  if (r_out) {
    r_out[0] = r0;
    r_out[1] = r1;
  }
  if (flag_out) flag_out[0] = fl;
  const u64 tail[4] = {(u64)__double_as_longlong(E), (u64)__double_as_longlong(r0), (u64)__double_as_longlong(r1),
                       (u64)__double_as_longlong(fl)};
  for (int i = 0; i < 4; ++i, k += 2) {
    x ^= tail[i];
    w += k * tail[i];
  }
  if (seq_out) {
    // no fences: the host does not rely on the order in which these words arrive (it re-checks the
    // checksums until they fit), and the end of the kernel makes all of them visible
    seq_out[1] = __longlong_as_double((long long)x);
    seq_out[2] = __longlong_as_double((long long)w);
    seq_out[0] = seq;
  }
}

// tail of a step, ONE workgroup: folds the energy partials (-> E = 1/2 sum) and, if present, the
// residual partials (sum r^2, sum b^2), evaluates the sensor rows (y_s = sum_k w[k] x[idxp[k]]: the new solution in the
// permuted numbering, idxp = the sensor dofs' permuted positions; sensor.py:96-98,166-197) and publishes everything to the (host-mapped) record with its checksums
// (fc_publish: no fence, the host validates what it reads), so the host can poll it instead of
// synchronising the stream.  Fixed summation order => reproducible.
// SC1: the partials were written by other workgroups of the SAME launch (fc_tail's last arriver) and are read at
// the coherence point; the summation order is the same either way.
template <bool SC1>
__device__ __forceinline__ void fc_final_body(int n_e, const double* __restrict__ e_partial, double* __restrict__ E_out, int n_r,
                                              const double* __restrict__ r_partial, double* __restrict__ r_out, int n_sens,
                                              const int* __restrict__ s_rowptr, const int* __restrict__ s_idx,
                                              const double* __restrict__ s_w, const double* __restrict__ up, double* __restrict__ y,
                                              const int* __restrict__ flag, double* __restrict__ flag_out,
                                              double* __restrict__ seq_out, double seq) {
  auto ld = [](const double* q) -> double { return SC1 ? fc_ld_sc1(q) : *q; };
  // flag word of the record: bit 0 = non-finite velocity -- partitioned runs sum the word over the ranks
  __shared__ double red[3][256];
  const int t = threadIdx.x;
  // a single workgroup is pure latency: issue every load up front (8 partials per thread and array,
  // predicated; the sensor rows' index/weight chain right behind them), reduce afterwards
  constexpr int U = 8;
  double pe[U], pr[U], pb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int i = t + 256 * u;
    pe[u] = (e_partial && i < n_e) ? ld(e_partial + i) : 0.0;
    pr[u] = (r_partial && i < n_r) ? ld(r_partial + i) : 0.0;
    pb[u] = (r_partial && i < n_r) ? ld(r_partial + n_r + i) : 0.0;
  }
  // sensors: one wave per row, waves take rows round-robin; results parked in LDS for the publisher
  __shared__ double ysh[64];
  const int wave = t >> 6, lane = t & 63;
  for (int s = wave; s < n_sens; s += 4) {
    double acc = 0.0;
    for (int k = s_rowptr[s] + lane; k < s_rowptr[s + 1]; k += 64) acc += s_w[k] * up[s_idx[k]];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) ysh[s] = acc;
  }
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    a0 += pe[u];
    a1 += pr[u];
    a2 += pb[u];
  }
  if (e_partial)
    for (int i = t + 256 * U; i < n_e; i += 256) a0 += ld(e_partial + i);
  if (r_partial)
    for (int i = t + 256 * U; i < n_r; i += 256) {
      a1 += ld(r_partial + i);
      a2 += ld(r_partial + n_r + i);
    }
  red[0][t] = a0;
  red[1][t] = a1;
  red[2][t] = a2;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (t < st) {
      red[0][t] += red[0][t + st];
      red[1][t] += red[1][t + st];
      red[2][t] += red[2][t + st];
    }
    __syncthreads();
  }
  if (t == 0) {
    const double fl = flag ? (double)((SC1 ? __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : flag[0]) & 1) : 0.0;
    fc_publish(ysh, n_sens, e_partial ? 0.5 * red[0][0] : 0.0, r_partial ? red[1][0] : 0.0, r_partial ? red[2][0] : 0.0,
               fl, y, E_out, r_out, flag_out, seq_out, seq);
  }
}

__global__ __launch_bounds__(256) void fc_final(int n_e, const double* __restrict__ e_partial,
                                                double* __restrict__ E_out, int n_r,
                                                const double* __restrict__ r_partial,
                                                double* __restrict__ r_out, int n_sens,
                                                const int* __restrict__ s_rowptr,
                                                const int* __restrict__ s_idx,
                                                const double* __restrict__ s_w,
                                                const double* __restrict__ up, double* __restrict__ y,
                                                const int* __restrict__ flag, double* __restrict__ flag_out,
                                                double* __restrict__ seq_out, double seq) {
  fc_final_body<false>(n_e, e_partial, E_out, n_r, r_partial, r_out, n_sens, s_rowptr, s_idx, s_w, up, y, flag, flag_out, seq_out, seq);
}

// Overlapped tail (single GPU): what the host WAITS for is only this -- the sensor rows on the new solution and the non-finite flag
// (raised by the down-sweep launches themselves), published right behind the last sweep launch.  Residual monitor and energy follow on
// a second stream while the host and the next step go on (fc_final_late publishes them to a record of their own).
__global__ __launch_bounds__(256) void fc_early(int n_sens, const int* __restrict__ s_rowptr, const int* __restrict__ s_idxp,
                                                const double* __restrict__ s_w, const double* __restrict__ x, double* __restrict__ y,
                                                int* __restrict__ flag, double* __restrict__ flag_out, double* __restrict__ seq_out, double seq,
                                                fc_u64* __restrict__ solved) {
  __shared__ double ysh[64];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  for (int s = wave; s < n_sens; s += 4) {
    double acc = 0.0;
    for (int k = s_rowptr[s] + lane; k < s_rowptr[s + 1]; k += 64) acc += s_w[k] * x[s_idxp[k]];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) ysh[s] = acc;
  }
  __syncthreads();
  if (t == 0) {
    const int fl = flag[0] & 1;
    flag[0] = 0;  // per step (the next step's sweeps raise it again if need be)
    fc_publish(ysh, n_sens, 0.0, 0.0, 0.0, (double)fl, y, y + 64, y + 65, flag_out, seq_out, seq);
    // the side stream's gate (fc_wait_solved): every kernel of this step's solve finished before this one started
    __hip_atomic_store(solved, (fc_u64)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// first kernel of a step's side-stream batch: one thread waits until the main stream's fc_early of step `seq` has run (bounded: a solve
// that never arrives -- a failed launch, a main stream held up for longer than ~50 ms -- ends the wait and leaves *gave_up = 1;
// fc_final_late then publishes the flag INSIDE the checksummed late record and the host reports FC_ERR_HIP for that step instead of
// accepting a residual / energy computed on buffers the main stream may still be writing).  The word is rewritten by every gate (0 or 1).
// HIP events would do the same across streams, but cost the host ~7 us per record / wait pair on this runtime.
__global__ void fc_wait_solved(const fc_u64* __restrict__ solved, fc_u64 seq, int* __restrict__ gave_up, long max_spin) {
  for (long spin = 0; spin < max_spin; ++spin) {
    if (__hip_atomic_load(solved, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= seq) {
      *gave_up = 0;
      return;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  *gave_up = 1;
}
// the late record: rec[0] = E, rec[1] = sum r^2, rec[2] = sum b^2, rec[3] = seq, rec[4], rec[5] = checksums (as fc_publish) over the three
// values, the gate's give-up flag rec[6] and seq
__global__ __launch_bounds__(256) void fc_final_late(int n_e, const double* __restrict__ e_partial, int n_r, const double* __restrict__ r_partial,
                                                     double* __restrict__ rec, double seq, const int* __restrict__ gave_up) {
  __shared__ double red[3][256];
  const int t = threadIdx.x;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  constexpr int U = 8;
  double pe[U], pr[U], pb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int i = t + 256 * u;
    pe[u] = (e_partial && i < n_e) ? e_partial[i] : 0.0;
    pr[u] = (r_partial && i < n_r) ? r_partial[i] : 0.0;
    pb[u] = (r_partial && i < n_r) ? r_partial[n_r + i] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    a0 += pe[u];
    a1 += pr[u];
    a2 += pb[u];
  }
  if (e_partial)
    for (int i = t + 256 * U; i < n_e; i += 256) a0 += e_partial[i];
  if (r_partial)
    for (int i = t + 256 * U; i < n_r; i += 256) {
      a1 += r_partial[i];
      a2 += r_partial[n_r + i];
    }
  red[0][t] = a0;
  red[1][t] = a1;
  red[2][t] = a2;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (t < st) {
      red[0][t] += red[0][t + st];
      red[1][t] += red[1][t + st];
      red[2][t] += red[2][t + st];
    }
    __syncthreads();
  }
  if (t == 0) {
    typedef unsigned long long u64;
    const double v[4] = {e_partial ? 0.5 * red[0][0] : 0.0, r_partial ? red[1][0] : 0.0, r_partial ? red[2][0] : 0.0, (gave_up && *gave_up) ? 1.0 : 0.0};
    u64 x = (u64)__double_as_longlong(seq), w = x, k = 3;
    for (int i = 0; i < 4; ++i, k += 2) {
      rec[i < 3 ? i : 6] = v[i];
      x ^= (u64)__double_as_longlong(v[i]);
      w += k * (u64)__double_as_longlong(v[i]);
    }
    rec[4] = __longlong_as_double((long long)x);
    rec[5] = __longlong_as_double((long long)w);
    rec[3] = seq;
  }
}

// what the LAST workgroup of a fused fc_tail does instead of a separate fc_final launch (enabled: cnt != nullptr)
struct FcFin {
  unsigned* cnt;         // arrival counters: group g at cnt[32 g] (one 128-byte line each), the top counter at cnt[32 n_groups]
  int group, n_groups;   // workgroups per group; groups
  int n_sens;
  const int* s_rowptr;
  const int* s_idxp;     // sensor dofs as positions in the permuted solution
  const double* s_w;
  double* y;
  double* E_out;
  double* r_out;
  double* flag_out;
  double* seq_out;
  double seq;
};

// ---------------------------------------------------------------------------------------------
// Tail of a step in ONE launch (fc_final follows).  The new state needs no work: the solution half of the sweep buffer IS
// the new (v, p) -- state vectors live in the solver's permuted numbering and the work buffers rotate (fc_hip.hip "state
// ring"), so nothing is scattered, copied or shifted (rounds 1-3 moved ~60 B per dof here, 52 MB per step on cavity_fine).
// Row workgroups: 8 lanes per permuted row evaluate r_i = b_i - (A x)_i (flowsolver.py:729's solve, checked) and test the
// row's own entry for finiteness (velocity rows: flowsolver.py:731,816-819).  Check workgroups (only when the residual monitor
// is off for this step): the finiteness test alone, one row per thread.  Cell workgroups: the energy integral of the new
// velocity, element by element.  Every workgroup leaves (sum r^2 | sum b^2 | sum e) in `partial` (three arrays of gridDim.x)
// for fc_final.
// (Folding fc_final in as well, "last workgroup to arrive reduces": with an agent-scope release per workgroup it cost
// 10x what the launch saves; with sc1 partials + drained two-level arrival counters (FcFin, FC_FUSED_FINAL=1) it
// costs exactly what the separate launch costs.  Kept opt-in.)
#define FC_TAIL_CHECK 4  // rows per thread of a check workgroup
template <bool FUSED>
__global__ __launch_bounds__(256) void fc_tail(
    int N, const unsigned char* __restrict__ velrow, const double* __restrict__ x, const double* __restrict__ b,
    const int* __restrict__ a_rowptr, const int* __restrict__ a_col, const double* __restrict__ a_val,
    int n_row_blocks, int n_check_blocks, int reps, int nc, const int* __restrict__ cnp, const double* __restrict__ geom,
    const unsigned char* __restrict__ rowkind, const int* __restrict__ cell_list, int ncl,
    int* __restrict__ flag, double* __restrict__ partial, FcFin fin) {
  constexpr int LANES = 8, RPB = 256 / LANES;
  const int t = threadIdx.x, lane = t % LANES;
  const int G = gridDim.x;
  double r2 = 0.0, b2 = 0.0, e = 0.0;
  bool bad = false;
  // the cell workgroups (a chain of two dependent gathers per lane) come FIRST in the grid, so that their latency
  // overlaps with the row workgroups' streaming instead of forming the launch's tail
  const int n_cell_blocks = G - n_row_blocks - n_check_blocks;
  const int kb = (int)blockIdx.x - n_cell_blocks;  // check block of this workgroup (< 0: a cell workgroup)
  const int rb = kb - n_check_blocks;              // row block (>= 0: a row workgroup)
  if (kb >= 0 && rb < 0) {
#pragma unroll
    for (int u = 0; u < FC_TAIL_CHECK; ++u) {
      const int i = (kb * FC_TAIL_CHECK + u) * 256 + t;
      // multi-GPU (rowkind != nullptr): 0 = another rank's row (never computed here), 1 = owned, 2 = replicated root row
      if (i < N && velrow[i] && (!rowkind || rowkind[i] != 0)) bad |= !isfinite(x[i]);
    }
  } else if (rb >= 0) {
    // rows: residual monitor (`reps` row groups per workgroup keep the number of partials that fc_final folds alone
    // <= ~2000 on large meshes)
    for (int rep = 0; rep < reps; ++rep) {
      const int i = (rb * reps + rep) * RPB + t / LANES;
      double sa = 0.0;
      // multi-GPU (rowkind != nullptr): only the rows a rank owns (kind 1) enter its residual sums
      const int kind = i < N ? (rowkind ? rowkind[i] : 1) : 0;
      if (kind == 1) {
        const int k0 = a_rowptr[i], k1 = a_rowptr[i + 1];
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int base = k0; base < k1; base += 4 * LANES) {
          const int j0 = base + lane, j1 = j0 + LANES, j2 = j1 + LANES, j3 = j2 + LANES;
          const int c0 = j0 < k1 ? a_col[j0] : 0, c1 = j1 < k1 ? a_col[j1] : 0;
          const int c2 = j2 < k1 ? a_col[j2] : 0, c3 = j3 < k1 ? a_col[j3] : 0;
          const double v0 = j0 < k1 ? a_val[j0] : 0.0, v1 = j1 < k1 ? a_val[j1] : 0.0;
          const double v2 = j2 < k1 ? a_val[j2] : 0.0, v3 = j3 < k1 ? a_val[j3] : 0.0;
          s0 += v0 * x[c0];
          s1 += v1 * x[c1];
          s2 += v2 * x[c2];
          s3 += v3 * x[c3];
        }
        sa = (s0 + s1) + (s2 + s3);
      }
#pragma unroll
      for (int off = LANES / 2; off > 0; off >>= 1) sa += __shfl_down(sa, off, LANES);
      if (lane == 0 && kind != 0) {
        if (velrow[i]) bad |= !isfinite(x[i]);
        if (kind == 1) {
          const double bb = b[i], res = bb - sa;
          r2 += res * res;
          b2 += bb * bb;
        }
      }
    }
  } else if (cnp) {
    // cells: perturbation energy  int |u|^2  of the NEW velocity (utils_flowsolver.py:195-203 / flowsolver.py:827-829),
    // read from the permuted solution through the cells' permuted node table; lane q = Radon point q (degree-4
    // integrand: the 7-point rule is exact), 32 cells per workgroup.  2 MB instead of the 11 MB of mass-matrix rows.
    for (int rep = 0; rep < reps; ++rep) {
      const int cl = ((int)blockIdx.x * reps + rep) * RPB + t / LANES;
      const int c = cl < ncl ? (cell_list ? cell_list[cl] : cl) : 0;  // multi-GPU: this rank's cells
      double w = 0.0;
      if (cl < ncl && lane < FC_NQ) {
        double ux = 0.0, uy = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const double ph = c_phi2[lane * 6 + a];
          ux += ph * x[cnp[(size_t)a * nc + c]];
          uy += ph * x[cnp[(size_t)(6 + a) * nc + c]];
        }
        w = c_qw[lane] * 0.5 * geom[4 * (size_t)nc + c] * (ux * ux + uy * uy);
      }
#pragma unroll
      for (int off = LANES / 2; off > 0; off >>= 1) w += __shfl_down(w, off, LANES);
      if (lane == 0) e += w;
    }
  }
  __shared__ double red[3][256];
  red[0][t] = r2;
  red[1][t] = b2;
  red[2][t] = e;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (t < st) {
      red[0][t] += red[0][t + st];
      red[1][t] += red[1][t + st];
      red[2][t] += red[2][t + st];
    }
    __syncthreads();
  }
  if (!FUSED) {
    if (bad) atomicOr(flag, 1);
    if (t == 0) {
      partial[blockIdx.x] = red[0][0];
      partial[G + blockIdx.x] = red[1][0];
      partial[2 * G + blockIdx.x] = red[2][0];
    }
    return;
  }
  const int any_bad = __syncthreads_or(bad ? 1 : 0);
  // fused final: partials and the flag go to the coherence point (sc1), are drained, then the workgroup arrives on its
  // group's counter and the group's last arriver on the top counter (two levels: ~12 ns per same-address atomic would
  // serialise thousands of arrivals on one word).  The last arriver of all folds the partials in fc_final's fixed order,
  // evaluates the sensors from the solution (complete before this launch) and publishes; counters reset themselves.
  __shared__ int last;
  if (t == 0) {
    if (any_bad) atomicOr(flag, 1);
    fc_st_sc1(partial + blockIdx.x, red[0][0]);
    fc_st_sc1(partial + G + blockIdx.x, red[1][0]);
    fc_st_sc1(partial + 2 * G + blockIdx.x, red[2][0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int grp = blockIdx.x / fin.group;
    const int gsz = (grp + 1) * fin.group <= G ? fin.group : G - grp * fin.group;
    int l = 0;
    unsigned* c = fin.cnt + 32 * (size_t)grp;
    if (__hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)gsz - 1u) {
      __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned* top = fin.cnt + 32 * (size_t)fin.n_groups;
      if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)fin.n_groups - 1u) {
        __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        l = 1;
      }
    }
    last = l;
  }
  __syncthreads();
  if (!last) return;
  const bool res = a_rowptr != nullptr;
  fc_final_body<true>(G, cnp ? partial + 2 * (size_t)G : nullptr, fin.E_out, res ? G : 0, res ? partial : nullptr, fin.r_out, fin.n_sens,
                      fin.s_rowptr, fin.s_idxp, fin.s_w, x, fin.y, flag, fin.flag_out, fin.seq_out, fin.seq);
}

// multi-GPU: energy share of this rank's cells, 1/2 ∫ |u|^2 (degree-4 integrand: exact with the 7-pt rule);
// one partial per block
__global__ __launch_bounds__(256) void fc_energy_elem(int nc, const int* __restrict__ cnp,
                                                      const double* __restrict__ geom,
                                                      const double* __restrict__ u,
                                                      const int* __restrict__ cell_list, int ncl,
                                                      double* __restrict__ partial) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double e = 0.0;
  if (t < ncl) {
    const int c = cell_list ? cell_list[t] : t;
    const double hdet = 0.5 * geom[4 * (size_t)nc + c];
    double ax[6], ay[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      ax[a] = u[cnp[(size_t)a * nc + c]];  // u: permuted state, cnp: the cells' permuted node table
      ay[a] = u[cnp[(size_t)(6 + a) * nc + c]];
    }
#pragma unroll
    for (int q = 0; q < FC_NQ; ++q) {
      double ux = 0.0, uy = 0.0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        ux += c_phi2[q * 6 + a] * ax[a];
        uy += c_phi2[q * 6 + a] * ay[a];
      }
      e += c_qw[q] * hdet * (ux * ux + uy * uy);
    }
  }
  __shared__ double red[256];
  red[threadIdx.x] = e;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// multi-GPU: v[i] = 0 on the rows this rank does not account for (another rank's rows; the root's rows unless lead)
__global__ void fc_mask_rows(int n, const unsigned char* __restrict__ rowkind, int lead, double* __restrict__ v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int kind = rowkind[i];
    if (!(kind == 1 || (kind == 2 && lead))) v[i] = 0.0;
  }
}

// multi-GPU: after the all-reduce of the step tail [y(64) | E | r2 | b2 | ... | flag@72] publish it
__global__ void fc_publish_tail(const double* __restrict__ tail, double* __restrict__ y, int n_sens,
                                double* __restrict__ E, double* __restrict__ r, double* __restrict__ flag_out,
                                double* __restrict__ seq_out, double seq) {
  if (threadIdx.x == 0) fc_publish(tail, n_sens, tail[64], tail[65], tail[66], tail[72], y, E, r, flag_out, seq_out, seq);
}

// one wave per sensor row: y_s = sum_k w[k] up[idx[k]]   (sensor.py:96-98,166-197)
__global__ void fc_sensors(int n_sens, const int* __restrict__ rowptr, const int* __restrict__ idx,
                           const double* __restrict__ w, const double* __restrict__ up,
                           double* __restrict__ y) {
  const int s = blockIdx.x;
  if (s >= n_sens) return;
  double acc = 0.0;
  for (int k = rowptr[s] + threadIdx.x; k < rowptr[s + 1]; k += 64) acc += w[k] * up[idx[k]];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (threadIdx.x == 0) y[s] = acc;
}

// energy partials: sum over velocity rows r of u[r] * (M u)[r]; 8 lanes per row
__global__ __launch_bounds__(256) void fc_energy_partial(int nrows, const int* __restrict__ rowptr,
                                                         const int* __restrict__ col,
                                                         const double* __restrict__ val,
                                                         const double* __restrict__ u,
                                                         double* __restrict__ partial) {
  constexpr int LANES = 8, RPB = 32;
  const int lane = threadIdx.x % LANES;
  const int row = blockIdx.x * RPB + threadIdx.x / LANES;
  double s = 0.0;
  if (row < nrows) {
    for (int k = rowptr[row] + lane; k < rowptr[row + 1]; k += LANES) {
      const int cidx = col[k];
      if (cidx < nrows) s += val[k] * u[cidx];
    }
    s *= u[row];
  }
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// out[seg] = scale * sum(partial[seg*n .. seg*n + n)) in a fixed order (one block per segment)
__global__ void fc_reduce_final(int n, const double* __restrict__ partial, double scale,
                                double* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  partial += (size_t)blockIdx.x * n;
  out += blockIdx.x;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = scale * red[0];
}

// ---------------------------------------------------------------------------------------------
// Device-resident Krylov drivers (fc_solve with FC_METHOD_BICGSTAB / FC_METHOD_GMRES): every scalar of the
// recurrences lives in `ks` on the device, the vector kernels read their coefficients from there and turn into
// no-ops once ks[KS_STATE] says "done", so the host enqueues several iterations ahead and looks at the state word
// only every few iterations (no synchronisation per dot product).
// ---------------------------------------------------------------------------------------------
enum { KS_STATE = 0, KS_ITERS, KS_BNORM2, KS_RNORM2, KS_RHO, KS_RHO_OLD, KS_ALPHA, KS_OMEGA, KS_COEF = 8 /* 4 triples */, KS_D0 = 24, KS_D1, KS_SIZE = 32 };
// state: 0 running, 1 converged, 2 (BiCGStab) converged at the half step: finish with omega = 0, < 0 breakdown

// out = c[0] v0 + c[1] v1 (+ c[2] v2), coefficients on the device; no-op when the solver is done
__global__ void fc_lin3_dev(int n, double* out, const double* __restrict__ c, const double* v0, const double* v1, const double* v2,
                            const double* __restrict__ ks) {
  if (ks[KS_STATE] == 1.0 || ks[KS_STATE] < 0.0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    double s = c[0] * v0[i] + c[1] * v1[i];
    if (v2) s += c[2] * v2[i];
    out[i] = s;
  }
}

// sum the two partial arrays of fc_dots2 (grid blocks each) into ks[KS_D0], ks[KS_D1] (fixed order), then thread 0
// advances the BiCGStab recurrences (phase = which pair of dot products just finished)
__global__ __launch_bounds__(256) void fc_bicg_phase(int phase, int nblk, const double* __restrict__ partial, double* ks, double rtol,
                                                     int reduce, int update) {
  __shared__ double r0[256], r1[256];
  if (reduce) {
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) {
      s0 += partial[i];
      s1 += partial[nblk + i];
    }
    r0[threadIdx.x] = s0;
    r1[threadIdx.x] = s1;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) {
        r0[threadIdx.x] += r0[threadIdx.x + st];
        r1[threadIdx.x] += r1[threadIdx.x + st];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      ks[KS_D0] = r0[0];
      ks[KS_D1] = r1[0];
    }
  }
  if (!update || threadIdx.x != 0) return;
  const double d0 = ks[KS_D0], d1 = ks[KS_D1];
  double* cS = ks + KS_COEF;       // s = r - alpha v
  double* cX = ks + KS_COEF + 3;   // x = x + alpha ph + omega sh
  double* cR = ks + KS_COEF + 6;   // r = s - omega t
  double* cP = ks + KS_COEF + 9;   // p = r + beta p - beta omega v
  const double st = ks[KS_STATE];
  if (phase == 0) {  // d0 = r.r = |b|^2, d1 = rh.r
    ks[KS_BNORM2] = d0;
    ks[KS_RNORM2] = d0;
    ks[KS_RHO] = d1;
    ks[KS_RHO_OLD] = 1.0;
    ks[KS_ALPHA] = 1.0;
    ks[KS_OMEGA] = 1.0;
    ks[KS_ITERS] = 0.0;
    ks[KS_STATE] = d0 > 0.0 ? 0.0 : 1.0;  // b = 0 -> x = 0
    cP[0] = 1.0, cP[1] = 0.0, cP[2] = 0.0;  // first iteration: p = r
    return;
  }
  if (phase == 5) {  // restart after a breakdown, rh = r = b - A x: d0 = r.r, d1 = rh.r
    ks[KS_RNORM2] = d0;
    ks[KS_RHO] = d1;
    ks[KS_RHO_OLD] = 1.0;
    ks[KS_ALPHA] = 1.0;
    ks[KS_OMEGA] = 1.0;
    ks[KS_STATE] = sqrt(d0) <= rtol * sqrt(ks[KS_BNORM2]) ? 1.0 : 0.0;
    cP[0] = 1.0, cP[1] = 0.0, cP[2] = 0.0;  // p = r
    return;
  }
  if (st == 1.0 || st < 0.0) return;
  if (phase == 1) {  // d0 = rh.v
    if (!isfinite(d0) || d0 == 0.0 || !isfinite(ks[KS_RHO]) || fabs(ks[KS_RHO]) < 1e-300 * ks[KS_BNORM2]) {
      ks[KS_STATE] = -1.0;
      return;
    }
    const double alpha = ks[KS_RHO] / d0;
    ks[KS_ALPHA] = alpha;
    cS[0] = 1.0, cS[1] = -alpha, cS[2] = 0.0;
  } else if (phase == 2) {  // d0 = s.s
    ks[KS_ITERS] += 1.0;
    if (sqrt(d0) <= rtol * sqrt(ks[KS_BNORM2])) ks[KS_STATE] = 2.0;
  } else if (phase == 3) {  // d0 = t.s, d1 = t.t
    double omega = 0.0;
    if (st != 2.0) {
      if (!isfinite(d1) || d1 == 0.0) {
        ks[KS_STATE] = -2.0;
        return;
      }
      omega = d0 / d1;
    }
    ks[KS_OMEGA] = omega;
    cX[0] = 1.0, cX[1] = ks[KS_ALPHA], cX[2] = omega;
    cR[0] = 1.0, cR[1] = -omega, cR[2] = 0.0;
  } else if (phase == 4) {  // d0 = r.r, d1 = rh.r
    ks[KS_RNORM2] = d0;
    if (st == 2.0 || sqrt(d0) <= rtol * sqrt(ks[KS_BNORM2])) {
      ks[KS_STATE] = 1.0;
      return;
    }
    if (ks[KS_OMEGA] == 0.0) {
      ks[KS_STATE] = -3.0;
      return;
    }
    ks[KS_RHO_OLD] = ks[KS_RHO];
    ks[KS_RHO] = d1;
    const double beta = (d1 / ks[KS_RHO_OLD]) * (ks[KS_ALPHA] / ks[KS_OMEGA]);
    cP[0] = 1.0, cP[1] = beta, cP[2] = -beta * ks[KS_OMEGA];
  }
}

// GMRES(m): h[i] = V_i . w for i < nv (one block per (vector, chunk); partial[i * gx + blockIdx.x])
// (multi-GPU: rowkind != nullptr restricts the sums to the rows this rank accounts for, as fc_dots2)
__global__ __launch_bounds__(256) void fc_multidot(int n, int nv, const double* __restrict__ V, const double* __restrict__ w,
                                                   double* __restrict__ partial, const double* __restrict__ ks,
                                                   const unsigned char* __restrict__ rowkind = nullptr, int lead = 1) {
  if (ks[KS_STATE] != 0.0) return;
  const int i = blockIdx.y;
  const double* __restrict__ v = V + (size_t)i * n;
  double s = 0.0;
  if (rowkind) {
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
      const int kind = rowkind[k];
      if (kind == 1 || (kind == 2 && lead)) s += v[k] * w[k];
    }
  } else {
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) s += v[k] * w[k];
  }
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(size_t)i * gridDim.x + blockIdx.x] = red[0];
}
// h[i] (+)= sum of its gx partials (fixed order); one wave per dot
__global__ void fc_multidot_reduce(int nv, int gx, const double* __restrict__ partial, double* h, int accumulate, const double* __restrict__ ks) {
  if (ks[KS_STATE] != 0.0) return;
  const int i = blockIdx.x;
  double s = 0.0;
  for (int k = threadIdx.x; k < gx; k += 64) s += partial[(size_t)i * gx + k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (threadIdx.x == 0 && i < nv) h[i] = accumulate ? h[i] + s : s;
}
// h[i] += h2[i]: the second Gram-Schmidt pass joins the Hessenberg column
__global__ void fc_small_add(int nv, const double* __restrict__ h2, double* __restrict__ h, const double* __restrict__ ks) {
  if (ks[KS_STATE] != 0.0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nv) h[i] += h2[i];
}
// w -= sum_i h[i] V_i
__global__ void fc_gmres_project(int n, int nv, const double* __restrict__ V, const double* __restrict__ h, double* w,
                                 const double* __restrict__ ks) {
  if (ks[KS_STATE] != 0.0) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  double s = w[k];
  for (int i = 0; i < nv; ++i) s -= h[i] * V[(size_t)i * n + k];
  w[k] = s;
}
// out = scale_num / sqrt(*norm2) * in   (v_{j+1} = w / |w|; norm2 on the device)
__global__ void fc_scale_by_norm(int n, const double* __restrict__ in, const double* __restrict__ norm2, double* out,
                                 const double* __restrict__ ks) {
  if (ks[KS_STATE] != 0.0) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = in[k] / sqrt(norm2[0]);
}
// out = sum_i y[i] V_i  (the correction in the Krylov basis)
__global__ void fc_gmres_combine(int n, int nv, const double* __restrict__ V, const double* __restrict__ y, double* out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  double s = 0.0;
  for (int i = 0; i < nv; ++i) s += y[i] * V[(size_t)i * n + k];
  out[k] = s;
}
// One thread: column j of the Hessenberg matrix (h[0..j] from the projections, h[j+1] = |w| from norm2) through the
// stored Givens rotations, new rotation, residual estimate |g[j+1]|; when converged or at j + 1 == m: back substitution
// y = R^-1 g.  gm layout: H (m+1) x m column-major | cs[m] | sn[m] | g[m+1] | y[m] | hcol[m+2] | norm2 | used
// (hcol2 != nullptr: the second Gram-Schmidt pass's coefficients are added to the column here instead of by a launch of their own;
//  npart != nullptr: |w|^2 is folded here from the gx partial sums of fc_multidot -- 64 threads, the order of fc_multidot_reduce --
//  instead of by a launch of its own: two launches less per Arnoldi step on single-GPU handles)
__global__ void fc_gmres_givens(int j, int m, double* gm, double* ks, double rtol, const double* __restrict__ hcol2 = nullptr,
                                const double* __restrict__ npart = nullptr, int gx = 0) {
  if (blockIdx.x != 0) return;
  double nsum = 0.0;
  if (npart) {
    if (ks[KS_STATE] == 0.0)
      for (int k = threadIdx.x; k < gx; k += 64) nsum += npart[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nsum += __shfl_down(nsum, off, 64);
  }
  if (threadIdx.x != 0) return;
  double* H = gm;
  double* cs = H + (size_t)(m + 1) * m;
  double* sn = cs + m;
  double* g = sn + m;
  double* y = g + m + 1;
  double* hcol = y + m;
  double* norm2 = hcol + m + 2;
  double* used = norm2 + 1;
  if (ks[KS_STATE] != 0.0) return;
  if (npart) norm2[0] = nsum;
  double* Hj = H + (size_t)j * (m + 1);
  for (int i = 0; i <= j; ++i) Hj[i] = hcol2 ? hcol[i] + hcol2[i] : hcol[i];
  Hj[j + 1] = sqrt(norm2[0]);
  for (int i = 0; i < j; ++i) {
    const double a = cs[i] * Hj[i] + sn[i] * Hj[i + 1], b = -sn[i] * Hj[i] + cs[i] * Hj[i + 1];
    Hj[i] = a;
    Hj[i + 1] = b;
  }
  const double a = Hj[j], b = Hj[j + 1], r = hypot(a, b);
  if (!(r > 0.0) || !isfinite(r)) {
    ks[KS_STATE] = -4.0;
    return;
  }
  cs[j] = a / r;
  sn[j] = b / r;
  Hj[j] = r;
  Hj[j + 1] = 0.0;
  g[j + 1] = -sn[j] * g[j];
  g[j] = cs[j] * g[j];
  ks[KS_ITERS] += 1.0;
  ks[KS_RNORM2] = g[j + 1] * g[j + 1];
  used[0] = (double)(j + 1);
  const bool conv = fabs(g[j + 1]) <= rtol * sqrt(ks[KS_BNORM2]);
  if (conv || j + 1 == m) {
    for (int i = j; i >= 0; --i) {
      double s = g[i];
      for (int k = i + 1; k <= j; ++k) s -= H[(size_t)k * (m + 1) + i] * y[k];
      y[i] = s / H[(size_t)i * (m + 1) + i];
    }
    ks[KS_STATE] = conv ? 3.0 : 4.0;  // 3: converged, 4: restart — both: the cycle's correction must be applied
  }
}
// start of a GMRES cycle: norm2[0] = |r|^2 (from D0), g = (|r|, 0, ...), state -> running or converged
__global__ void fc_gmres_begin(int m, double* gm, double* ks, double rtol, int first) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double* g = gm + (size_t)(m + 1) * m + 2 * m;
  double* norm2 = g + (m + 1) + m + (m + 2);
  const double r2 = ks[KS_D0];
  if (first) {
    ks[KS_BNORM2] = ks[KS_D1];
    ks[KS_ITERS] = 0.0;
  }
  ks[KS_RNORM2] = r2;
  norm2[0] = r2;
  for (int i = 0; i <= m; ++i) g[i] = 0.0;
  g[0] = sqrt(r2);
  ks[KS_STATE] = (sqrt(r2) <= rtol * sqrt(ks[KS_BNORM2]) || !(ks[KS_BNORM2] > 0.0)) ? 1.0 : 0.0;
}

// truncated factors: x = dscale * y on the rows whose pivot blocks are not kept (diagonal stand-in for their Schur complement)
__global__ void fc_diag_stage(int n, const double* __restrict__ dscale, const double* __restrict__ y, double* __restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = dscale[i] * y[i];
}
