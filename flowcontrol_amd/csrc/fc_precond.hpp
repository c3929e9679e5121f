// fc_precond.hpp -- host side of the FACTORISATION-FREE preconditioner of the device Krylov methods (fc_setup_krylov):
// a SIMPLE-type block preconditioner of the Taylor-Hood saddle-point operator
//
//        A = [ F  Bt ]      M^-1 r:   u1 = J_k(F) r_u                 k damped-Jacobi sweeps on the velocity block
//            [ B  0  ]                zp = -S^-1 (r_p - B u1)         S = B diag(F)^-1 Bt, one AMG V-cycle
//                                     zu = u1 - diag(F)^-1 Bt zp
//
// (the reference's plug-in point is FlowSolver._make_solver, src/flowcontrol/flowsolver.py:812-814; BASELINE.json north_star:
// "HIP BiCGStab/GMRES with CSR SpMV and block-Jacobi/ILU(0) preconditioning").  The time-step operators are mass dominated
// (3 / (2 dt) M + convection + nu K), so the velocity block is well served by Jacobi sweeps; what needs a global solve is the
// pressure Schur complement, a Poisson-like SPD matrix on the P1 vertices -- here a smoothed-aggregation algebraic multigrid
// hierarchy built from the assembled values alone (no mesh knowledge, no boundary-condition guesswork: S is formed
// algebraically from the eliminated operator).  Nothing here factorises anything but the coarsest AMG level (<= 256 dofs,
// dense inverse).  Everything in this file is index / setup work on the host; the applies are device kernels
// (fc_pc_csr / fc_pc_dense / fc_pc_final in fc_precond.hip.h).
//
// Measured on the cylinder O1 BDF2 operator (56 203 dofs), GMRES to 1e-10 from a zero guess (tests/support/precond_study.py,
// profiles/r05_precond_study.txt): additive Vanka patches alone need ~750 iterations, block-triangular with one Jacobi sweep 47,
// SIMPLE with one sweep 33 / two sweeps 20 (exact S^-1), SIMPLE with two sweeps and one V-cycle 31; on the device, inside time
// steps (warm start) 18.6 per step.
#pragma once
#include <algorithm>
#include <cmath>
#include <climits>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace fcpc {

struct Csr {
  int nrows = 0, ncols = 0;
  std::vector<int> rp, ci;
  std::vector<double> v;
  int64_t nnz() const { return (int64_t)ci.size(); }
};

inline Csr transpose(const Csr& A) {
  Csr T;
  T.nrows = A.ncols, T.ncols = A.nrows;
  T.rp.assign((size_t)T.nrows + 1, 0);
  for (int c : A.ci) T.rp[(size_t)c + 1]++;
  for (int r = 0; r < T.nrows; ++r) T.rp[(size_t)r + 1] += T.rp[(size_t)r];
  T.ci.resize(A.ci.size()), T.v.resize(A.v.size());
  std::vector<int> fill(T.rp.begin(), T.rp.end() - 1);
  for (int r = 0; r < A.nrows; ++r)
    for (int k = A.rp[(size_t)r]; k < A.rp[(size_t)r + 1]; ++k) {
      const int q = fill[(size_t)A.ci[(size_t)k]]++;
      T.ci[(size_t)q] = r, T.v[(size_t)q] = A.v[(size_t)k];
    }
  return T;  // rows of A ascending => columns of every row of T ascending
}

// C = A * diag(s) * B (s may be null); the columns of every row come out sorted (fixed summation order: bit-reproducible)
inline Csr spgemm(const Csr& A, const Csr& B, const double* s = nullptr) {
  if (A.ncols != B.nrows) throw std::runtime_error("fcpc::spgemm: shapes disagree");
  Csr C;
  C.nrows = A.nrows, C.ncols = B.ncols;
  C.rp.assign((size_t)C.nrows + 1, 0);
  std::vector<double> acc((size_t)B.ncols, 0.0);
  std::vector<int> mark((size_t)B.ncols, -1), cols;
  for (int i = 0; i < A.nrows; ++i) {
    cols.clear();
    for (int k = A.rp[(size_t)i]; k < A.rp[(size_t)i + 1]; ++k) {
      const int j = A.ci[(size_t)k];
      const double a = A.v[(size_t)k] * (s ? s[j] : 1.0);
      for (int q = B.rp[(size_t)j]; q < B.rp[(size_t)j + 1]; ++q) {
        const int c = B.ci[(size_t)q];
        if (mark[(size_t)c] != i) mark[(size_t)c] = i, acc[(size_t)c] = 0.0, cols.push_back(c);
        acc[(size_t)c] += a * B.v[(size_t)q];
      }
    }
    std::sort(cols.begin(), cols.end());
    for (int c : cols) C.ci.push_back(c), C.v.push_back(acc[(size_t)c]);
    C.rp[(size_t)i + 1] = (int)C.ci.size();
  }
  return C;
}

inline std::vector<double> diagonal(const Csr& A) {
  std::vector<double> d((size_t)A.nrows, 0.0);
  for (int i = 0; i < A.nrows; ++i)
    for (int k = A.rp[(size_t)i]; k < A.rp[(size_t)i + 1]; ++k)
      if (A.ci[(size_t)k] == i) d[(size_t)i] = A.v[(size_t)k];
  return d;
}

// spectral radius of D^-1 A by power iteration from a fixed start vector (deterministic); a lower bound that is within a few
// per cent after 40 steps on the matrices met here (Jacobi damping derived from it keeps a 1.5 x safety margin)
inline double rho_dinv(const Csr& A, const std::vector<double>& d, int iters = 40) {
  const int n = A.nrows;
  if (n == 0) return 1.0;
  std::vector<double> x((size_t)n), y((size_t)n);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < n; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    x[(size_t)i] = 0.5 + (double)((s >> 33) & 0xFFFF) / 65536.0;
  }
  double rho = 1.0;
  for (int it = 0; it < iters; ++it) {
    double ny = 0.0, nx = 0.0;
    for (int i = 0; i < n; ++i) {
      double a = 0.0;
      for (int k = A.rp[(size_t)i]; k < A.rp[(size_t)i + 1]; ++k) a += A.v[(size_t)k] * x[(size_t)A.ci[(size_t)k]];
      y[(size_t)i] = a / d[(size_t)i];
      ny += y[(size_t)i] * y[(size_t)i], nx += x[(size_t)i] * x[(size_t)i];
    }
    if (!(ny > 0.0) || !(nx > 0.0)) return rho;
    rho = std::sqrt(ny / nx);
    const double inv = 1.0 / std::sqrt(ny);
    for (int i = 0; i < n; ++i) x[(size_t)i] = y[(size_t)i] * inv;
  }
  return rho;
}

struct Level {
  Csr A, P, R;
  std::vector<double> wdinv;  // omega / diag(A): the damped-Jacobi smoother
  double rho = 1.0;
};

struct Amg {
  std::vector<Level> levels;      // sparse levels, finest first
  int n_coarse = 0;
  std::vector<double> coarse_inv;  // dense inverse of the coarsest operator, row-major
  int64_t nnz_total = 0;
};

// greedy aggregation on the strength graph |a_ij| >= theta sqrt(|a_ii a_jj|) (three passes: root + all its strong neighbours
// while none of them is taken; leftovers join a neighbouring aggregate of pass 1; what is still left becomes singletons)
inline int aggregate(const Csr& A, double theta, std::vector<int>& agg) {
  const int n = A.nrows;
  const std::vector<double> d = diagonal(A);
  std::vector<int> sp((size_t)n + 1, 0), si;
  for (int i = 0; i < n; ++i) {
    for (int k = A.rp[(size_t)i]; k < A.rp[(size_t)i + 1]; ++k) {
      const int j = A.ci[(size_t)k];
      if (j != i && std::fabs(A.v[(size_t)k]) >= theta * std::sqrt(std::fabs(d[(size_t)i] * d[(size_t)j]))) si.push_back(j);
    }
    sp[(size_t)i + 1] = (int)si.size();
  }
  agg.assign((size_t)n, -1);
  int na = 0;
  for (int i = 0; i < n; ++i) {
    if (agg[(size_t)i] >= 0) continue;
    bool free_nb = true;
    for (int k = sp[(size_t)i]; k < sp[(size_t)i + 1] && free_nb; ++k) free_nb = agg[(size_t)si[(size_t)k]] < 0;
    if (!free_nb) continue;
    agg[(size_t)i] = na;
    for (int k = sp[(size_t)i]; k < sp[(size_t)i + 1]; ++k) agg[(size_t)si[(size_t)k]] = na;
    ++na;
  }
  const std::vector<int> pass1 = agg;
  for (int i = 0; i < n; ++i) {
    if (pass1[(size_t)i] >= 0) continue;
    for (int k = sp[(size_t)i]; k < sp[(size_t)i + 1]; ++k)
      if (pass1[(size_t)si[(size_t)k]] >= 0) {
        agg[(size_t)i] = pass1[(size_t)si[(size_t)k]];
        break;
      }
  }
  for (int i = 0; i < n; ++i)
    if (agg[(size_t)i] < 0) agg[(size_t)i] = na++;
  return na;
}

// dense inverse (row-major) by Gauss-Jordan with partial pivoting; throws on a singular matrix
inline std::vector<double> dense_inverse(const Csr& A) {
  const int n = A.nrows;
  std::vector<double> M((size_t)n * n, 0.0), I((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) {
    I[(size_t)i * n + i] = 1.0;
    for (int k = A.rp[(size_t)i]; k < A.rp[(size_t)i + 1]; ++k) M[(size_t)i * n + A.ci[(size_t)k]] = A.v[(size_t)k];
  }
  for (int c = 0; c < n; ++c) {
    int p = c;
    for (int r = c + 1; r < n; ++r)
      if (std::fabs(M[(size_t)r * n + c]) > std::fabs(M[(size_t)p * n + c])) p = r;
    if (!(std::fabs(M[(size_t)p * n + c]) > 0.0)) throw std::runtime_error("fcpc: the coarsest AMG operator is singular (enclosed flow without a pressure pin?)");
    if (p != c)
      for (int j = 0; j < n; ++j) std::swap(M[(size_t)p * n + j], M[(size_t)c * n + j]), std::swap(I[(size_t)p * n + j], I[(size_t)c * n + j]);
    const double inv = 1.0 / M[(size_t)c * n + c];
    for (int j = 0; j < n; ++j) M[(size_t)c * n + j] *= inv, I[(size_t)c * n + j] *= inv;
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = M[(size_t)r * n + c];
      if (f == 0.0) continue;
      for (int j = 0; j < n; ++j) M[(size_t)r * n + j] -= f * M[(size_t)c * n + j], I[(size_t)r * n + j] -= f * I[(size_t)c * n + j];
    }
  }
  return I;
}

// smoothed-aggregation hierarchy of an SPD-like matrix (Vanek, Mandel, Brezina 1996): tentative prolongator = normalised
// aggregate indicator, one damped-Jacobi smoothing step P = (I - 4 / (3 rho) D^-1 A) T, Galerkin coarse operators R A P
inline Amg build_amg(Csr A0, double theta = 0.08, int coarse_max = 256, int max_levels = 12) {
  Amg H;
  Csr A = std::move(A0);
  while (A.nrows > coarse_max && (int)H.levels.size() < max_levels) {
    const int n = A.nrows;
    std::vector<int> agg;
    const int na = aggregate(A, theta, agg);
    if (na >= n) break;  // no coarsening possible (diagonal matrix)
    std::vector<int> cnt((size_t)na, 0);
    for (int i = 0; i < n; ++i) cnt[(size_t)agg[(size_t)i]]++;
    Csr T;
    T.nrows = n, T.ncols = na;
    T.rp.resize((size_t)n + 1), T.ci.resize((size_t)n), T.v.resize((size_t)n);
    for (int i = 0; i < n; ++i) T.rp[(size_t)i] = i, T.ci[(size_t)i] = agg[(size_t)i], T.v[(size_t)i] = 1.0 / std::sqrt((double)cnt[(size_t)agg[(size_t)i]]);
    T.rp[(size_t)n] = n;
    Level L;
    const std::vector<double> d = diagonal(A);
    for (double v : d)
      if (!(std::fabs(v) > 0.0)) throw std::runtime_error("fcpc: zero diagonal in an AMG operator");
    L.rho = rho_dinv(A, d);
    const double w = 4.0 / (3.0 * L.rho);
    L.wdinv.resize((size_t)n);
    std::vector<double> wd((size_t)n);
    for (int i = 0; i < n; ++i) L.wdinv[(size_t)i] = w / d[(size_t)i], wd[(size_t)i] = -w / d[(size_t)i];
    // P = T - w D^-1 (A T)
    Csr AT = spgemm(A, T);
    Csr P;
    P.nrows = n, P.ncols = na;
    P.rp.assign((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) {
      bool had = false;
      const int a = agg[(size_t)i];
      const double t = T.v[(size_t)i];
      for (int k = AT.rp[(size_t)i]; k < AT.rp[(size_t)i + 1]; ++k) {
        const int c = AT.ci[(size_t)k];
        if (!had && c > a) P.ci.push_back(a), P.v.push_back(t), had = true;
        double val = wd[(size_t)i] * AT.v[(size_t)k];
        if (c == a) val += t, had = true;
        P.ci.push_back(c), P.v.push_back(val);
      }
      if (!had) P.ci.push_back(a), P.v.push_back(t);
      P.rp[(size_t)i + 1] = (int)P.ci.size();
    }
    L.R = transpose(P);
    Csr AP = spgemm(A, P);
    Csr Ac = spgemm(L.R, AP);
    L.P = std::move(P);
    L.A = std::move(A);
    H.nnz_total += L.A.nnz() + L.P.nnz() + L.R.nnz();
    H.levels.push_back(std::move(L));
    A = std::move(Ac);
  }
  H.n_coarse = A.nrows;
  if (A.nrows > 4096) throw std::runtime_error("fcpc: AMG coarsening stalled above 4096 dofs");
  H.coarse_inv = dense_inverse(A);
  H.nnz_total += (int64_t)A.nrows * A.nrows;
  return H;
}

// One V(1,1)-cycle needs only TWO sparse products per level when the damped-Jacobi smoothers are folded into the transfer
// operators on the host (every launch of the apply is latency-bound, so launches are what counts):
//   down:  r_c = R (r - A Wd r)                           = G r,          G = R (I - A Wd)
//   up:    z   = x2 + Wd (r - A x2),  x2 = Wd r + P z_c   = K r + Q z_c,  K = Wd (2 I - A Wd),  Q = (I - Wd A) P
// U = [K | Q] acts on the concatenated vector [r ; z_c] (the device keeps the two halves next to each other).
inline Csr fold_down(const Level& L) {
  Csr M = L.A;  // I - A Wd
  for (int i = 0; i < M.nrows; ++i)
    for (int k = M.rp[(size_t)i]; k < M.rp[(size_t)i + 1]; ++k) {
      const int j = M.ci[(size_t)k];
      M.v[(size_t)k] = (i == j ? 1.0 : 0.0) - M.v[(size_t)k] * L.wdinv[(size_t)j];
    }
  return spgemm(L.R, M);
}
inline Csr fold_up(const Level& L) {
  const int n = L.A.nrows, nc = L.P.ncols;
  const Csr AP = spgemm(L.A, L.P);
  Csr U;
  U.nrows = n, U.ncols = n + nc;
  U.rp.assign((size_t)n + 1, 0);
  for (int i = 0; i < n; ++i) {
    const double w = L.wdinv[(size_t)i];
    for (int k = L.A.rp[(size_t)i]; k < L.A.rp[(size_t)i + 1]; ++k) {  // K = Wd (2 I - A Wd)
      const int j = L.A.ci[(size_t)k];
      U.ci.push_back(j), U.v.push_back(w * ((i == j ? 2.0 : 0.0) - L.A.v[(size_t)k] * L.wdinv[(size_t)j]));
    }
    // Q = P - Wd (A P): the pattern of A P contains the pattern of P (the diagonal of A is not zero)
    int q = L.P.rp[(size_t)i];
    const int q1 = L.P.rp[(size_t)i + 1];
    for (int k = AP.rp[(size_t)i]; k < AP.rp[(size_t)i + 1]; ++k) {
      const int c = AP.ci[(size_t)k];
      double val = -w * AP.v[(size_t)k];
      while (q < q1 && L.P.ci[(size_t)q] < c) U.ci.push_back(n + L.P.ci[(size_t)q]), U.v.push_back(L.P.v[(size_t)q]), ++q;
      if (q < q1 && L.P.ci[(size_t)q] == c) val += L.P.v[(size_t)q], ++q;
      U.ci.push_back(n + c), U.v.push_back(val);
    }
    for (; q < q1; ++q) U.ci.push_back(n + L.P.ci[(size_t)q]), U.v.push_back(L.P.v[(size_t)q]);
    U.rp[(size_t)i + 1] = (int)U.ci.size();
  }
  return U;
}
// C = alpha A + beta B (same shape; columns of every row sorted)
inline Csr add(const Csr& A, double alpha, const Csr& B, double beta) {
  if (A.nrows != B.nrows || A.ncols != B.ncols) throw std::runtime_error("fcpc::add: shapes disagree");
  Csr C;
  C.nrows = A.nrows, C.ncols = A.ncols;
  C.rp.assign((size_t)A.nrows + 1, 0);
  for (int i = 0; i < A.nrows; ++i) {
    int p = A.rp[(size_t)i], q = B.rp[(size_t)i];
    const int p1 = A.rp[(size_t)i + 1], q1 = B.rp[(size_t)i + 1];
    while (p < p1 || q < q1) {
      const int ca = p < p1 ? A.ci[(size_t)p] : INT32_MAX, cb = q < q1 ? B.ci[(size_t)q] : INT32_MAX;
      double v = 0.0;
      const int c = ca < cb ? ca : cb;
      if (ca == c) v += alpha * A.v[(size_t)p++];
      if (cb == c) v += beta * B.v[(size_t)q++];
      C.ci.push_back(c), C.v.push_back(v);
    }
    C.rp[(size_t)i + 1] = (int)C.ci.size();
  }
  return C;
}
inline Csr identity(int n) {
  Csr I;
  I.nrows = I.ncols = n;
  I.rp.resize((size_t)n + 1), I.ci.resize((size_t)n), I.v.assign((size_t)n, 1.0);
  for (int i = 0; i < n; ++i) I.rp[(size_t)i] = i, I.ci[(size_t)i] = i;
  I.rp[(size_t)n] = n;
  return I;
}
inline Csr scale_rows(Csr A, const std::vector<double>& d) {
  for (int i = 0; i < A.nrows; ++i)
    for (int k = A.rp[(size_t)i]; k < A.rp[(size_t)i + 1]; ++k) A.v[(size_t)k] *= d[(size_t)i];
  return A;
}
// The same folding for a V(2,2)-cycle (two damped-Jacobi sweeps before and after the coarse correction) -- still two products per
// level, only denser ones (patterns of A^2 / A^3 instead of A): with E = I - Wd A (one sweep's error propagator), W2 = (I + E) Wd,
//   down:  r_c = R (I - A W2) r                   up:  z = (I + E^2) W2 r + E^2 P z_c
inline void fold_v22(const Level& L, Csr& G, Csr& U) {
  const int n = L.A.nrows, nc = L.P.ncols;
  const Csr I = identity(n);
  const Csr E = add(I, 1.0, scale_rows(L.A, L.wdinv), -1.0);          // I - Wd A
  Csr Wd = I;
  Wd.v = L.wdinv;
  const Csr W2 = spgemm(add(I, 1.0, E, 1.0), Wd);                      // (I + E) Wd
  G = spgemm(L.R, add(I, 1.0, spgemm(L.A, W2), -1.0));                 // R (I - A W2)
  const Csr E2 = spgemm(E, E);
  const Csr K = spgemm(add(I, 1.0, E2, 1.0), W2);                      // (I + E^2) W2
  const Csr Q = spgemm(E2, L.P);                                       // E^2 P
  U.nrows = n, U.ncols = n + nc;
  U.rp.assign((size_t)n + 1, 0);
  U.ci.clear(), U.v.clear();
  for (int i = 0; i < n; ++i) {
    for (int k = K.rp[(size_t)i]; k < K.rp[(size_t)i + 1]; ++k) U.ci.push_back(K.ci[(size_t)k]), U.v.push_back(K.v[(size_t)k]);
    for (int k = Q.rp[(size_t)i]; k < Q.rp[(size_t)i + 1]; ++k) U.ci.push_back(n + Q.ci[(size_t)k]), U.v.push_back(Q.v[(size_t)k]);
    U.rp[(size_t)i + 1] = (int)U.ci.size();
  }
}

// two damped-Jacobi sweeps on F u = r from u = 0 as ONE product: u = Wd (2 I - F Wd) r
inline Csr fold_jacobi2(const Csr& F, const std::vector<double>& wd) {
  Csr K = F;
  for (int i = 0; i < K.nrows; ++i)
    for (int k = K.rp[(size_t)i]; k < K.rp[(size_t)i + 1]; ++k) {
      const int j = K.ci[(size_t)k];
      K.v[(size_t)k] = wd[(size_t)i] * ((i == j ? 2.0 : 0.0) - K.v[(size_t)k] * wd[(size_t)j]);
    }
  return K;
}

// The blocks of the saddle-point operator in COMPACT numberings (velocity dofs / pressure dofs, each in the order of their
// permuted positions), from the handle's CSR in W numbering.  Exact zeros (the rows / columns the symmetric Dirichlet
// elimination emptied) are dropped.
struct Blocks {
  int nu = 0, np = 0;
  std::vector<int> vpos, ppos;  // compact index -> position in the permuted vector
  Csr F, B, Bt;
  std::vector<double> dF;  // diag(F)
};

inline Blocks split_blocks(int N, int n_vel, const std::vector<int>& rowptr, const std::vector<int>& col, const double* val,
                           const std::vector<int>& perm /* permuted row -> W dof */) {
  Blocks X;
  std::vector<int> cidx((size_t)N, -1);  // W dof -> compact index inside its block
  for (int i = 0; i < N; ++i) {
    const int w = perm[(size_t)i];
    if (w < n_vel) cidx[(size_t)w] = (int)X.vpos.size(), X.vpos.push_back(i);
    else cidx[(size_t)w] = (int)X.ppos.size(), X.ppos.push_back(i);
  }
  X.nu = (int)X.vpos.size(), X.np = (int)X.ppos.size();
  X.F.nrows = X.F.ncols = X.nu;
  X.Bt.nrows = X.nu, X.Bt.ncols = X.np;
  X.B.nrows = X.np, X.B.ncols = X.nu;
  X.F.rp.assign((size_t)X.nu + 1, 0), X.Bt.rp.assign((size_t)X.nu + 1, 0), X.B.rp.assign((size_t)X.np + 1, 0);
  X.dF.assign((size_t)X.nu, 0.0);
  std::vector<std::pair<int, double>> ru, rpp;
  auto flush = [](Csr& M, int row, std::vector<std::pair<int, double>>& e) {
    std::sort(e.begin(), e.end(), [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
    for (const auto& p : e) M.ci.push_back(p.first), M.v.push_back(p.second);
    M.rp[(size_t)row + 1] = (int)M.ci.size();
    e.clear();
  };
  for (int k = 0; k < X.nu; ++k) {
    const int w = perm[(size_t)X.vpos[(size_t)k]];
    for (int q = rowptr[(size_t)w]; q < rowptr[(size_t)w + 1]; ++q) {
      const int c = col[(size_t)q];
      const double a = val[q];
      if (c == w) X.dF[(size_t)k] = a;
      if (a == 0.0) continue;
      if (c < n_vel) ru.emplace_back(cidx[(size_t)c], a);
      else rpp.emplace_back(cidx[(size_t)c], a);
    }
    flush(X.F, k, ru), flush(X.Bt, k, rpp);
  }
  for (int k = 0; k < X.np; ++k) {
    const int w = perm[(size_t)X.ppos[(size_t)k]];
    for (int q = rowptr[(size_t)w]; q < rowptr[(size_t)w + 1]; ++q) {
      const int c = col[(size_t)q];
      const double a = val[q];
      if (a == 0.0) continue;
      if (c >= n_vel) throw std::runtime_error("fcpc: pressure-pressure entries in the operator (not a Taylor-Hood saddle-point matrix)");
      ru.emplace_back(cidx[(size_t)c], a);
    }
    flush(X.B, k, ru);
  }
  return X;
}

}  // namespace fcpc
