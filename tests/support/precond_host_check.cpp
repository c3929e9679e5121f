// Host-logic check of csrc/fc_precond.hpp (no GPU): smoothed-aggregation hierarchy of a 2-D Poisson matrix, V-cycle convergence, and the
// folded transfer operators (fold_down / fold_up, fold_v22) against the plain V(1,1) / V(2,2) cycles they replace.  Built and run by
// tests/test_precond_host.py; prints "key value" lines.
#include <cstdio>
#include <functional>

#include "../../flowcontrol_amd/csrc/fc_precond.hpp"

using fcpc::Csr;

static std::vector<double> spmv(const Csr& M, const std::vector<double>& v) {
  std::vector<double> y((size_t)M.nrows, 0.0);
  for (int i = 0; i < M.nrows; ++i) {
    double s = 0.0;
    for (int k = M.rp[(size_t)i]; k < M.rp[(size_t)i + 1]; ++k) s += M.v[(size_t)k] * v[(size_t)M.ci[(size_t)k]];
    y[(size_t)i] = s;
  }
  return y;
}

int main() {
  const int n = 48, N = n * n;  // 5-point Laplacian with Dirichlet boundary, row by row
  Csr A;
  A.nrows = A.ncols = N;
  A.rp.assign((size_t)N + 1, 0);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      const int r = i * n + j;
      auto put = [&](int c, double v) { A.ci.push_back(c), A.v.push_back(v); };
      if (i > 0) put(r - n, -1.0);
      if (j > 0) put(r - 1, -1.0);
      put(r, 4.0 + 0.01 * ((i * 7 + j * 3) % 5));
      if (j < n - 1) put(r + 1, -1.0);
      if (i < n - 1) put(r + n, -1.0);
      A.rp[(size_t)r + 1] = (int)A.ci.size();
    }
  const fcpc::Amg H = fcpc::build_amg(A);
  std::printf("levels %zu\ncoarse %d\n", H.levels.size(), H.n_coarse);
  const int L = (int)H.levels.size();
  auto coarse = [&](const std::vector<double>& r) {
    const int m = H.n_coarse;
    std::vector<double> y((size_t)m, 0.0);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) y[(size_t)i] += H.coarse_inv[(size_t)i * m + j] * r[(size_t)j];
    return y;
  };
  std::function<std::vector<double>(int, const std::vector<double>&, int)> plain = [&](int l, const std::vector<double>& r, int sweeps) {
    if (l == L) return coarse(r);
    const fcpc::Level& V = H.levels[(size_t)l];
    const int m = V.A.nrows;
    std::vector<double> x((size_t)m, 0.0);
    for (int s = 0; s < sweeps; ++s) {
      const std::vector<double> Ax = spmv(V.A, x);
      for (int i = 0; i < m; ++i) x[(size_t)i] += V.wdinv[(size_t)i] * (r[(size_t)i] - Ax[(size_t)i]);
    }
    std::vector<double> t = spmv(V.A, x);
    for (int i = 0; i < m; ++i) t[(size_t)i] = r[(size_t)i] - t[(size_t)i];
    const std::vector<double> xc = plain(l + 1, spmv(V.R, t), sweeps), Px = spmv(V.P, xc);
    for (int i = 0; i < m; ++i) x[(size_t)i] += Px[(size_t)i];
    for (int s = 0; s < sweeps; ++s) {
      const std::vector<double> Ax = spmv(V.A, x);
      for (int i = 0; i < m; ++i) x[(size_t)i] += V.wdinv[(size_t)i] * (r[(size_t)i] - Ax[(size_t)i]);
    }
    return x;
  };
  std::vector<Csr> G1((size_t)L), U1((size_t)L), G2((size_t)L), U2((size_t)L);
  for (int l = 0; l < L; ++l) {
    G1[(size_t)l] = fcpc::fold_down(H.levels[(size_t)l]);
    U1[(size_t)l] = fcpc::fold_up(H.levels[(size_t)l]);
    fcpc::fold_v22(H.levels[(size_t)l], G2[(size_t)l], U2[(size_t)l]);
  }
  std::function<std::vector<double>(int, const std::vector<double>&, const std::vector<Csr>&, const std::vector<Csr>&)> folded =
      [&](int l, const std::vector<double>& r, const std::vector<Csr>& G, const std::vector<Csr>& U) {
        if (l == L) return coarse(r);
        std::vector<double> cat(r);
        const std::vector<double> zc = folded(l + 1, spmv(G[(size_t)l], r), G, U);
        cat.insert(cat.end(), zc.begin(), zc.end());
        return spmv(U[(size_t)l], cat);
      };
  std::vector<double> r((size_t)N);
  for (int i = 0; i < N; ++i) r[(size_t)i] = std::sin(0.37 * i + 0.2);
  auto rel = [&](const std::vector<double>& a, const std::vector<double>& b) {
    double d = 0.0, nb = 0.0;
    for (size_t i = 0; i < a.size(); ++i) d += (a[i] - b[i]) * (a[i] - b[i]), nb += b[i] * b[i];
    return std::sqrt(d / nb);
  };
  std::printf("fold11 %.3e\nfold22 %.3e\n", rel(folded(0, r, G1, U1), plain(0, r, 1)), rel(folded(0, r, G2, U2), plain(0, r, 2)));
  for (int sweeps = 1; sweeps <= 2; ++sweeps) {
    std::vector<double> x((size_t)N, 0.0);
    double res = 1.0, prev = 1.0, rate = 0.0;
    for (int it = 0; it < 8; ++it) {
      std::vector<double> t = spmv(A, x);
      double nr = 0.0, nb = 0.0;
      for (int i = 0; i < N; ++i) t[(size_t)i] = r[(size_t)i] - t[(size_t)i], nr += t[(size_t)i] * t[(size_t)i], nb += r[(size_t)i] * r[(size_t)i];
      prev = res, res = std::sqrt(nr / nb);
      if (it > 0) rate = res / prev;
      const std::vector<double> dx = sweeps == 1 ? folded(0, t, G1, U1) : folded(0, t, G2, U2);
      for (int i = 0; i < N; ++i) x[(size_t)i] += dx[(size_t)i];
    }
    std::printf("residual%d %.3e\nrate%d %.3f\n", sweeps, res, sweeps, rate);
  }
  // blocks of a tiny saddle-point matrix: [F Bt; B 0] with N_vel = 4, one pressure, identity permutation reversed
  {
    const int Nv = 4, Nt = 5;
    std::vector<int> rp = {0, 3, 6, 9, 12, 16}, col = {0, 1, 4, 0, 1, 4, 2, 3, 4, 2, 3, 4, 0, 1, 2, 3};
    std::vector<double> val = {4, 1, 0.5, 1, 3, -0.5, 5, 0, 0.25, 2, 6, -0.25, 0.5, -0.5, 0.25, -0.25};
    std::vector<int> perm = {4, 3, 2, 1, 0};
    const fcpc::Blocks X = fcpc::split_blocks(Nt, Nv, rp, col, val.data(), perm);
    std::printf("blocks %d %d %lld %lld %lld\n", X.nu, X.np, (long long)X.F.nnz(), (long long)X.B.nnz(), (long long)X.Bt.nnz());
    std::vector<double> dinv((size_t)X.nu);
    for (int i = 0; i < X.nu; ++i) dinv[(size_t)i] = 1.0 / X.dF[(size_t)i];
    const Csr S = fcpc::spgemm(X.B, X.Bt, dinv.data());
    std::printf("schur %.12f\n", S.v[0]);  // 0.5*0.5/4 + 0.5*0.5/3 + 0.25*0.25/5 + 0.25*0.25/6
  }
  return 0;
}
