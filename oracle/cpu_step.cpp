// TEST INFRASTRUCTURE / CPU BASELINE — not part of the product (only tests/, bench.py's cpu_baseline leg and
// __graft_entry__ may build or load this).
//
// Compiled single-thread restatement of what the reference does per time step on the CPU
// (src/flowcontrol/flowsolver.py:721-762): the element loop of dolfin.SystemAssembler.assemble(rhs)
// (flowsolver.py:728; forms nsforms.py:238-305: BDF1 / BDF2 right-hand sides, exact 7-point degree-5 rule on
// affine P2/P1 triangles, body force interpolated in P2), the Dirichlet lifting with re-evaluated actuator values,
// the sensor functionals (sensor.py:96-98,166-197) and the perturbation energy (flowsolver.py:827-829).
// The sparse triangular solves of LUSolver.solve (flowsolver.py:729) stay in SuperLU (scipy, compiled C).
// FFC generates scalar C++ of this shape for the reference; the numpy einsum oracle (ns_oracle.py) is the
// specification this file is checked against (tests/test_cpu_step.py, 1e-12).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {
struct Tab {
  double phi[7][6], dphi[7][6][2], w[7];
  Tab() {
    const double s15 = std::sqrt(15.0);
    const double a1 = (6.0 - s15) / 21.0, a2 = (6.0 + s15) / 21.0;
    const double lam[7][3] = {{1.0 / 3, 1.0 / 3, 1.0 / 3}, {1 - 2 * a1, a1, a1}, {a1, 1 - 2 * a1, a1}, {a1, a1, 1 - 2 * a1},
                              {1 - 2 * a2, a2, a2},        {a2, 1 - 2 * a2, a2}, {a2, a2, 1 - 2 * a2}};
    const double ww[7] = {9.0 / 40, (155 - s15) / 1200, (155 - s15) / 1200, (155 - s15) / 1200,
                          (155 + s15) / 1200, (155 + s15) / 1200, (155 + s15) / 1200};
    const double dl[3][2] = {{-1, -1}, {1, 0}, {0, 1}};
    const int ev[3][2] = {{1, 2}, {2, 0}, {0, 1}};
    for (int q = 0; q < 7; ++q) {
      w[q] = ww[q];
      for (int i = 0; i < 3; ++i) {
        phi[q][i] = lam[q][i] * (2 * lam[q][i] - 1);
        for (int d = 0; d < 2; ++d) dphi[q][i][d] = (4 * lam[q][i] - 1) * dl[i][d];
      }
      for (int k = 0; k < 3; ++k) {
        const int i = ev[k][0], j = ev[k][1];
        phi[q][3 + k] = 4 * lam[q][i] * lam[q][j];
        for (int d = 0; d < 2; ++d) dphi[q][3 + k][d] = 4 * (lam[q][i] * dl[j][d] + lam[q][j] * dl[i][d]);
      }
    }
  }
};
const Tab T;
}  // namespace

extern "C" {

// b (N) = element-loop right-hand side of order 1 / 2 (nsforms.py:258-262 / 291-300), no boundary conditions.
//   g = cm_n u_n + cm_nn u_nn + cc_n (u_n.grad)u_n + cc_nn (u_nn.grad)u_nn + f ;  L_e[a, j] = ∫ g_j φ_a
// cells (nc,3) vertex ids, cell_nodes (nc,6) P2 node ids, coords (nv,2); f may be NULL (P2 nodal force, 2 nn)
void cpu_rhs_elem(int nc, int nn, int N, const double* coords, const int64_t* cells, const int64_t* cell_nodes,
                  const double* un, const double* unn, const double* f, double cm_n, double cm_nn, double cc_n, double cc_nn,
                  double* b) {
  std::memset(b, 0, sizeof(double) * (size_t)N);
  for (int c = 0; c < nc; ++c) {
    const int64_t* v = cells + 3 * (size_t)c;
    const int64_t* nd = cell_nodes + 6 * (size_t)c;
    const double x0 = coords[2 * v[0]], y0 = coords[2 * v[0] + 1];
    const double a = coords[2 * v[1]] - x0, bb = coords[2 * v[2]] - x0;
    const double cc = coords[2 * v[1] + 1] - y0, d = coords[2 * v[2] + 1] - y0;
    const double det = a * d - bb * cc;
    const double j00 = d / det, j01 = -bb / det, j10 = -cc / det, j11 = a / det;
    double ux[6], uy[6], wx[6], wy[6], fx[6], fy[6];
    for (int k = 0; k < 6; ++k) {
      ux[k] = un[nd[k]], uy[k] = un[nn + nd[k]];
      wx[k] = unn ? unn[nd[k]] : 0.0, wy[k] = unn ? unn[nn + nd[k]] : 0.0;
      fx[k] = f ? f[nd[k]] : 0.0, fy[k] = f ? f[nn + nd[k]] : 0.0;
    }
    double Lx[6] = {0, 0, 0, 0, 0, 0}, Ly[6] = {0, 0, 0, 0, 0, 0};
    for (int q = 0; q < 7; ++q) {
      double u = 0, vv = 0, uxi = 0, uet = 0, vxi = 0, vet = 0, w = 0, z = 0, wxi = 0, wet = 0, zxi = 0, zet = 0, gx = 0, gy = 0;
      for (int k = 0; k < 6; ++k) {
        const double ph = T.phi[q][k], dx = T.dphi[q][k][0], de = T.dphi[q][k][1];
        u += ph * ux[k], vv += ph * uy[k], uxi += dx * ux[k], uet += de * ux[k], vxi += dx * uy[k], vet += de * uy[k];
        w += ph * wx[k], z += ph * wy[k], wxi += dx * wx[k], wet += de * wx[k], zxi += dx * wy[k], zet += de * wy[k];
        gx += ph * fx[k], gy += ph * fy[k];
      }
      const double ux_x = uxi * j00 + uet * j10, ux_y = uxi * j01 + uet * j11, uy_x = vxi * j00 + vet * j10, uy_y = vxi * j01 + vet * j11;
      const double wx_x = wxi * j00 + wet * j10, wx_y = wxi * j01 + wet * j11, wy_x = zxi * j00 + zet * j10, wy_y = zxi * j01 + zet * j11;
      gx += cm_n * u + cm_nn * w + cc_n * (u * ux_x + vv * ux_y) + cc_nn * (w * wx_x + z * wx_y);
      gy += cm_n * vv + cm_nn * z + cc_n * (u * uy_x + vv * uy_y) + cc_nn * (w * wy_x + z * wy_y);
      const double wq = T.w[q] * 0.5 * det;
      for (int k = 0; k < 6; ++k) {
        Lx[k] += wq * T.phi[q][k] * gx;
        Ly[k] += wq * T.phi[q][k] * gy;
      }
    }
    for (int k = 0; k < 6; ++k) {
      b[nd[k]] += Lx[k];
      b[nn + nd[k]] += Ly[k];
    }
  }
}

// SystemAssembler's lifting with re-evaluated Dirichlet values:  b -= sum_k u_k lift_k ;  b[dof_i] = sum_k prof[i][k] u_k
void cpu_lift(int N, int n_act, int n_bc, const double* lift /* [n_act][N] */, const int64_t* bc_dofs, const double* prof /* [n_bc][n_act] */,
              const double* uctrl, double* b) {
  for (int k = 0; k < n_act; ++k) {
    const double uk = uctrl[k];
    if (uk == 0.0) continue;
    const double* l = lift + (size_t)k * N;
    for (int i = 0; i < N; ++i) b[i] -= uk * l[i];
  }
  for (int i = 0; i < n_bc; ++i) {
    double g = 0.0;
    for (int k = 0; k < n_act; ++k) g += prof[(size_t)i * n_act + k] * uctrl[k];
    b[bc_dofs[i]] = g;
  }
}

// 1/2 ∫ |u|^2 (flowsolver.py:827-829), element by element (degree-4 integrand: exact with the 7-point rule)
double cpu_energy(int nc, int nn, const double* coords, const int64_t* cells, const int64_t* cell_nodes, const double* u) {
  double e = 0.0;
  for (int c = 0; c < nc; ++c) {
    const int64_t* v = cells + 3 * (size_t)c;
    const int64_t* nd = cell_nodes + 6 * (size_t)c;
    const double x0 = coords[2 * v[0]], y0 = coords[2 * v[0] + 1];
    const double det = (coords[2 * v[1]] - x0) * (coords[2 * v[2] + 1] - y0) - (coords[2 * v[2]] - x0) * (coords[2 * v[1] + 1] - y0);
    for (int q = 0; q < 7; ++q) {
      double ux = 0, uy = 0;
      for (int k = 0; k < 6; ++k) ux += T.phi[q][k] * u[nd[k]], uy += T.phi[q][k] * u[nn + nd[k]];
      e += T.w[q] * 0.5 * det * (ux * ux + uy * uy);
    }
  }
  return 0.5 * e;
}

// sensors: y_s = sum_k w[k] up[idx[k]]
void cpu_sensors(int n_sens, const int64_t* rowptr, const int64_t* idx, const double* w, const double* up, double* y) {
  for (int s = 0; s < n_sens; ++s) {
    double acc = 0.0;
    for (int64_t k = rowptr[s]; k < rowptr[s + 1]; ++k) acc += w[k] * up[idx[k]];
    y[s] = acc;
  }
}

}  // extern "C"
