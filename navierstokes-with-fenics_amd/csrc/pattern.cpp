// Host-side sparsity pattern + slot-map construction (what dolfin's
// SparsityPatternBuilder / TensorLayout do inside FunctionSpace assembly in the
// reference: implicit in every dlfn.*VariationalSolver, e.g.
// source/ns_ipcs_solver.py:136-147).  Runs once per mesh at nsfem_create().
#include "nsfem_internal.hpp"
#include <algorithm>
#include <numeric>

namespace nsfem {

void build_pattern(int n_rows, int n_cols, int n_cells, const int32_t* rowmap, int nr,
                   const int32_t* colmap, int nc, bool want_diag, HostPattern& out) {
  out.n_rows = n_rows;
  out.n_cols = n_cols;
  out.nr = nr;
  out.nc = nc;
  // 1. count candidate columns per row
  std::vector<int64_t> start((size_t)n_rows + 1, 0);
  for (int c = 0; c < n_cells; ++c)
    for (int i = 0; i < nr; ++i) {
      int r = rowmap[(size_t)c * nr + i];
      if (r < 0 || r >= n_rows) throw Error(NSFEM_ERR_ARG, "dof map entry out of range (rows)");
      start[(size_t)r + 1] += nc;
    }
  for (int r = 0; r < n_rows; ++r) start[r + 1] += start[r];
  std::vector<int32_t> cand((size_t)start[n_rows]);
  std::vector<int64_t> fill(start.begin(), start.end() - 1);
  for (int c = 0; c < n_cells; ++c)
    for (int i = 0; i < nr; ++i) {
      int r = rowmap[(size_t)c * nr + i];
      int64_t& f = fill[r];
      for (int j = 0; j < nc; ++j) {
        int cc = colmap[(size_t)c * nc + j];
        if (cc < 0 || cc >= n_cols) throw Error(NSFEM_ERR_ARG, "dof map entry out of range (cols)");
        cand[f++] = cc;
      }
    }
  // 2. sort + unique per row
  out.rowptr.assign((size_t)n_rows + 1, 0);
  for (int r = 0; r < n_rows; ++r) {
    auto b = cand.begin() + start[r], e = cand.begin() + start[r + 1];
    std::sort(b, e);
    auto u = std::unique(b, e);
    out.rowptr[r + 1] = (int32_t)(u - b);
  }
  int64_t total = 0;
  for (int r = 0; r < n_rows; ++r) {
    int32_t len = out.rowptr[r + 1];
    out.rowptr[r] = (int32_t)total;
    total += len;
    if (total > INT32_MAX) throw Error(NSFEM_ERR_ARG, "pattern exceeds int32 nnz");
  }
  out.rowptr[n_rows] = (int32_t)total;
  out.col.resize((size_t)total);
  for (int r = 0; r < n_rows; ++r) {
    int32_t len = out.rowptr[r + 1] - out.rowptr[r];
    std::copy(cand.begin() + start[r], cand.begin() + start[r] + len,
              out.col.begin() + out.rowptr[r]);
  }
  std::vector<int32_t>().swap(cand);
  // 3. diagonal positions
  out.diag.clear();
  if (want_diag) {
    out.diag.assign((size_t)n_rows, -1);
    for (int r = 0; r < n_rows; ++r) {
      auto b = out.col.begin() + out.rowptr[r], e = out.col.begin() + out.rowptr[r + 1];
      auto it = std::lower_bound(b, e, r);
      if (it != e && *it == r) out.diag[r] = (int32_t)(it - out.col.begin());
    }
  }
  // 4. slot map, SoA [nr*nc][n_cells]
  out.slot.resize((size_t)n_cells * nr * nc);
  for (int c = 0; c < n_cells; ++c)
    for (int i = 0; i < nr; ++i) {
      int r = rowmap[(size_t)c * nr + i];
      auto b = out.col.begin() + out.rowptr[r], e = out.col.begin() + out.rowptr[r + 1];
      for (int j = 0; j < nc; ++j) {
        int cc = colmap[(size_t)c * nc + j];
        auto it = std::lower_bound(b, e, cc);
        out.slot[(size_t)(i * nc + j) * n_cells + c] = (int32_t)(it - out.col.begin());
      }
    }
}

// 7-point, degree-5 Radon rule on the reference triangle (weights sum to 1/2) and
// the P2 / P1 Lagrange tables at its points.
void fill_quad_tables(QuadTables& t) {
  const double s15 = std::sqrt(15.0);
  const double a1 = (6.0 - s15) / 21.0, a2 = (6.0 + s15) / 21.0;
  const double w1 = (155.0 - s15) / 2400.0, w2 = (155.0 + s15) / 2400.0;
  const double pts[7][2] = {{1.0 / 3.0, 1.0 / 3.0},
                            {a1, a1}, {1.0 - 2.0 * a1, a1}, {a1, 1.0 - 2.0 * a1},
                            {a2, a2}, {1.0 - 2.0 * a2, a2}, {a2, 1.0 - 2.0 * a2}};
  const double wts[7] = {9.0 / 80.0, w1, w1, w1, w2, w2, w2};
  const double dl[3][2] = {{-1.0, -1.0}, {1.0, 0.0}, {0.0, 1.0}};
  const int pr[3][2] = {{1, 2}, {0, 2}, {0, 1}};
  for (int q = 0; q < 7; ++q) {
    t.w[q] = wts[q];
    double l[3] = {1.0 - pts[q][0] - pts[q][1], pts[q][0], pts[q][1]};
    for (int i = 0; i < 3; ++i) {
      t.phi1[q][i] = l[i];
      t.phi2[q][i] = l[i] * (2.0 * l[i] - 1.0);
      for (int d = 0; d < 2; ++d) t.dphi2[q][i][d] = (4.0 * l[i] - 1.0) * dl[i][d];
    }
    for (int e = 0; e < 3; ++e) {
      int a = pr[e][0], b = pr[e][1];
      t.phi2[q][3 + e] = 4.0 * l[a] * l[b];
      for (int d = 0; d < 2; ++d) t.dphi2[q][3 + e][d] = 4.0 * (l[a] * dl[b][d] + l[b] * dl[a][d]);
    }
  }
  // degree-4, 6-point Strang-Fix rule (FFC's "default" scheme for quadrature degree 4, which the
  // reference's CFL projection requests: source/ns_problem.py:570): Phi[q][i] and its inverse
  const double a = 0.445948490915965, b = 0.091576213509771;
  const double p6[6][2] = {{a, a}, {1.0 - 2.0 * a, a}, {a, 1.0 - 2.0 * a},
                           {b, b}, {1.0 - 2.0 * b, b}, {b, 1.0 - 2.0 * b}};
  double m[6][12];
  for (int q = 0; q < 6; ++q) {
    double l[3] = {1.0 - p6[q][0] - p6[q][1], p6[q][0], p6[q][1]};
    for (int i = 0; i < 3; ++i) t.cfl_phi[q][i] = l[i] * (2.0 * l[i] - 1.0);
    for (int e = 0; e < 3; ++e) t.cfl_phi[q][3 + e] = 4.0 * l[pr[e][0]] * l[pr[e][1]];
    for (int i = 0; i < 6; ++i) { m[q][i] = t.cfl_phi[q][i]; m[q][6 + i] = (i == q) ? 1.0 : 0.0; }
  }
  for (int c = 0; c < 6; ++c) {                       // Gauss-Jordan with partial pivoting
    int piv = c;
    for (int r = c + 1; r < 6; ++r) if (std::fabs(m[r][c]) > std::fabs(m[piv][c])) piv = r;
    for (int k = 0; k < 12; ++k) std::swap(m[c][k], m[piv][k]);
    const double d = 1.0 / m[c][c];
    for (int k = 0; k < 12; ++k) m[c][k] *= d;
    for (int r = 0; r < 6; ++r) {
      if (r == c) continue;
      const double f = m[r][c];
      for (int k = 0; k < 12; ++k) m[r][k] -= f * m[c][k];
    }
  }
  for (int i = 0; i < 6; ++i)
    for (int q = 0; q < 6; ++q) t.cfl_inv[i][q] = m[i][6 + q];
}

void p2_mass_jacobi_bounds(int dim, double& lmin, double& lmax) {
  constexpr int N = 10;
  const int n = dim == 2 ? 6 : 10;
  double S[N][N] = {{0.0}};
  if (dim == 2) {
    QuadTables t;
    fill_quad_tables(t);
    for (int q = 0; q < 7; ++q)
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) S[i][j] += t.w[q] * t.phi2[q][i] * t.phi2[q][j];
  } else {
    QuadTables3 t;
    fill_quad_tables_3d(t);
    for (int q = 0; q < 15; ++q)
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) S[i][j] += t.w[q] * t.phi2[q][i] * t.phi2[q][j];
  }
  double d[N];
  for (int i = 0; i < n; ++i) d[i] = std::sqrt(S[i][i]);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) S[i][j] /= d[i] * d[j];
  // cyclic Jacobi rotations on the symmetric n x n matrix
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) off += S[i][j] * S[i][j];
    if (off < 1e-30) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        if (std::fabs(S[p][q]) < 1e-300) continue;
        const double theta = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
        const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(tt * tt + 1.0), sn = tt * c;
        for (int k = 0; k < n; ++k) {
          const double a = S[k][p], b = S[k][q];
          S[k][p] = c * a - sn * b;
          S[k][q] = sn * a + c * b;
        }
        for (int k = 0; k < n; ++k) {
          const double a = S[p][k], b = S[q][k];
          S[p][k] = c * a - sn * b;
          S[q][k] = sn * a + c * b;
        }
      }
  }
  lmin = lmax = S[0][0];
  for (int i = 1; i < n; ++i) {
    lmin = std::min(lmin, S[i][i]);
    lmax = std::max(lmax, S[i][i]);
  }
}

}  // namespace nsfem

namespace nsfem {

// Inverted index: for every target (CSR slot / dof) the ascending list of sources that
// scatter into it.  Lets the assembly kernels store element tensors with plain stores and
// sum per target in a fixed order (deterministic, no atomics).
void build_inverse_index(int n_targets, int64_t n_sources,
                         const std::function<int32_t(int64_t)>& target_of,
                         std::vector<int32_t>& ptr, std::vector<int32_t>& idx) {
  if (n_sources > INT32_MAX) throw Error(NSFEM_ERR_ARG, "inverse index exceeds int32");
  ptr.assign((size_t)n_targets + 1, 0);
  for (int64_t s = 0; s < n_sources; ++s) ptr[(size_t)target_of(s) + 1]++;
  for (int t = 0; t < n_targets; ++t) ptr[t + 1] += ptr[t];
  idx.resize((size_t)n_sources);
  std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
  for (int64_t s = 0; s < n_sources; ++s) idx[fill[target_of(s)]++] = (int32_t)s;
}

}  // namespace nsfem
