// Host-side sparsity pattern + slot-map construction (what dolfin's
// SparsityPatternBuilder / TensorLayout do inside FunctionSpace assembly in the
// reference: implicit in every dlfn.*VariationalSolver, e.g.
// source/ns_ipcs_solver.py:136-147).  Runs once per mesh at nsfem_create().
#include "nsfem_internal.hpp"
#include <algorithm>
#include <exception>
#include <numeric>
#include <thread>

namespace nsfem {

// host threads of the set-up loops (NSFEM_HOST_THREADS; default min(16, hardware threads): a GPU box gives one
// rank 16 cores)
int host_threads() {
  static const int n = [] {
    const char* e = std::getenv("NSFEM_HOST_THREADS");
    int t = e ? std::atoi(e) : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    return std::max(1, std::min(t, 64));
  }();
  return n;
}

// f(begin, end, thread) over [0, n) cut at the given boundaries (size threads + 1)
template <class F>
static void run_blocks(const std::vector<int64_t>& cut, F&& f) {
  const int nt = (int)cut.size() - 1;
  if (nt <= 1) { f(cut.front(), cut.back(), 0); return; }
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err((size_t)nt);
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      try { f(cut[t], cut[t + 1], t); } catch (...) { err[t] = std::current_exception(); }
    });
  for (auto& x : th) x.join();
  for (auto& e : err) if (e) std::rethrow_exception(e);
}
static std::vector<int64_t> even_cuts(int64_t n, int nt) {
  nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, n));
  std::vector<int64_t> c((size_t)nt + 1);
  for (int t = 0; t <= nt; ++t) c[t] = n * t / nt;
  return c;
}

// Pattern, diagonal positions, slot map and (want_contrib) the inverted index of the slot map.  Everything is
// built ROW-wise from the adjacency row -> (cell, local index), so that row blocks are independent: the blocks
// run on host threads (a 13 M-dof tetrahedral mesh has 3.1e8 candidate entries in its P2 x P2 pattern: 14 s on
// one core, the bulk of nsfem_create).
void build_pattern(int n_rows, int n_cols, int n_cells, const int32_t* rowmap, int nr,
                   const int32_t* colmap, int nc, bool want_diag, HostPattern& out, bool want_contrib) {
  out.n_rows = n_rows;
  out.n_cols = n_cols;
  out.nr = nr;
  out.nc = nc;
  const int nt = host_threads();
  // 0. adjacency: row -> the (cell, i) pairs that hit it, ascending in the cell
  const int64_t n_src = (int64_t)n_cells * nr;
  if (n_src > INT32_MAX) throw Error(NSFEM_ERR_ARG, "mesh too large for int32 adjacency");
  std::vector<int32_t> aptr((size_t)n_rows + 1, 0), adj((size_t)n_src);
  for (int64_t q = 0; q < n_src; ++q) {
    const int r = rowmap[q];
    if (r < 0 || r >= n_rows) throw Error(NSFEM_ERR_ARG, "dof map entry out of range (rows)");
    aptr[(size_t)r + 1]++;
  }
  for (int r = 0; r < n_rows; ++r) aptr[r + 1] += aptr[r];
  {
    std::vector<int32_t> fill(aptr.begin(), aptr.end() - 1);
    for (int64_t q = 0; q < n_src; ++q) adj[fill[rowmap[q]]++] = (int32_t)q;
  }
  for (int64_t q = 0; q < (int64_t)n_cells * nc; ++q)
    if (colmap[q] < 0 || colmap[q] >= n_cols) throw Error(NSFEM_ERR_ARG, "dof map entry out of range (cols)");
  // row blocks balanced by their adjacency
  std::vector<int64_t> cut;
  {
    const int ntr = (int)std::max<int64_t>(1, std::min<int64_t>(nt, n_rows));
    cut.assign((size_t)ntr + 1, 0);
    int r = 0;
    for (int t = 1; t < ntr; ++t) {
      const int64_t target = n_src * t / ntr;
      while (r < n_rows && aptr[r] < target) ++r;
      cut[t] = r;
    }
    cut[ntr] = n_rows;
  }
  // 1. sorted unique columns of every row (thread-local buffers, then one copy into place)
  std::vector<int32_t> len((size_t)n_rows, 0);
  std::vector<std::vector<int32_t>> cols(cut.size() - 1);
  run_blocks(cut, [&](int64_t r0, int64_t r1, int t) {
    std::vector<int32_t>& mine = cols[t];
    std::vector<int32_t> cand;
    for (int64_t r = r0; r < r1; ++r) {
      cand.clear();
      for (int32_t k = aptr[r]; k < aptr[r + 1]; ++k) {
        const int64_t cell = adj[k] / nr;
        const int32_t* cm = colmap + cell * nc;
        cand.insert(cand.end(), cm, cm + nc);
      }
      std::sort(cand.begin(), cand.end());
      const auto u = std::unique(cand.begin(), cand.end());
      len[r] = (int32_t)(u - cand.begin());
      mine.insert(mine.end(), cand.begin(), u);
    }
  });
  out.rowptr.assign((size_t)n_rows + 1, 0);
  int64_t total = 0;
  for (int r = 0; r < n_rows; ++r) {
    out.rowptr[r] = (int32_t)total;
    total += len[r];
    if (total > INT32_MAX) throw Error(NSFEM_ERR_ARG, "pattern exceeds int32 nnz");
  }
  out.rowptr[n_rows] = (int32_t)total;
  out.col.resize((size_t)total);
  run_blocks(cut, [&](int64_t r0, int64_t, int t) {
    std::copy(cols[t].begin(), cols[t].end(), out.col.begin() + out.rowptr[r0]);
    std::vector<int32_t>().swap(cols[t]);
  });
  // 2. diagonal positions
  out.diag.clear();
  if (want_diag) {
    out.diag.assign((size_t)n_rows, -1);
    run_blocks(cut, [&](int64_t r0, int64_t r1, int) {
      for (int64_t r = r0; r < r1; ++r) {
        auto b = out.col.begin() + out.rowptr[r], e = out.col.begin() + out.rowptr[r + 1];
        auto it = std::lower_bound(b, e, (int32_t)r);
        if (it != e && *it == r) out.diag[r] = (int32_t)(it - out.col.begin());
      }
    });
  }
  // 3. slot map, SoA [nr*nc][n_cells]: cells are independent
  out.slot.resize((size_t)n_cells * nr * nc);
  run_blocks(even_cuts(n_cells, nt), [&](int64_t c0, int64_t c1, int) {
    for (int64_t c = c0; c < c1; ++c)
      for (int i = 0; i < nr; ++i) {
        const int r = rowmap[(size_t)c * nr + i];
        auto b = out.col.begin() + out.rowptr[r], e = out.col.begin() + out.rowptr[r + 1];
        for (int j = 0; j < nc; ++j) {
          const int cc = colmap[(size_t)c * nc + j];
          auto it = std::lower_bound(b, e, cc);
          out.slot[(size_t)(i * nc + j) * n_cells + c] = (int32_t)(it - out.col.begin());
        }
      }
  });
  // 4. inverted index of the slot map: per slot the sources cell * (nr nc) + i * nc + j in ascending order (a slot
  // gets at most one entry per cell, and a row's cells are visited in ascending order); the slots of a row belong
  // to the row's thread
  out.cptr.clear();
  out.cidx.clear();
  if (want_contrib) {
    const int64_t loc = (int64_t)nr * nc;
    if ((int64_t)n_cells * loc > INT32_MAX) throw Error(NSFEM_ERR_ARG, "inverse index exceeds int32");
    out.cptr.assign((size_t)total + 1, 0);
    run_blocks(cut, [&](int64_t r0, int64_t r1, int) {
      for (int64_t r = r0; r < r1; ++r)
        for (int32_t k = aptr[r]; k < aptr[r + 1]; ++k) {
          const int64_t cell = adj[k] / nr, i = adj[k] % nr;
          for (int j = 0; j < nc; ++j) out.cptr[(size_t)out.slot[(size_t)(i * nc + j) * n_cells + cell] + 1]++;
        }
    });
    for (int64_t t = 0; t < total; ++t) out.cptr[t + 1] += out.cptr[t];
    out.cidx.resize((size_t)n_cells * loc);
    std::vector<int32_t> cursor(out.cptr.begin(), out.cptr.end() - 1);
    run_blocks(cut, [&](int64_t r0, int64_t r1, int) {
      for (int64_t r = r0; r < r1; ++r)
        for (int32_t k = aptr[r]; k < aptr[r + 1]; ++k) {
          const int64_t cell = adj[k] / nr, i = adj[k] % nr;
          for (int j = 0; j < nc; ++j) {
            const int32_t sl = out.slot[(size_t)(i * nc + j) * n_cells + cell];
            out.cidx[cursor[sl]++] = (int32_t)(cell * loc + i * nc + j);
          }
        }
    });
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
