// Geometric multigrid preconditioners for the IPCS pressure Poisson problem and the
// constant part L = alpha0/k M + c_v K of the momentum operator.
//
// The reference solves every system with sparse LU (PETSc): the Krylov
// solvers and their preconditioners are new functionality, judged on converged results.
//
// Hierarchy: [P2 on the fine mesh ->] P1 on the fine mesh -> P1 on nested coarser meshes.
// All spaces are nested, so the Galerkin coarse operators equal the operators assembled
// on the coarse meshes: they are integrated on the device by the same P1 element kernel
// as the fine ones (no sparse triple products).  Smoother: Chebyshev-accelerated Jacobi,
// fused into the SpMV kernel (EPI_CHEB epilogue: one launch per smoothing step);
// transfers are SpMVs with explicit P and R = P^T; Dirichlet dofs are handled by row
// masks propagated to the coarse levels by injection; the coarsest problem (<= ~1000
// unknowns) is solved by one dense mat-vec with a host-computed (pseudo-)inverse.
#include "nsfem_internal.hpp"
#include <algorithm>

namespace nsfem {

// -------------------------------------------------------------------- kernels
__global__ __launch_bounds__(256) void k_cheb_first(int64_t n, const double* __restrict__ b,
                                                    const double* __restrict__ dinv,
                                                    const uint8_t* __restrict__ mask, double c2,
                                                    double* __restrict__ d,
                                                    double* __restrict__ x) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const double v = (mask && mask[i]) ? 0.0 : c2 * dinv[i] * b[i];
    d[i] = v;
    x[i] = v;
  }
}

// additive levels (the product t = A x comes from apply_additive): smoother update, residual,
// diagonal / absolute row sums of the local part, dense scatter of the coarsest part
__global__ __launch_bounds__(256) void k_cheb_update(int64_t n, const double* __restrict__ t,
                                                     const double* __restrict__ b,
                                                     const double* __restrict__ dinv,
                                                     const uint8_t* __restrict__ mask, double c1,
                                                     double c2, double* __restrict__ d,
                                                     const double* __restrict__ x,
                                                     double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    double dn = 0.0, xn = 0.0;
    if (!(mask && mask[i])) {
      dn = c2 * dinv[i] * (b[i] - t[i]);
      if (c1 != 0.0) dn += c1 * d[i];
      xn = x[i] + dn;
    }
    d[i] = dn;
    out[i] = xn;
  }
}
__global__ __launch_bounds__(256) void k_resid_update(int64_t n, const double* __restrict__ t,
                                                      const double* __restrict__ b,
                                                      const uint8_t* __restrict__ mask,
                                                      double* __restrict__ r) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    r[i] = (mask && mask[i]) ? 0.0 : b[i] - t[i];
}
// out[row] = a_ii (what = 0) or sum_j |a_ij| (what = 1) of a scalar CSR matrix
__global__ __launch_bounds__(256) void k_row_measure(int n_rows, const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ diag,
                                                     const double* __restrict__ vals, int what,
                                                     double* __restrict__ out) {
  for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n_rows; row += gridDim.x * blockDim.x) {
    double s = 0.0;
    if (what == 0) s = vals[diag[row]];
    else
      for (int k = rowptr[row]; k < rowptr[row + 1]; ++k) s += fabs(vals[k]);
    out[row] = s;
  }
}
__global__ __launch_bounds__(256) void k_invert_masked(int64_t n, const double* __restrict__ t,
                                                       const uint8_t* __restrict__ mask,
                                                       double* __restrict__ dinv) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    dinv[i] = ((mask && mask[i]) || t[i] == 0.0) ? 0.0 : 1.0 / t[i];
}
__global__ __launch_bounds__(256) void k_max_product(int64_t n, const double* __restrict__ t,
                                                     const double* __restrict__ dinv,
                                                     const uint8_t* __restrict__ mask,
                                                     double* __restrict__ parts) {
  __shared__ double sh[256];
  double best = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    if (!(mask && mask[i])) best = fmax(best, t[i] * dinv[i]);
  sh[threadIdx.x] = best;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) parts[blockIdx.x] = sh[0];
}
// dense[g(i)][g(j)] += a_ij with g(i) = idx[i] (unstructured partitions) or (off + i) % n_glob
__global__ __launch_bounds__(256) void k_scatter_dense(int n_rows, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col,
                                                       const double* __restrict__ vals, int64_t off,
                                                       const int32_t* __restrict__ idx, int n_glob,
                                                       double* __restrict__ dense) {
  for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n_rows; row += gridDim.x * blockDim.x) {
    const size_t gi = idx ? (size_t)idx[row] : (size_t)((off + row) % n_glob);
    for (int k = rowptr[row]; k < rowptr[row + 1]; ++k) {
      const size_t gj = idx ? (size_t)idx[col[k]] : (size_t)((off + col[k]) % n_glob);
      atomicAdd(&dense[gi * n_glob + gj], vals[k]);   // (wrapped tiny periodic levels: rows may coincide)
    }
  }
}
// global coarse vector <-> local coarsest level through an index list (nv values per node)
__global__ __launch_bounds__(256) void k_glob_scatter_add(int64_t n, int nv, const double* __restrict__ b,
                                                          const int32_t* __restrict__ idx,
                                                          double* __restrict__ gb) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * nv;
       t += (int64_t)gridDim.x * blockDim.x) {
    const double v = b[t];
    if (v != 0.0) gb[(int64_t)idx[t / nv] * nv + t % nv] += v;       // (ghost entries are zero; owned ids are unique)
  }
}
__global__ __launch_bounds__(256) void k_glob_gather(int64_t n, int nv, const double* __restrict__ gx,
                                                     const int32_t* __restrict__ idx,
                                                     double* __restrict__ x) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * nv;
       t += (int64_t)gridDim.x * blockDim.x)
    x[t] = gx[(int64_t)idx[t / nv] * nv + t % nv];
}

// x[(row, v)] = sum_c Ainv[v][c][row] b[(c, v)]   (Ainv symmetric, stored per component)
__global__ __launch_bounds__(256) void k_dense_apply(int n, int nv,
                                                     const double* __restrict__ Ainv,
                                                     const double* __restrict__ b,
                                                     double* __restrict__ x) {
  // one wavefront per output (row, component): lanes stride the columns (A is symmetric, so
  // row `row` of A is read contiguously), shuffle reduction
  const int idx = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (idx >= n * nv) return;
  const int v = idx / n, row = idx % n;
  const double* A = Ainv + (size_t)v * n * n + (size_t)row * n;
  double acc = 0.0;
  for (int c = lane; c < n; c += 64) acc += A[c] * b[(size_t)c * nv + v];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) x[(size_t)row * nv + v] = acc;
}

// Gershgorin bound of D^{-1} A over unmasked rows: max_i dinv_i sum_j |a_ij|
__global__ __launch_bounds__(256) void k_gershgorin(int n_rows, int nv,
                                                    const int32_t* __restrict__ rowptr,
                                                    const double* __restrict__ vals,
                                                    const uint8_t* __restrict__ mask,
                                                    const double* __restrict__ dinv,
                                                    double* __restrict__ parts) {
  __shared__ double sh[256];
  double best = 0.0;
  for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n_rows;
       row += gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int k = rowptr[row]; k < rowptr[row + 1]; ++k) s += fabs(vals[k]);
    for (int v = 0; v < nv; ++v) {
      const size_t i = (size_t)row * nv + v;
      if (!(mask && mask[i])) best = fmax(best, s * dinv[i]);
    }
  }
  sh[threadIdx.x] = best;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) parts[blockIdx.x] = sh[0];
}

// max over rows of K_ii / M_ii (both matrices on the same scalar pattern)
__global__ __launch_bounds__(256) void k_diag_ratio(int n_rows, const int32_t* __restrict__ rowptr,
                                                    const int32_t* __restrict__ col,
                                                    const double* __restrict__ M,
                                                    const double* __restrict__ K,
                                                    double* __restrict__ parts) {
  __shared__ double sh[256];
  double best = 0.0;
  for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n_rows;
       row += gridDim.x * blockDim.x)
    for (int k = rowptr[row]; k < rowptr[row + 1]; ++k)
      if (col[k] == row && M[k] > 0.0) best = fmax(best, K[k] / M[k]);
  sh[threadIdx.x] = best;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) parts[blockIdx.x] = sh[0];
}

double diag_ratio_max(hipStream_t s, const Pattern& pat, const double* M, const double* K,
                      double* parts) {
  std::vector<double> hp(kParts);
  hipLaunchKernelGGL(k_diag_ratio, dim3(kParts), dim3(256), 0, s, pat.n_rows, pat.rowptr.p, pat.col.p,
                     M, K, parts);
  NSFEM_HIP(hipGetLastError());
  NSFEM_HIP(hipMemcpyAsync(hp.data(), parts, sizeof(double) * kParts, hipMemcpyDeviceToHost, s));
  NSFEM_HIP(hipStreamSynchronize(s));
  return *std::max_element(hp.begin(), hp.end());
}

// --------------------------------------------------------------- host helpers
// transfer matrix from CSR triplets (rows = finer level, cols = coarser level) + R = P^T
void Transfer::build(hipStream_t s, int n_fine, int n_coarse, const int32_t* rowptr,
                     const int32_t* col, const double* val, int kind) {
  const int nnz = rowptr[n_fine];
  patP.n_rows = n_fine; patP.n_cols = n_coarse; patP.nnz = nnz;
  patP.h_rowptr.assign(rowptr, rowptr + n_fine + 1);
  patP.h_col.assign(col, col + nnz);
  for (int k = 0; k < nnz; ++k)
    NSFEM_REQUIRE(col[k] >= 0 && col[k] < n_coarse, "prolongation column out of range");
  patP.rowptr.upload(patP.h_rowptr, s);
  patP.col.upload(patP.h_col, s);
  P.pat = &patP; P.br = P.bc = 1;
  P.vals.upload(val, (size_t)nnz, s);
  h_val.assign(val, val + nnz);
  lattice = -1;
  NSFEM_HIP(hipStreamSynchronize(s));
  // transpose
  std::vector<int32_t> rp((size_t)n_coarse + 1, 0), rc((size_t)nnz);
  std::vector<double> rv((size_t)nnz);
  for (int k = 0; k < nnz; ++k) rp[col[k] + 1]++;
  for (int j = 0; j < n_coarse; ++j) rp[j + 1] += rp[j];
  std::vector<int32_t> fill(rp.begin(), rp.end() - 1);
  // h_inj: the finer-level node that REPRESENTS a coarse node when row flags (Dirichlet sets) are handed down the
  // hierarchy.  Nested levels: the coinciding node (a row of P with the single entry 1).  Non-nested levels
  // (interpolation between meshes with an odd number of cells): the fine node the coarse hat function weighs most,
  // provided the weight is at least 1/2 -- on a boundary that is a node of the same boundary line.
  h_inj.assign((size_t)n_coarse, -1);
  std::vector<double> best((size_t)n_coarse, 0.0);
  // (the caller knows how it built the levels; only without that knowledge the values decide)
  bool nested = kind != 2;
  for (int i = 0; kind == 0 && i < n_fine && nested; ++i)
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if (val[k] != 1.0 && val[k] != 0.5) { nested = false; break; }
  for (int i = 0; i < n_fine; ++i)
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const int j = col[k];
      rc[fill[j]] = i;
      rv[fill[j]] = val[k];
      fill[j]++;
      if (nested) {
        if (rowptr[i + 1] - rowptr[i] == 1 && std::fabs(val[k] - 1.0) < 1e-14) h_inj[j] = i;
      } else if (val[k] >= 0.5 && val[k] > best[(size_t)j]) {
        best[(size_t)j] = val[k];
        h_inj[j] = i;
      }
    }
  // (on partitioned meshes the coarse ghost line above the strip has no fine counterpart:
  //  h_inj stays -1 there; such nodes are flagged as ghosts by the caller)
  patR.n_rows = n_coarse; patR.n_cols = n_fine; patR.nnz = nnz;
  patR.h_rowptr = rp; patR.h_col = rc;
  patR.rowptr.upload(rp, s);
  patR.col.upload(rc, s);
  R.pat = &patR; R.br = R.bc = 1;
  R.vals.upload(rv, s);
  build_rowblocks(patP, s);
  build_rowblocks(patR, s);
}

// Is P the interpolation between a wf x hf lattice (rows, lexicographic) and its even-even sublattice
// ((wf + 1) / 2 wide): even-even rows one entry 1, the other rows 1/2 at the two coarse nodes (I, J) and
// (I + (i & 1), J + (j & 1))?  (Checked once, on the host copy; the lattice kernel then applies P and R = P^T
// from this rule instead of launching the CSR kernels.)
bool Transfer::is_lattice(int wf, int hf) {
  if (lattice >= 0) return lattice == 1;
  lattice = 0;
  if (wf < 3 || hf < 3 || (wf & 1) == 0 || (hf & 1) == 0) return false;
  const int wc = (wf + 1) / 2, hc = (hf + 1) / 2;
  if (patP.n_rows != wf * hf || patP.n_cols != wc * hc || h_val.size() != (size_t)patP.nnz) return false;
  for (int j = 0; j < hf; ++j)
    for (int i = 0; i < wf; ++i) {
      const int r = j * wf + i, b = patP.h_rowptr[r], n = patP.h_rowptr[r + 1] - b;
      const int pi = i & 1, pj = j & 1, c0 = (j >> 1) * wc + (i >> 1), c1 = c0 + pi + pj * wc;
      if (!pi && !pj) {
        if (n != 1 || patP.h_col[b] != c0 || h_val[b] != 1.0) return false;
      } else {
        if (n != 2 || h_val[b] != 0.5 || h_val[b + 1] != 0.5) return false;
        const int a0 = patP.h_col[b], a1 = patP.h_col[b + 1];
        if (!((a0 == c0 && a1 == c1) || (a0 == c1 && a1 == c0))) return false;
      }
    }
  lattice = 1;
  return true;
}

static void invert_dense(std::vector<double>& a, int n) {
  // Gauss-Jordan with partial pivoting, in place
  std::vector<int> piv(n);
  std::vector<double> inv((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) inv[(size_t)i * n + i] = 1.0;
  for (int c = 0; c < n; ++c) {
    int p = c;
    double best = std::fabs(a[(size_t)c * n + c]);
    for (int r = c + 1; r < n; ++r)
      if (std::fabs(a[(size_t)r * n + c]) > best) { best = std::fabs(a[(size_t)r * n + c]); p = r; }
    if (best == 0.0) throw Error(NSFEM_ERR_BREAKDOWN, "singular coarse multigrid matrix");
    if (p != c)
      for (int k = 0; k < n; ++k) {
        std::swap(a[(size_t)p * n + k], a[(size_t)c * n + k]);
        std::swap(inv[(size_t)p * n + k], inv[(size_t)c * n + k]);
      }
    const double d = 1.0 / a[(size_t)c * n + c];
    for (int k = 0; k < n; ++k) { a[(size_t)c * n + k] *= d; inv[(size_t)c * n + k] *= d; }
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = a[(size_t)r * n + c];
      if (f == 0.0) continue;
      for (int k = 0; k < n; ++k) {
        a[(size_t)r * n + k] -= f * a[(size_t)c * n + k];
        inv[(size_t)r * n + k] -= f * inv[(size_t)c * n + k];
      }
    }
  }
  a.swap(inv);
}

// In-place Gauss-Jordan inversion on the device for the larger coarsest levels (the matrices
// handed in are symmetric positive definite -- Dirichlet rows/columns eliminated, singular
// operators regularised -- so no pivoting is needed): per pivot one kernel saves the pivot row and
// column, one applies the rank-1 update to the whole matrix.
__global__ __launch_bounds__(256) void k_gj_save(int n, int p, const double* __restrict__ A,
                                                 double* __restrict__ rowp, double* __restrict__ colp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  rowp[i] = A[(size_t)p * n + i];
  colp[i] = A[(size_t)i * n + p];
}
__global__ __launch_bounds__(256) void k_gj_update(int n, int p, double* __restrict__ A,
                                                   const double* __restrict__ rowp,
                                                   const double* __restrict__ colp) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)n * n) return;
  const int i = (int)(t / n), j = (int)(t % n);
  const double ipiv = 1.0 / rowp[p];
  double v;
  if (i == p) v = (j == p) ? ipiv : rowp[j] * ipiv;
  else if (j == p) v = -colp[i] * ipiv;
  else v = A[t] - colp[i] * rowp[j] * ipiv;
  A[t] = v;
}

static void invert_dense_any(hipStream_t s, std::vector<double>& a, int n) {
  if (n < 256) { invert_dense(a, n); return; }
  DevBuf<double> A, rowp, colp;
  A.upload(a, s);
  rowp.alloc((size_t)n);
  colp.alloc((size_t)n);
  const int g1 = (n + 255) / 256;
  const int g2 = (int)(((int64_t)n * n + 255) / 256);
  for (int p = 0; p < n; ++p) {
    hipLaunchKernelGGL(k_gj_save, dim3(g1), dim3(256), 0, s, n, p, A.p, rowp.p, colp.p);
    hipLaunchKernelGGL(k_gj_update, dim3(g2), dim3(256), 0, s, n, p, A.p, rowp.p, colp.p);
  }
  NSFEM_HIP(hipGetLastError());
  NSFEM_HIP(hipMemcpyAsync(a.data(), A.p, sizeof(double) * a.size(), hipMemcpyDeviceToHost, s));
  NSFEM_HIP(hipStreamSynchronize(s));
  for (size_t i = 0; i < a.size(); ++i)
    if (!std::isfinite(a[i])) throw Error(NSFEM_ERR_BREAKDOWN, "singular coarse multigrid matrix");
}

// ------------------------------------------------------------------ Multigrid
void Multigrid::setup_work(hipStream_t s) {
  for (size_t l = 0; l < lv.size(); ++l) {
    MGLevel& L = lv[l];
    const size_t n = (size_t)L.n * nv;
    for (DevBuf<double>* b : {&L.xa, &L.xb, &L.r, &L.d, &L.dinv, &L.xc, &L.d2}) { b->alloc(n); b->zero(s); }
    if (L.additive) { L.t.alloc(n); L.t.zero(s); }
    if (l > 0) {
      L.x.alloc(n); L.x.zero(s);
      L.b.alloc(n); L.b.zero(s);
    }
    if (l > 0 || own_mask0) {
      L.own_mask.alloc(n); L.own_mask.zero(s);
      L.mask = L.own_mask.p;
    }
  }
  if (!parts.p) parts.alloc(kParts);
}

// masks (host, level 0) -> coarse levels by injection; dinv, lambda_max, coarse inverse
void Multigrid::refresh(hipStream_t s, const std::vector<uint8_t>& mask0, bool singular) {
  legs_kind = -1;                  // fused launches are planned again from the new masks / coefficients
  std::vector<uint8_t> cur = mask0, nxt;
  std::vector<double> hp(kParts);
  const size_t n_used = truncated() ? active : lv.size();
  for (size_t l = 0; l < n_used; ++l) {
    MGLevel& L = lv[l];
    L.ratio = 0.0;
    L.sidm_for = nullptr;          // the masks may change below: the lattice kernel's byte array is rebuilt
    const size_t n = (size_t)L.n * nv;
    NSFEM_REQUIRE(cur.size() == n, "multigrid mask size mismatch");
    if (l > 0 || own_mask0) {
      NSFEM_HIP(hipMemcpyAsync(L.own_mask.p, cur.data(), n, hipMemcpyHostToDevice, s));
      NSFEM_HIP(hipStreamSynchronize(s));
    }
    if (L.additive) {
      // diagonal and absolute row sums of the whole operator: local parts, ghost rows added at
      // the owners (sum_r sum_j |a^r_ij| >= sum_j |a_ij|: still a Gershgorin bound)
      NSFEM_REQUIRE(nv == 1 && L.A->br == 1, "additive multigrid levels are scalar");
      if (!L.t.p) { L.t.alloc(n); L.t.zero(s); }
      const Pattern& p = *L.A->pat;
      const int g = std::min((p.n_rows + 255) / 256, 2048);
      hipLaunchKernelGGL(k_row_measure, dim3(g), dim3(256), 0, s, p.n_rows, p.rowptr.p, p.diag.p,
                         L.A->vals.p, 0, L.t.p);
      if (comm_active() && L.has_halo) comm->exchange_add(s, L.halo, L.t.p, 1);
      hipLaunchKernelGGL(k_invert_masked, dim3(g), dim3(256), 0, s, (int64_t)n, L.t.p, L.mask, L.dinv.p);
      hipLaunchKernelGGL(k_row_measure, dim3(g), dim3(256), 0, s, p.n_rows, p.rowptr.p, p.diag.p,
                         L.A->vals.p, 1, L.t.p);
      if (comm_active() && L.has_halo) comm->exchange_add(s, L.halo, L.t.p, 1);
      hipLaunchKernelGGL(k_max_product, dim3(kParts), dim3(256), 0, s, (int64_t)n, L.t.p, L.dinv.p,
                         L.mask, parts.p);
    } else {
      launch_inv_diag(s, *L.A, nv, L.mask, L.dinv.p);
      hipLaunchKernelGGL(k_gershgorin, dim3(kParts), dim3(256), 0, s, L.A->pat->n_rows, nv,
                         L.A->pat->rowptr.p, L.A->vals.p, L.mask, L.dinv.p, parts.p);
    }
    NSFEM_HIP(hipGetLastError());
    NSFEM_HIP(hipMemcpyAsync(hp.data(), parts.p, sizeof(double) * kParts, hipMemcpyDeviceToHost, s));
    NSFEM_HIP(hipStreamSynchronize(s));
    L.lmax = *std::max_element(hp.begin(), hp.end());
    if (comm_active()) {
      // every rank must smooth with the same Chebyshev polynomial
      NSFEM_HIP(hipMemcpyAsync(parts.p, &L.lmax, sizeof(double), hipMemcpyHostToDevice, s));
      comm->allreduce_max(s, parts.p, 1);
      NSFEM_HIP(hipMemcpyAsync(&L.lmax, parts.p, sizeof(double), hipMemcpyDeviceToHost, s));
      NSFEM_HIP(hipStreamSynchronize(s));
    }
    if (!(L.lmax > 0.0) || !std::isfinite(L.lmax)) L.lmax = 2.0;
    if (l + 1 < n_used) {
      const std::vector<int32_t>& inj = *L.h_inj;
      const MGLevel& C = lv[l + 1];
      nxt.assign((size_t)C.n * nv, 0);
      for (int j = 0; j < C.n; ++j)
        for (int v = 0; v < nv; ++v) {
          uint8_t m = inj[j] >= 0 ? cur[(size_t)inj[j] * nv + v] : 0;
          if (C.h_ghost && (*C.h_ghost)[j]) m = 2;
          nxt[(size_t)j * nv + v] = m;
        }
      cur.swap(nxt);
    }
  }
  if (truncated()) {
    // the last active level is solved by Chebyshev iteration over [trunc_lmin, 1.05 lmax]:
    // error <= 2 ((sqrt(kappa) - 1) / (sqrt(kappa) + 1))^steps <= trunc_tol
    MGLevel& T = lv[active - 1];
    NSFEM_REQUIRE(trunc_lmin > 0.0, "truncated multigrid cycle without a lower spectral bound");
    const double kappa = std::max(1.05 * T.lmax / trunc_lmin, 1.0 + 1e-9);
    T.ratio = kappa;
    const double sk = std::sqrt(kappa);
    trunc_steps = std::max(2, (int)std::ceil(std::log(2.0 / trunc_tol) / std::log((sk + 1.0) / (sk - 1.0))));
    ready = true;
    return;
  }
  if (comm_active() && !smoother_only) {
    refresh_global_coarse(s, cur, singular);
    ready = true;
    return;
  }
  // coarsest level: dense (pseudo-)inverse per component
  MGLevel& C = lv.back();
  dense_coarse = C.n <= coarse_dense_max;
  if (dense_coarse) {
    const Pattern& p = *C.A->pat;
    const int n = C.n;
    std::vector<double> v((size_t)p.nnz);
    NSFEM_HIP(hipMemcpyAsync(v.data(), C.A->vals.p, sizeof(double) * p.nnz, hipMemcpyDeviceToHost, s));
    NSFEM_HIP(hipStreamSynchronize(s));
    std::vector<double> all((size_t)nv * n * n);
    for (int c = 0; c < nv; ++c) {
      std::vector<double> a((size_t)n * n, 0.0);
      bool any_mask = false;
      for (int i = 0; i < n; ++i) {
        const bool mi = cur[(size_t)i * nv + c] != 0;
        any_mask |= mi;
        for (int k = p.h_rowptr[i]; k < p.h_rowptr[i + 1]; ++k) {
          const int j = p.h_col[k];
          const bool mj = cur[(size_t)j * nv + c] != 0;
          if (!mi && !mj) a[(size_t)i * n + j] = v[k];
        }
        if (mi) a[(size_t)i * n + i] = 1.0;
      }
      const bool sing = singular && !any_mask;
      double gamma = 0.0;
      if (sing) {
        for (int i = 0; i < n; ++i) gamma += a[(size_t)i * n + i];
        gamma /= n;
        for (size_t t = 0; t < a.size(); ++t) a[t] += gamma / n;
      }
      invert_dense_any(s, a, n);
      if (sing)
        for (size_t t = 0; t < a.size(); ++t) a[t] -= 1.0 / (gamma * n);
      // masked rows/cols of the inverse act as zero (corrections vanish there)
      for (int i = 0; i < n; ++i)
        if (cur[(size_t)i * nv + c])
          for (int j = 0; j < n; ++j) a[(size_t)i * n + j] = a[(size_t)j * n + i] = 0.0;
      std::copy(a.begin(), a.end(), all.begin() + (size_t)c * n * n);
    }
    coarse_inv.upload(all, s);
  }
  ready = true;
}

// partitioned meshes: the coarsest problem is solved redundantly on every rank on the GLOBAL
// coarsest mesh; its Dirichlet mask is the all-reduced union of the ranks' owned masks
void Multigrid::refresh_global_coarse(hipStream_t s, const std::vector<uint8_t>& cur,
                                      bool singular) {
  NSFEM_REQUIRE(globA && n_glob > 0, "partitioned multigrid needs the global coarsest mesh");
  const MGLevel& C = lv.back();
  const int n = n_glob;
  std::vector<double> gm((size_t)n * nv, 0.0);
  for (int i = 0; i < C.n; ++i)
    for (int v = 0; v < nv; ++v)
      if (cur[(size_t)i * nv + v] == 1) gm[glob_of((size_t)i) * nv + v] = 1.0;
  gb.alloc((size_t)n * nv);
  gx.alloc((size_t)n * nv);
  NSFEM_HIP(hipMemcpyAsync(gb.p, gm.data(), sizeof(double) * gm.size(), hipMemcpyHostToDevice, s));
  comm->allreduce_sum(s, gb.p, (int64_t)gm.size());
  NSFEM_HIP(hipMemcpyAsync(gm.data(), gb.p, sizeof(double) * gm.size(), hipMemcpyDeviceToHost, s));
  NSFEM_HIP(hipStreamSynchronize(s));
  NSFEM_REQUIRE(!(tail && C.additive), "additive (algebraic Schur) levels need a dense global coarsest "
                                       "problem: build the partition without a replicated tail");
  if (tail) {                      // replicated hierarchy instead of one dense solve
    std::vector<uint8_t> gmask(gm.size());
    for (size_t i = 0; i < gm.size(); ++i) gmask[i] = gm[i] > 0.5 ? 1 : 0;
    tail->refresh(s, gmask, singular);
    dense_coarse = false;
    return;
  }
  const Pattern& p = *globA->pat;
  NSFEM_REQUIRE(p.n_rows == n, "global coarsest operator size mismatch");
  std::vector<double> v((size_t)p.nnz), full;
  if (C.additive) {
    // the global coarsest operator is the sum of the ranks' coarsest parts: scatter into a dense
    // matrix in the global numbering, all-reduce
    NSFEM_REQUIRE(nv == 1, "additive multigrid levels are scalar");
    DevBuf<double> dense;
    dense.alloc((size_t)n * n);
    dense.zero(s);
    const Pattern& cp = *C.A->pat;
    hipLaunchKernelGGL(k_scatter_dense, dim3((cp.n_rows + 255) / 256), dim3(256), 0, s, cp.n_rows,
                       cp.rowptr.p, cp.col.p, C.A->vals.p, glob_off, glob_idx, n, dense.p);
    NSFEM_HIP(hipGetLastError());
    comm->allreduce_sum(s, dense.p, (int64_t)n * n);
    full.resize((size_t)n * n);
    NSFEM_HIP(hipMemcpyAsync(full.data(), dense.p, sizeof(double) * full.size(), hipMemcpyDeviceToHost, s));
    NSFEM_HIP(hipStreamSynchronize(s));
  } else {
    NSFEM_HIP(hipMemcpyAsync(v.data(), globA->vals.p, sizeof(double) * p.nnz, hipMemcpyDeviceToHost, s));
    NSFEM_HIP(hipStreamSynchronize(s));
  }
  std::vector<double> all((size_t)nv * n * n);
  for (int c = 0; c < nv; ++c) {
    std::vector<double> a((size_t)n * n, 0.0);
    bool any_mask = false;
    for (int i = 0; i < n; ++i) {
      const bool mi = gm[(size_t)i * nv + c] > 0.5;
      any_mask |= mi;
      if (C.additive) {
        for (int j = 0; j < n; ++j)
          if (!mi && !(gm[(size_t)j * nv + c] > 0.5)) a[(size_t)i * n + j] = full[(size_t)i * n + j];
      } else {
        for (int k = p.h_rowptr[i]; k < p.h_rowptr[i + 1]; ++k) {
          const int j = p.h_col[k];
          if (!mi && !(gm[(size_t)j * nv + c] > 0.5)) a[(size_t)i * n + j] = v[k];
        }
      }
      if (mi) a[(size_t)i * n + i] = 1.0;
    }
    const bool sing = singular && !any_mask;
    double gamma = 0.0;
    if (sing) {
      for (int i = 0; i < n; ++i) gamma += a[(size_t)i * n + i];
      gamma /= n;
      for (size_t t = 0; t < a.size(); ++t) a[t] += gamma / n;
    }
    invert_dense_any(s, a, n);
    if (sing)
      for (size_t t = 0; t < a.size(); ++t) a[t] -= 1.0 / (gamma * n);
    for (int i = 0; i < n; ++i)
      if (gm[(size_t)i * nv + c] > 0.5)
        for (int j = 0; j < n; ++j) a[(size_t)i * n + j] = a[(size_t)j * n + i] = 0.0;
    std::copy(a.begin(), a.end(), all.begin() + (size_t)c * n * n);
  }
  coarse_inv.upload(all, s);
  dense_coarse = true;
}

void Multigrid::cheb_coeffs(const MGLevel& L, int k, double rho_prev, double& c1, double& c2,
                            double& rho) const {
  const double b = 1.05 * L.lmax, a = b / (L.ratio > 0.0 ? L.ratio : eig_ratio);
  const double theta = 0.5 * (b + a), delta = 0.5 * (b - a), sigma = theta / delta;
  if (k == 0) {
    c1 = 0.0;
    c2 = 1.0 / theta;
    rho = 1.0 / sigma;
  } else {
    rho = 1.0 / (2.0 * sigma - rho_prev);
    c1 = rho * rho_prev;
    c2 = 2.0 * rho / delta;
  }
}

bool Multigrid::lattice_ok(const MGLevel& L) const {
  return !L.additive && !(comm_active() && L.has_halo) && lattice_smoother_available(*L.A, nv);
}

bool ghost_lattice_lines(const std::vector<uint8_t>& g, int W, int H, int& lo, int& hi) {
  lo = hi = -1;
  if (W <= 0 || H <= 0 || (size_t)W * H != g.size()) return false;
  std::vector<int> line(H);
  for (int j = 0; j < H; ++j) {
    int cnt = 0;
    for (int i = 0; i < W; ++i) cnt += g[(size_t)j * W + i] != 0;
    if (cnt != 0 && cnt != W) return false;
    line[j] = cnt == W;
  }
  int a = 0, b = 0;
  while (a < H && line[a]) ++a;
  while (b < H - a && line[H - 1 - b]) ++b;
  for (int j = a; j < H - b; ++j)
    if (line[j]) return false;
  if (a + b >= H) return false;
  lo = a;
  hi = b;
  return true;
}

bool Multigrid::lattice_ok_relaxed(MGLevel& L) {
  if (L.additive || !(comm_active() && L.has_halo) || !relaxed_halo || !partitioned_lattice_kernels() ||
      !lattice_smoother_available(*L.A, nv) || !L.h_ghost)
    return false;
  if (L.ghost_lo == -2) {
    int lo, hi;
    if ((int)L.h_ghost->size() == L.n && ghost_lattice_lines(*L.h_ghost, L.A->dict->lat_w, L.A->dict->lat_h, lo, hi)) {
      L.ghost_lo = lo;
      L.ghost_hi = hi;
    } else {
      L.ghost_lo = L.ghost_hi = -1;
    }
  }
  return L.ghost_lo >= 0;
}

// algorithmic bytes of one launch of the lattice kernel: one byte per row (dictionary entry), the
// mask, the dictionary once, and the vectors the launch reads / writes
int64_t lattice_launch_bytes(const BlockMat& A, int nv, bool from_zero, bool d_in, bool d_out,
                             bool r_out) {
  const int64_t rows = A.pat->n_rows, n = rows * nv;
  const int vecs = (from_zero ? 0 : 1) + 1 /* b */ + 1 /* x_out */ + (d_in ? 1 : 0) + (d_out ? 1 : 0) +
                   (r_out ? 1 : 0);
  return rows + n + (int64_t)A.dict->n_stencils * (A.dict->lmax * 12 + 8) + 8 * n * vecs;
}

void Multigrid::smooth_lattice(hipStream_t s, MGLevel& L, const double* b, const double* x_in,
                               double* x_out, int steps, bool ident_last, double* r_out, const double* xc,
                               const double* rf) {
  NSFEM_REQUIRE(steps >= 1 && steps <= 64, "smoothing sequence too long");
  double c1[64], c2[64], rho = 0.0;
  for (int k = 0; k < steps; ++k) {
    double rn;
    cheb_coeffs(L, k, rho, c1[k], c2[k], rn);
    rho = rn;
  }
  // entry | mask byte of the level: rebuilt when the operator's dictionary or the level's masks changed
  // (Multigrid::refresh resets sidm_for)
  const uint8_t* sidm = nullptr;
  if (L.A->dict->n_stencils <= 64 && nv <= 2) {
    if (L.sidm_for != (const void*)L.A->dict || L.sidm_mask != (const void*)L.mask || L.sidm.n != (size_t)L.n) {
      if (L.sidm.n != (size_t)L.n) L.sidm.alloc((size_t)L.n);
      launch_lattice_sidm(s, *L.A, nv, L.mask, L.sidm.p);
      L.sidm_for = (const void*)L.A->dict;
      L.sidm_mask = (const void*)L.mask;
    }
    sidm = L.sidm.p;
  }
  const bool timed = prof && &L == &lv[0] && prof_n + 2 <= prof_ev.size();
  if (timed) NSFEM_HIP(hipEventRecord(prof_ev[prof_n], s));
  const double* cur = x_in;
  const double* d_cur = nullptr;
  int k = 0;
  while (k < steps) {
    const double* xc_k = k == 0 ? xc : nullptr;          // the first launch applies the fused transfers
    const double* rf_k = k == 0 ? rf : nullptr;
    const bool fz = cur == nullptr && xc_k == nullptr;
    int ns = std::min(steps - k, lattice_smoother_max_steps(*L.A, fz, false));
    double* rr = nullptr;
    if (k + ns == steps && r_out) {          // the residual rides along when the halo allows it
      const int with = lattice_smoother_max_steps(*L.A, fz, true);
      if (ns <= with) rr = r_out;
      else ns = std::max(1, std::min(ns - 1, with));
    }
    const bool last = k + ns == steps;
    double* out = last ? x_out : (cur == L.xa.p ? L.xb.p : L.xa.p);
    double* d_out = last ? nullptr : (d_cur == L.d.p ? L.d2.p : L.d.p);
    NSFEM_REQUIRE(out != cur, "lattice smoother: the caller must smooth out of place");
    const bool strip = comm_active() && L.has_halo;        // (only reached when lattice_ok_relaxed(L))
    ++lattice_launches;
    launch_cheb_lattice(s, *L.A, nv, cur, b, d_cur, out, d_out, rr, L.mask, ns, c1 + k, c2 + k,
                        ident_last && last ? 1 : 0, sidm, xc_k, rf_k, rf_k ? const_cast<double*>(b) : nullptr,
                        strip ? L.ghost_lo : 0, strip ? L.ghost_hi : 0, strip && last ? 1 : 0);
    if (timed) {
      ++prof_launches;
      prof_steps += ns;
      // (a fused prolongation replaces the read of x_in by the 4 x smaller coarse vector unless both are read)
      prof_bytes += lattice_launch_bytes(*L.A, nv, fz || (xc_k && !cur), d_cur != nullptr, d_out != nullptr, rr != nullptr) +
                    (xc_k ? 2 * (int64_t)L.n * nv : 0);
    }
    cur = out;
    d_cur = d_out;
    k += ns;
    if (last && r_out && !rr) {              // (halo too wide for the fused residual)
      launch_residual(s, *L.A, nv, cur, b, r_out, L.mask, MASK_ZERO);
    }
  }
  if (timed) {
    NSFEM_HIP(hipEventRecord(prof_ev[prof_n + 1], s));
    prof_n += 2;
  }
}

// `steps` Chebyshev steps on A x = b.  x_in == nullptr: zero initial guess.  The result
// is written to x_out (which may alias x_in only when steps >= 2).
void Multigrid::smooth(hipStream_t s, MGLevel& L, const double* b, const double* x_in,
                       double* x_out, int steps, bool ghosts_valid, bool ident_last, bool first_done) {
  if (L.additive) {
    smooth_additive(s, L, b, x_in, x_out, steps, first_done);
    return;
  }
  if (!first_done && x_out != x_in && steps >= 1 && lattice_ok(L)) {
    smooth_lattice(s, L, b, x_in, x_out, steps, ident_last, nullptr);
    return;
  }
  // partitioned strip, relaxed halo mode: ONE exchange of the start vector's ghost lines (none from zero, none when
  // the caller filled them), then the whole sequence in the lattice kernel with the ghost lines frozen; the ghost
  // rows of the result are zeros, as the last one-step launch of a sequence leaves them
  if (!first_done && steps >= 1 && lattice_ok_relaxed(L)) {
    const double* in = x_in;
    if (in && !ghosts_valid) comm->exchange(s, L.halo, const_cast<double*>(in), nv);
    if (in && in == x_out) {               // the kernel works out of place
      NSFEM_HIP(hipMemcpyAsync(L.xc.p, in, sizeof(double) * (size_t)L.n * nv, hipMemcpyDeviceToDevice, s));
      in = L.xc.p;
    }
    smooth_lattice(s, L, b, in, x_out, steps, ident_last, nullptr);
    return;
  }
  const int64_t n = (int64_t)L.n * nv;
  double rho = 0.0, c1, c2;
  const double* cur = x_in;
  const bool relaxed = relaxed_halo && comm_active() && L.has_halo;
  bool filled = ghosts_valid && relaxed;
  int k0 = 0;
  if (first_done) {                 // step 0 (from zero) came out of the restriction kernel: x1 in L.xa
    cheb_coeffs(L, 0, 0.0, c1, c2, rho);
    cur = L.xa.p;
    k0 = 1;
  }
  for (int k = k0; k < steps; ++k) {
    double rho_new;
    cheb_coeffs(L, k, rho, c1, c2, rho_new);
    rho = rho_new;
    double* out;
    if (k == steps - 1 && (x_out != cur)) out = x_out;
    else out = (cur == L.xa.p) ? L.xb.p : L.xa.p;
    if (cur == nullptr) {
      int grid = (int)std::min<int64_t>((n + 255) / 256, 2048);
      hipLaunchKernelGGL(k_cheb_first, dim3(grid), dim3(256), 0, s, n, b, L.dinv.p, L.mask, c2,
                         L.d.p, out);
      NSFEM_HIP(hipGetLastError());
    } else {
      const bool need_fill = !relaxed || (!filled && x_in != nullptr);   // from zero: ghosts stay zero
      filled = true;
      // in-situ timing: ONE event pair around the run of consecutive finest-level smoothing
      // launches of this call (a pair per launch adds ~4 us of event overhead to a 40 us kernel)
      // (partitioned levels: halo exchanges sit between the launches -> one pair per launch)
      const bool timed = prof && &L == &lv[0] && prof_n + 2 <= prof_ev.size();
      if (timed && !prof_open) {
        NSFEM_HIP(hipEventRecord(prof_ev[prof_n], s));
        prof_open = true;
      }
      const int gmode = relaxed && k + 1 < steps ? 1 : 0;
      product_with_halo(need_fill && comm_active() && L.has_halo ? comm : nullptr, &L.halo, nv, s, cur,
                        L.A->pat, [&](int phase) {
        launch_cheb_step(s, *L.A, nv, cur, b, L.dinv.p, L.d.p, c1, c2, out, L.mask, gmode, phase,
                         ident_last && k == steps - 1 ? 1 : 0);
      });
      if (prof_open) {
        ++prof_launches;
        ++prof_steps;
        if (comm_active() && L.has_halo) {
          NSFEM_HIP(hipEventRecord(prof_ev[prof_n + 1], s));
          prof_n += 2;
          prof_open = false;
        }
      }
    }
    cur = out;
  }
  if (prof_open) {
    NSFEM_HIP(hipEventRecord(prof_ev[prof_n + 1], s));
    prof_n += 2;
    prof_open = false;
  }
  if (cur != x_out)
    NSFEM_HIP(hipMemcpyAsync(x_out, cur, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
}

// t = A x on an additive level: ghosts of x filled, every local row multiplied, the ghost rows'
// partial sums added at their owners (ghost entries of t are left as partial sums: every consumer
// masks them)
void Multigrid::apply_additive(hipStream_t s, MGLevel& L, const double* x, double* t) {
  halo_fill(s, L, x);
  launch_spmv(s, *L.A, nv, x, t, nullptr, MASK_NONE, 0, 0);
  if (comm_active() && L.has_halo) comm->exchange_add(s, L.halo, t, nv);
}

void Multigrid::smooth_additive(hipStream_t s, MGLevel& L, const double* b, const double* x_in,
                                double* x_out, int steps, bool first_done) {
  const int64_t n = (int64_t)L.n * nv;
  const int grid = (int)std::min<int64_t>((n + 255) / 256, 2048);
  double rho = 0.0, c1, c2;
  const double* cur = x_in;
  int k0 = 0;
  if (first_done) {
    cheb_coeffs(L, 0, 0.0, c1, c2, rho);
    cur = L.xa.p;
    k0 = 1;
  }
  for (int k = k0; k < steps; ++k) {
    double rho_new;
    cheb_coeffs(L, k, rho, c1, c2, rho_new);
    rho = rho_new;
    double* out;
    if (k == steps - 1 && (x_out != cur)) out = x_out;
    else out = (cur == L.xa.p) ? L.xb.p : L.xa.p;
    if (cur == nullptr) {
      hipLaunchKernelGGL(k_cheb_first, dim3(grid), dim3(256), 0, s, n, b, L.dinv.p, L.mask, c2, L.d.p, out);
    } else {
      apply_additive(s, L, cur, L.t.p);
      hipLaunchKernelGGL(k_cheb_update, dim3(grid), dim3(256), 0, s, n, L.t.p, b, L.dinv.p, L.mask,
                         c1, c2, L.d.p, cur, out);
    }
    NSFEM_HIP(hipGetLastError());
    cur = out;
  }
  if (cur != x_out)
    NSFEM_HIP(hipMemcpyAsync(x_out, cur, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
}

void Multigrid::halo_fill(hipStream_t s, const MGLevel& L, const double* v) {
  if (comm_active() && L.has_halo) comm->exchange(s, L.halo, const_cast<double*>(v), nv);
}

// does level l begin its work with a smoothing sequence from a zero start vector?
bool Multigrid::starts_from_zero(size_t l) const {
  if (truncated() && l + 1 == active) return true;
  if (l + 1 == lv.size()) return !(comm_active() && !smoother_only) && !dense_coarse;
  return (pre_degree >= 0 ? pre_degree : degree) > 0;
}

bool Multigrid::restrict_to(hipStream_t s, size_t l, const double* src) {
  MGLevel& L = lv[l];
  MGLevel& C = lv[l + 1];
  static const bool fuse = std::getenv("NSFEM_NO_FUSED_FIRST") == nullptr;
  // (levels smoothed by the lattice kernel run their first step themselves)
  if (fuse && starts_from_zero(l + 1) && !lattice_ok(C) && !lattice_ok_relaxed(C)) {
    double c1, c2, rho;
    cheb_coeffs(C, 0, 0.0, c1, c2, rho);
    launch_spmv_cheb_first(s, *L.R, nv, src, C.b.p, C.mask, C.dinv.p, c2, C.d.p, C.xa.p);
    return true;
  }
  launch_spmv(s, *L.R, nv, src, C.b.p, C.mask, MASK_ZERO);
  return false;
}

// level l hands its right-hand side to level l + 1 inside that level's first smoothing launch (fused restriction)
bool Multigrid::chain_child_forms_b(size_t l) {
  return transfer_lattice(l) && starts_from_zero(l + 1) &&
         !((l + 2 == lv.size()) && !(truncated() && l + 2 == active) && dense_coarse);
}

bool Multigrid::transfer_lattice(size_t l) {
  if (!lattice_transfers_enabled() || l + 1 >= lv.size()) return false;
  MGLevel& L = lv[l];
  if (!L.transfer || !lattice_ok(L) || !lattice_ok(lv[l + 1])) return false;
  const StencilDict& df = *L.A->dict;
  const StencilDict& dc = *lv[l + 1].A->dict;
  if (dc.lat_w != (df.lat_w + 1) / 2 || dc.lat_h != (df.lat_h + 1) / 2) return false;
  return L.transfer->is_lattice(df.lat_w, df.lat_h);
}

// Cycle leg on the multi-step lattice kernel: every smoothing sequence is ONE launch, always out of place;
// pre-smoothing from zero together with the residual; where the transfer to the next level is the lattice
// interpolation, the prolongation is applied by the post-smoothing launch's staging and the restriction by
// the first launch of the coarser level (rf != nullptr: b = R rf has not been formed yet -- this level's
// first launch forms and stores it).
const double* Multigrid::vcycle_lattice(hipStream_t s, size_t l, const double* b, double* x, const double* rf) {
  MGLevel& L = lv[l];
  if (truncated() && l + 1 == active) {
    smooth_lattice(s, L, b, nullptr, x, trunc_steps, identity_rows && l == 0 && trunc_steps >= 2, nullptr, nullptr, rf);
    return x;
  }
  if (l + 1 == lv.size()) {                       // smoothed coarsest level
    smooth_lattice(s, L, b, nullptr, x, coarse_steps, false, nullptr, nullptr, rf);
    return x;
  }
  MGLevel& C = lv[l + 1];
  const int pre = pre_degree >= 0 ? pre_degree : degree;
  const bool ident = identity_rows && l == 0;
  const bool tl = transfer_lattice(l);
  // the child forms its own right-hand side when it begins with a smoothing launch from zero
  const bool child_lattice = tl && starts_from_zero(l + 1) &&
                             !((l + 2 == lv.size()) && !(truncated() && l + 2 == active) && dense_coarse);
  auto descend = [&](const double* fine) -> const double* {
    if (child_lattice) return vcycle_lattice(s, l + 1, C.b.p, C.x.p, fine);
    if (restricted_to == l + 1) {                   // (this level's b was formed together with the parent's)
      restricted_to = 0;
      return vcycle(s, l + 1, C.b.p, C.x.p, false);
    }
    // lattice transfers: the 7-point gather kernel instead of the CSR product with R -- and when the child restricts
    // at once as well (no pre-smoothing, explicit restriction), both restrictions in one launch
    if (tl && pre == 0 && l + 2 < lv.size() && !(truncated() && l + 2 == active) && lattice_ok(C) &&
        transfer_lattice(l + 1) && !chain_child_forms_b(l + 1)) {
      MGLevel& D = lv[l + 2];
      if (launch_restrict_lattice2(s, nv, D.A->dict->lat_w, D.A->dict->lat_h, C.A->dict->lat_w, C.A->dict->lat_h,
                                   L.A->dict->lat_w, L.A->dict->lat_h, fine, C.mask, D.mask, C.b.p, D.b.p)) {
        restricted_to = l + 2;
        return vcycle(s, l + 1, C.b.p, C.x.p, false);
      }
    }
    if (!(tl && launch_restrict_lattice(s, nv, C.A->dict->lat_w, C.A->dict->lat_h, L.A->dict->lat_w, L.A->dict->lat_h,
                                        fine, C.mask, C.b.p)))
      launch_spmv(s, *L.R, nv, fine, C.b.p, C.mask, MASK_ZERO);
    return vcycle(s, l + 1, C.b.p, C.x.p, false);
  };
  if (pre > 0) {
    double* wb = l == 0 ? L.xc.p : x;             // pre-smoothed iterate (level 0: x is the caller's result)
    smooth_lattice(s, L, b, nullptr, wb, pre, false, L.r.p, nullptr, rf);
    const double* xc = descend(L.r.p);
    double* out = l == 0 ? x : L.xc.p;
    if (tl) {
      smooth_lattice(s, L, b, wb, out, degree, ident, nullptr, xc, nullptr);
    } else {
      launch_spmv_accumulate(s, *L.P, nv, xc, wb, L.mask, 0);
      smooth_lattice(s, L, b, wb, out, degree, ident, nullptr);
    }
    return out;
  }
  NSFEM_REQUIRE(!rf, "a level without pre-smoothing cannot form its right-hand side in a smoothing launch");
  const double* xc = descend(b);
  if (tl) {
    smooth_lattice(s, L, b, nullptr, x, degree, ident, nullptr, xc, nullptr);
  } else {
    launch_spmv(s, *L.P, nv, xc, L.xc.p, L.mask, MASK_ZERO);      // x = P x_c (every row stored)
    smooth_lattice(s, L, b, L.xc.p, x, degree, ident, nullptr);
  }
  return x;
}

const double* Multigrid::vcycle(hipStream_t s, size_t l, const double* b, double* x, bool first_done) {
  MGLevel& L = lv[l];
  const int64_t n = (int64_t)L.n * nv;
  if (lattice_ok(L) && !first_done && !(l + 1 == lv.size() && !(truncated() && l + 1 == active) &&
                                        ((comm_active() && !smoother_only) || dense_coarse)))
    return vcycle_lattice(s, l, b, x, nullptr);
  if (truncated() && l + 1 == active) {
    smooth(s, L, b, nullptr, x, trunc_steps, false, identity_rows && l == 0 && trunc_steps >= 2, first_done);
    return x;
  }
  if (l + 1 == lv.size()) {
    if (comm_active() && !smoother_only) {
      // gather the owned right-hand sides into the global coarse vector (ghost entries are
      // zero, so overlapping lines add up correctly), solve redundantly, copy the local part
      gb.zero(s);
      // (periodic partitions: the local level may run past the end of the global numbering and
      // continue at its start -- two segments)
      const int64_t ntot = (int64_t)n_glob * nv, off = (int64_t)glob_off * nv;
      const int ggrid = (int)std::min<int64_t>((n + 255) / 256, 1024);
      // owned entries only (ghost entries of b are zero; on tiny periodic levels a ghost plane may
      // even coincide with an owned plane of the same rank, so plain copies could overwrite data):
      // segment-wise ADD into the zeroed global vector
      if (glob_idx) {
        hipLaunchKernelGGL(k_glob_scatter_add, dim3(ggrid), dim3(256), 0, s, (int64_t)L.n, nv, b, glob_idx, gb.p);
        NSFEM_HIP(hipGetLastError());
      } else
      for (int64_t pos = 0; pos < n;) {
        const int64_t g = (off + pos) % ntot, len = std::min<int64_t>(n - pos, ntot - g);
        launch_axpby(s, len, 1.0, gb.p + g, 1.0, b + pos, gb.p + g);
        pos += len;
      }
      comm->allreduce_sum(s, gb.p, ntot);
      if (tail) {
        (void)tail->vcycle(s, 0, gb.p, gx.p);       // (level 0: the result is in gx)
      } else {
        const int tot = n_glob * nv;
        hipLaunchKernelGGL(k_dense_apply, dim3((tot * 64 + 255) / 256), dim3(256), 0, s, n_glob, nv,
                           coarse_inv.p, gb.p, gx.p);
        NSFEM_HIP(hipGetLastError());
      }
      if (glob_idx) {
        hipLaunchKernelGGL(k_glob_gather, dim3(ggrid), dim3(256), 0, s, (int64_t)L.n, nv, gx.p, glob_idx, x);
        NSFEM_HIP(hipGetLastError());
        return x;
      }
      for (int64_t pos = 0; pos < n;) {
        const int64_t g = (off + pos) % ntot, len = std::min<int64_t>(n - pos, ntot - g);
        NSFEM_HIP(hipMemcpyAsync(x + pos, gx.p + g, sizeof(double) * len, hipMemcpyDeviceToDevice, s));
        pos += len;
      }
      return x;
    }
    if (dense_coarse) {
      const int tot = L.n * nv;
      hipLaunchKernelGGL(k_dense_apply, dim3((tot * 64 + 255) / 256), dim3(256), 0, s, L.n, nv,
                         coarse_inv.p, b, x);
      NSFEM_HIP(hipGetLastError());
    } else {
      smooth(s, L, b, nullptr, x, coarse_steps, false, false, first_done);
    }
    return x;
  }
  MGLevel& C = lv[l + 1];
  const int pre = pre_degree >= 0 ? pre_degree : degree;
  bool child_first = false;
  if (pre > 0) {
    smooth(s, L, b, nullptr, x, pre, false, false, first_done);
    if (L.additive) {
      apply_additive(s, L, x, L.t.p);
      hipLaunchKernelGGL(k_resid_update, dim3((int)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256),
                         0, s, n, L.t.p, b, L.mask, L.r.p);
      NSFEM_HIP(hipGetLastError());
    } else {
      halo_fill(s, L, x);
      launch_residual(s, *L.A, nv, x, b, L.r.p, L.mask, MASK_ZERO);
    }
    halo_fill(s, L, L.r.p);
    child_first = restrict_to(s, l, L.r.p);
  } else {            // no pre-smoothing: x = 0, the residual is b itself
    const double* src = b;
    if (comm_active() && L.has_halo) {
      // the restriction needs the ghost entries of b, but b belongs to the caller (a Krylov
      // vector whose ghost entries must stay zero for the dot products): fill a copy
      NSFEM_HIP(hipMemcpyAsync(L.r.p, b, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
      halo_fill(s, L, L.r.p);
      src = L.r.p;
    }
    child_first = restrict_to(s, l, src);
  }
  const double* xc = vcycle(s, l + 1, C.b.p, C.x.p, child_first);
  if (xc != C.x.p) NSFEM_HIP(hipMemcpyAsync(C.x.p, xc, sizeof(double) * (size_t)C.n * nv, hipMemcpyDeviceToDevice, s));
  // (a child solved on the replicated global coarse mesh comes back with its ghost rows filled from the global
  // solution: no exchange)
  const bool child_global = l + 2 == lv.size() && !(truncated() && l + 2 == active) && comm_active() && !smoother_only;
  if (!child_global) halo_fill(s, C, C.x.p);
  const bool relaxed = relaxed_halo && comm_active() && L.has_halo;
  // relaxed mode: the ghost rows of x take part in the prolongation -- interpolated from the
  // (exchanged) coarse ghosts, added to ghost values that were valid before (filled for the
  // residual), this is exactly what the owner computes for them -> no exchange before smoothing
  if (pre > 0) launch_spmv_accumulate(s, *L.P, nv, C.x.p, x, L.mask, relaxed ? 2 : 0);
  else if (relaxed && lattice_ok_relaxed(L)) {      // (the lattice kernel smooths out of place: prolong into its input)
    launch_spmv(s, *L.P, nv, C.x.p, L.xc.p, L.mask, MASK_ZERO, 2);
    smooth(s, L, b, L.xc.p, x, degree, true, identity_rows && l == 0);
    return x;
  }
  else launch_spmv(s, *L.P, nv, C.x.p, x, L.mask, MASK_ZERO, relaxed ? 2 : 0);   // x = P x_c (every row stored)
  smooth(s, L, b, x, x, degree, relaxed, identity_rows && l == 0);
  (void)n;
  return x;
}

void Multigrid::apply(hipStream_t s, const double* r, double* z) {
  NSFEM_REQUIRE(ready, "multigrid hierarchy not refreshed");
  restricted_to = 0;
  if (vcycle_legs(s, r, z)) return;                 // fused multi-level launches (mglegs.hip)
  (void)vcycle(s, 0, r, z);                         // (level 0: the result is in z)
}

}  // namespace nsfem
