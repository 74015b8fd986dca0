// Sparse mat-vec, fused Krylov vector kernels, BiCGStab and CG for gfx950.
//
// Replaces PETSc Mat/Vec + sparse LU (PETScLUSolver / dolfin default "lu",
// source/ns_solver_base.py:938; all *VariationalSolver.solve() calls).  The
// reference has no iterative solver; parity is on converged
// solutions.
//
// Design (MI355X):
//   * block-CSR with fp64 blocks (2x2 for the velocity Jacobian, 1x2 / 2x1 for
//     div / grad, scalar blocks applied to two interleaved right-hand sides for
//     the scalar P2 operators): 36 B per 2x2 block instead of 48 B in scalar CSR;
//   * one G-lane group of a 64-wide wavefront per block row, xor-shuffle
//     reduction inside the group, XCD-aware blockIdx remap so that each XCD's L2
//     holds a contiguous slice of x;
//   * Dirichlet conditions are applied as row masks inside the SpMV (identity
//     rows = dolfin's non-symmetric DirichletBC.apply; zero rows on vectors that
//     vanish on the constrained dofs = symmetric elimination for CG), so no
//     matrix is ever modified for boundary conditions;
//   * every dot product emits kParts per-block partial sums; the kernels that
//     consume the scalar re-reduce them in a fixed order (bitwise reproducible,
//     no atomics, no extra launch, no host round trip for alpha/beta/omega).
#include "nsfem_internal.hpp"
#include "lattice_shapes.hpp"
#include <unordered_map>

namespace nsfem {

// ------------------------------------------------------------------- SpMV
// One kernel body, four epilogues:
//   EPI_STORE  y = A x                     EPI_RESID  y = b - A x
//   EPI_ACCUM  y += A x  (prolongation)    EPI_CHEB   Chebyshev/Jacobi smoother step:
//                                            d = c1 d + c2 dinv (b - A x); y = x + d
// Row masks (Dirichlet dofs): identity rows (dolfin's non-symmetric bc.apply) or zero
// rows (symmetric elimination on vectors that vanish on the constrained dofs).
enum Epi { EPI_STORE = 0, EPI_RESID = 1, EPI_ACCUM = 2, EPI_CHEB = 3 };

struct SpmvArgs {
  const double* x;
  const double* b;
  double* y;
  const uint8_t* mask;
  int maskmode;
  const double* dinv;
  double* d;
  double c1, c2;
  int nt;          // stream matrix values / column ids with non-temporal loads
  int ghost;       // rows flagged 2 (ghosts of a partitioned mesh): 0 output 0, 1 carry x
                   // (smoother step with frozen ghost values), 2 computed like free rows
  int skip0, skipn; // stream kernel: logical row blocks >= skip0 are shifted by skipn
  int phase;       // 0 all row blocks, 1 interior blocks only, 2 halo-adjacent blocks only
  double* y2;      // EPI_STORE only, non-null: the first Chebyshev step from a zero start on the vector
                   // just produced (a restricted residual): d = y2 = c1 * dinv * y  (fused k_cheb_first)
  int ident;       // smoother step: rows flagged 1 take y = b (identity rows of a preconditioner for a
                   // Newton matrix with dolfin-style Dirichlet rows) instead of 0
  int dict_ok;     // the product may use the matrix's stencil-dictionary copy (smoothing steps only:
                   // the copy equals the CSR matrix to the dedupe tolerance, not bitwise)
  int dbg;         // knock-out build only (NSFEM_KNOCKOUTS; wrong results): 1 = gather from
                   // the chunk's own rows (perfectly local x), 2 = skip the x gather altogether
  // dictionary kernel, EPI_STORE: after the product row r also adds the run gptr[r] .. gptr[r + 1] of
  // NV-wide entries of gbuf (the node-sorted element vectors of the matrix-free convection action: the node
  // gather fused into the product with the constant part of the Jacobian); null: off
  const int32_t* gptr;
  const double* gbuf;
};
struct XcdSplit { int s[9]; };   // SELL kernel: workgroup range [s[x], s[x+1]) of XCD x (s[8] = 0: round robin)

template <int BR, int BC, int NV, int G, int EPI>
__global__ __launch_bounds__(256) void k_spmv(int n_rows, const int32_t* __restrict__ rowptr,
                                              const int32_t* __restrict__ col,
                                              const double* __restrict__ vals, SpmvArgs a) {
  constexpr int NO = BR * NV;            // outputs per block row
  constexpr int RPB = 256 / G;           // block rows per workgroup
  // XCD-aware remap: workgroups b, b+8, ... share an XCD (round-robin dispatch);
  // give each XCD one contiguous range of rows.  gridDim.x is a multiple of 8.
  const int per = gridDim.x >> 3;
  const int lb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int row = lb * RPB + threadIdx.x / G;
  if (row >= n_rows) return;
  const int lane = threadIdx.x % G;
  const double* __restrict__ x = a.x;
  double acc[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) acc[o] = 0.0;
  const int s = rowptr[row], e = rowptr[row + 1];
  for (int k = s + lane; k < e; k += G) {
    double av[BR * BC], xv[BC * NV];
    int c;
    if (a.nt) {
      c = __builtin_nontemporal_load(col + k);
#pragma unroll
      for (int t = 0; t < BR * BC; ++t)
        av[t] = __builtin_nontemporal_load(vals + (size_t)k * (BR * BC) + t);
    } else {
      c = col[k];
#pragma unroll
      for (int t = 0; t < BR * BC; ++t) av[t] = vals[(size_t)k * (BR * BC) + t];
    }
#pragma unroll
    for (int t = 0; t < BC * NV; ++t) xv[t] = x[(size_t)c * (BC * NV) + t];
#pragma unroll
    for (int r = 0; r < BR; ++r)
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int cc = 0; cc < BC; ++cc) acc[r * NV + v] += av[r * BC + cc] * xv[cc * NV + v];
  }
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1)
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[o] += __shfl_xor(acc[o], off, G);
  if (lane < NO) {
    double val = acc[0];
#pragma unroll
    for (int o = 1; o < NO; ++o)
      if (lane == o) val = acc[o];
    const size_t idx = (size_t)row * NO + lane;
    // mask values: 0 free, 1 Dirichlet (mode-dependent), 2 ghost of a partitioned mesh (the
    // owner computes the row: output 0, which also keeps ghosts out of the dot products)
    int mv = (a.maskmode != MASK_NONE) ? a.mask[idx] : 0;
    if (mv == 2 && a.ghost == 2) mv = 0;
    const bool m = mv != 0;
    if (mv == 2) {
      if (EPI == EPI_CHEB) {
        a.d[idx] = 0.0;
        a.y[idx] = a.ghost == 1 ? x[idx] : 0.0;
      } else {
        a.y[idx] = 0.0;
        if (EPI == EPI_STORE && a.y2) a.d[idx] = a.y2[idx] = 0.0;
      }
    } else if (EPI == EPI_RESID) {
      // identity rows: b - x ; zero rows: 0
      if (m)
        val = (a.maskmode == MASK_IDENTITY) ? a.b[idx] - x[idx] : 0.0;
      else
        val = a.b[idx] - val;
      a.y[idx] = val;
    } else if (EPI == EPI_ACCUM) {
      // y += c2 * A x ; masked rows: zero (MASK_ZERO) or left untouched (MASK_IDENTITY)
      if (!m) a.y[idx] += a.c2 * val;
      else if (a.maskmode == MASK_ZERO) a.y[idx] = 0.0;
    } else if (EPI == EPI_CHEB) {
      double dn = 0.0, xn = 0.0;
      if (!m) {
        dn = a.c2 * a.dinv[idx] * (a.b[idx] - val);
        if (a.c1 != 0.0) dn += a.c1 * a.d[idx];
        xn = x[idx] + dn;
      } else if (a.ident) {
        xn = a.b[idx];
      }
      a.d[idx] = dn;
      a.y[idx] = xn;
    } else {
      if (m) val = (a.maskmode == MASK_IDENTITY) ? x[idx] : 0.0;
      else val *= a.c2;
      a.y[idx] = val;
      if (a.y2) {
        const double dn = m ? 0.0 : a.c1 * a.dinv[idx] * val;
        a.d[idx] = dn;
        a.y2[idx] = dn;
      }
    }
  }
}

// --- CSR-stream variant (Greathouse & Daga): a workgroup owns a contiguous block of rows with
// <= kStreamNnz nonzeros; phase 1 streams the nonzeros with ALL lanes (perfectly coalesced
// value / column loads, one nonzero per lane and pass), multiplies with the gathered x and parks
// the products in LDS; phase 2 sums each row's contiguous LDS segment (ascending order) and runs
// the same epilogue.  No idle lanes on the 9/19-entry rows of the P2 operators.
#ifndef NSFEM_STREAM_NNZ
#define NSFEM_STREAM_NNZ 1024
#endif
constexpr int kStreamNnz = NSFEM_STREAM_NNZ;
// which shapes use the stream kernel unless NSFEM_SPMV_STREAM overrides it (set from the
// measured sweep, see profiles/)
static inline bool kStreamDefault(bool shape22) { (void)shape22; return true; }

// version 1 (round 1) of the kernel: the default (NSFEM_STREAM_V=2 selects the wide-load variant below)
template <int BR, int BC, int NV, int EPI>
__global__ __launch_bounds__(256) void k_spmv_stream_v1(int n_rblk, const int32_t* __restrict__ rblk,
                                                     const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ col,
                                                     const double* __restrict__ vals, SpmvArgs a) {
  constexpr int NO = BR * NV;
  __shared__ double prod[kStreamNnz * NO];
  const int per = gridDim.x >> 3;
  int lb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (lb >= n_rblk) return;
  if (lb >= a.skip0) lb += a.skipn;     // halo-adjacent launch: jump over the interior row blocks
  const int r0 = rblk[lb], r1 = rblk[lb + 1];
  const int s0 = rowptr[r0], s1 = rowptr[r1];
  const double* __restrict__ x = a.x;
  // epilogue operands of this thread's first output entry: issued before the streaming phase so
  // that their latency hides behind it (the smoother epilogue reads b, dinv, d, x and the mask)
  const int nout = (r1 - r0) * NO;
  const bool pre = (int)threadIdx.x < nout;
  const size_t pidx = (size_t)r0 * NO + threadIdx.x;
  int pmv = 0;
  double pb = 0.0, pdinv = 0.0, pd = 0.0, px = 0.0;
  if (pre && (EPI == EPI_CHEB || EPI == EPI_RESID)) {
    pmv = (a.maskmode != MASK_NONE) ? a.mask[pidx] : 0;
    pb = a.b[pidx];
    if (EPI == EPI_CHEB) {
      pdinv = a.dinv[pidx];
      if (a.c1 != 0.0) pd = a.d[pidx];
      px = x[pidx];
    }
  }
  // streaming phase: a chunk holds <= kStreamNnz = 4 x 256 nonzeros, i.e. at most 4 per lane.
  // All column ids and values of a lane are loaded first (independent, coalesced), then all x
  // gathers are issued, then the products go to LDS: 4 gathers in flight per lane instead of a
  // load -> gather -> store dependency chain per nonzero.
  constexpr int kPer = kStreamNnz / 256;
  int cc_[kPer];
  double av[kPer][BR * BC];
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int k = s0 + (int)threadIdx.x + u * 256;
    cc_[u] = -1;
    if (k < s1) {
      if (a.nt) {        // matrix streams bypass the caches' LRU (they are read once per launch)
        cc_[u] = __builtin_nontemporal_load(col + k);
#pragma unroll
        for (int t = 0; t < BR * BC; ++t)
          av[u][t] = __builtin_nontemporal_load(vals + (size_t)k * (BR * BC) + t);
      } else {
        cc_[u] = col[k];
#pragma unroll
        for (int t = 0; t < BR * BC; ++t) av[u][t] = vals[(size_t)k * (BR * BC) + t];
      }
    }
  }
  double xv[kPer][BC * NV];
#pragma unroll
  for (int u = 0; u < kPer; ++u)
    if (cc_[u] >= 0) {
#pragma unroll
      for (int t = 0; t < BC * NV; ++t) xv[u][t] = x[(size_t)cc_[u] * (BC * NV) + t];
    }
#pragma unroll
  for (int u = 0; u < kPer; ++u)
    if (cc_[u] >= 0) {
      const int kk = (int)threadIdx.x + u * 256;
#pragma unroll
      for (int r = 0; r < BR; ++r)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          double acc = 0.0;
#pragma unroll
          for (int cc = 0; cc < BC; ++cc) acc += av[u][r * BC + cc] * xv[u][cc * NV + v];
          prod[kk * NO + r * NV + v] = acc;
        }
    }
  __syncthreads();
  // one thread per output entry (row, o)
  for (int t = threadIdx.x; t < nout; t += 256) {
    const int row = r0 + t / NO, o = t % NO;
    double val = 0.0;
    int k = rowptr[row] - s0;
    const int e = rowptr[row + 1] - s0;
    for (; k + 4 <= e; k += 4) {          // 4 LDS reads in flight, summed in ascending order
      const double v0 = prod[k * NO + o], v1 = prod[(k + 1) * NO + o];
      const double v2 = prod[(k + 2) * NO + o], v3 = prod[(k + 3) * NO + o];
      val += v0;
      val += v1;
      val += v2;
      val += v3;
    }
    for (; k < e; ++k) val += prod[k * NO + o];
    const size_t idx = (size_t)row * NO + o;
    const bool first = (EPI == EPI_CHEB || EPI == EPI_RESID) && t == (int)threadIdx.x;
    int mv = first ? pmv : ((a.maskmode != MASK_NONE) ? a.mask[idx] : 0);
    if (mv == 2 && a.ghost == 2) mv = 0;
    const bool m = mv != 0;
    if (mv == 2) {
      if (EPI == EPI_CHEB) {
        a.d[idx] = 0.0;
        a.y[idx] = a.ghost == 1 ? (first ? px : x[idx]) : 0.0;
      } else {
        a.y[idx] = 0.0;
        if (EPI == EPI_STORE && a.y2) a.d[idx] = a.y2[idx] = 0.0;
      }
    } else if (EPI == EPI_RESID) {
      const double bv = first ? pb : a.b[idx];
      if (m)
        val = (a.maskmode == MASK_IDENTITY) ? bv - x[idx] : 0.0;
      else
        val = bv - val;
      a.y[idx] = val;
    } else if (EPI == EPI_ACCUM) {
      if (!m) a.y[idx] += a.c2 * val;
      else if (a.maskmode == MASK_ZERO) a.y[idx] = 0.0;
    } else if (EPI == EPI_CHEB) {
      double dn = 0.0, xn = 0.0;
      if (!m) {
        dn = a.c2 * (first ? pdinv : a.dinv[idx]) * ((first ? pb : a.b[idx]) - val);
        if (a.c1 != 0.0) dn += a.c1 * (first ? pd : a.d[idx]);
        xn = (first ? px : x[idx]) + dn;
      } else if (a.ident) {
        xn = first ? pb : a.b[idx];
      }
      a.d[idx] = dn;
      a.y[idx] = xn;
    } else {
      if (m) val = (a.maskmode == MASK_IDENTITY) ? x[idx] : 0.0;
      else val *= a.c2;
      a.y[idx] = val;
      if (a.y2) {
        const double dn = m ? 0.0 : a.c1 * a.dinv[idx] * val;
        a.d[idx] = dn;
        a.y2[idx] = dn;
      }
    }
  }
}

// Version 2 of the streaming phase (round 2): the kernel was bound by the length of its dependent
// load chain times the workgroups a CU can hold, not by HBM (scripts/micro/readbw.hip: 6.2 TB/s
// read-only, 5.2 TB/s 3 reads + 1 write on the same box against 3.7-4.7 TB/s here):
//   * every lane owns FOUR CONSECUTIVE nonzeros of a 16-byte aligned window over the chunk: one
//     16-byte column load and two 16-byte value loads per lane (scalar blocks) instead of eight
//     4 / 8-byte loads -- the chunk limit is 1020 nonzeros so that the aligned window fits;
//   * one 16-byte record {r0, r1, s0, s1} per chunk instead of two dependent index loads;
//   * the row pointers of the chunk's rows are fetched at the start (latency hidden behind the
//     streaming phase) and parked in LDS for the reduction phase.
template <int BR, int BC, int NV, int EPI>
__global__ __launch_bounds__(256) void k_spmv_stream(int n_rblk, const int4* __restrict__ rbinfo,
                                                     const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ col,
                                                     const double* __restrict__ vals, SpmvArgs a) {
  constexpr int NO = BR * NV;
  constexpr int BS = BR * BC;
  // dynamic LDS: the products of the chunk, then its row pointers (sized for the pattern's
  // longest chunk: the 3-component kernel must stay below 160 KB / 6 per workgroup)
  extern __shared__ double smem[];
  double* prod = smem;
  int32_t* rp = reinterpret_cast<int32_t*>(smem + kStreamNnz * NO);
  const int per = gridDim.x >> 3;
  int lb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (lb >= n_rblk) return;
  if (lb >= a.skip0) lb += a.skipn;     // halo-adjacent launch: jump over the interior row blocks
  const int4 info = rbinfo[lb];
  const int r0 = info.x, r1 = info.y, s0 = info.z, s1 = info.w;
  const double* __restrict__ x = a.x;
  const int tid = threadIdx.x;
  // row pointers of the chunk (relative to s0) -> LDS, needed after the barrier only
  const int nrows = r1 - r0;
  for (int t = tid; t <= nrows; t += 256) rp[t] = rowptr[r0 + t] - s0;
  // epilogue operands of this thread's first output entry: issued before the streaming phase so
  // that their latency hides behind it (the smoother epilogue reads b, dinv, d, x and the mask)
  const int nout = nrows * NO;
  const bool pre = tid < nout;
  const size_t pidx = (size_t)r0 * NO + tid;
  int pmv = 0;
  double pb = 0.0, pdinv = 0.0, pd = 0.0, px = 0.0;
  if (pre && (EPI == EPI_CHEB || EPI == EPI_RESID)) {
    pmv = (a.maskmode != MASK_NONE) ? a.mask[pidx] : 0;
    pb = a.b[pidx];
    if (EPI == EPI_CHEB) {
      pdinv = a.dinv[pidx];
      if (a.c1 != 0.0) pd = a.d[pidx];
      px = x[pidx];
    }
  }
  // streaming phase: lane `tid` owns the nonzeros a0 + 4 tid .. a0 + 4 tid + 3 of the aligned
  // window starting at a0 = s0 rounded down to a multiple of 4 (s1 - a0 <= 1023 by construction)
  const int a0 = s0 & ~3;
  const int k0 = a0 + 4 * tid;
  int cc_[4];
  double av[4][BS];
#pragma unroll
  for (int u = 0; u < 4; ++u) cc_[u] = -1;
  if (k0 < s1) {
    // (k0 is a multiple of 4: 16-byte aligned column quad, 32-byte aligned value quad; every device
    // buffer carries 64 bytes of slack (DevBuf::alloc), entries outside [s0, s1) are discarded)
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v4i* cq4 = reinterpret_cast<const v4i*>(col + k0);
    const v4i c4 = a.nt ? __builtin_nontemporal_load(cq4) : *cq4;
#pragma unroll
    for (int u = 0; u < 4; ++u) cc_[u] = (k0 + u >= s0 && k0 + u < s1) ? c4[u] : -1;
    if (BS == 1) {
      const v2d* vq = reinterpret_cast<const v2d*>(vals + k0);
      const v2d v01 = a.nt ? __builtin_nontemporal_load(vq) : vq[0];
      const v2d v23 = a.nt ? __builtin_nontemporal_load(vq + 1) : vq[1];
      av[0][0] = v01[0]; av[1][0] = v01[1]; av[2][0] = v23[0]; av[3][0] = v23[1];
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (cc_[u] >= 0) {
#pragma unroll
          for (int t = 0; t < BS; ++t)
            av[u][t] = a.nt ? __builtin_nontemporal_load(vals + (size_t)(k0 + u) * BS + t)
                            : vals[(size_t)(k0 + u) * BS + t];
        }
    }
  }
  if (NSFEM_KO(a.dbg == 1)) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (cc_[u] >= 0) cc_[u] = r0 + (4 * tid + u) % nrows;
  }
  double xv[4][BC * NV];
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (cc_[u] >= 0) {
#pragma unroll
      for (int t = 0; t < BC * NV; ++t)
        xv[u][t] = NSFEM_KO(a.dbg == 2) ? (double)cc_[u] : x[(size_t)cc_[u] * (BC * NV) + t];
    }
  // products -> LDS.  Nonzero q = 4 tid + u of the aligned window is parked at slot u * 256 + tid:
  // consecutive lanes write consecutive slots (the natural slot q would put the lanes 4 NO doubles
  // apart: a 16-way bank conflict)
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (cc_[u] >= 0) {
      const int slot = u * 256 + tid;
#pragma unroll
      for (int r = 0; r < BR; ++r)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          double acc = 0.0;
#pragma unroll
          for (int cc = 0; cc < BC; ++cc) acc += av[u][r * BC + cc] * xv[u][cc * NV + v];
          prod[slot * NO + r * NV + v] = acc;
        }
    }
  __syncthreads();
  // one thread per output entry (row, o); window position q of the row's k-th nonzero is
  // k + (s0 - a0), its slot (q & 3) * 256 + (q >> 2)
  const int shift = s0 - a0;
  for (int t = tid; t < nout; t += 256) {
    const int lrow = t / NO, o = t % NO;
    const int row = r0 + lrow;
    double val = 0.0;
    int q = rp[lrow] + shift;
    const int e = rp[lrow + 1] + shift;
    for (; q + 4 <= e; q += 4) {          // 4 LDS reads in flight, summed in ascending order
      const double v0 = prod[(((q)&3) * 256 + ((q) >> 2)) * NO + o];
      const double v1 = prod[(((q + 1) & 3) * 256 + ((q + 1) >> 2)) * NO + o];
      const double v2 = prod[(((q + 2) & 3) * 256 + ((q + 2) >> 2)) * NO + o];
      const double v3 = prod[(((q + 3) & 3) * 256 + ((q + 3) >> 2)) * NO + o];
      val += v0;
      val += v1;
      val += v2;
      val += v3;
    }
    for (; q < e; ++q) val += prod[((q & 3) * 256 + (q >> 2)) * NO + o];
    const size_t idx = (size_t)row * NO + o;
    const bool first = (EPI == EPI_CHEB || EPI == EPI_RESID) && t == tid;
    int mv = first ? pmv : ((a.maskmode != MASK_NONE) ? a.mask[idx] : 0);
    if (mv == 2 && a.ghost == 2) mv = 0;
    const bool m = mv != 0;
    if (mv == 2) {
      if (EPI == EPI_CHEB) {
        a.d[idx] = 0.0;
        a.y[idx] = a.ghost == 1 ? (first ? px : x[idx]) : 0.0;
      } else {
        a.y[idx] = 0.0;
        if (EPI == EPI_STORE && a.y2) a.d[idx] = a.y2[idx] = 0.0;
      }
    } else if (EPI == EPI_RESID) {
      const double bv = first ? pb : a.b[idx];
      if (m)
        val = (a.maskmode == MASK_IDENTITY) ? bv - x[idx] : 0.0;
      else
        val = bv - val;
      a.y[idx] = val;
    } else if (EPI == EPI_ACCUM) {
      if (!m) a.y[idx] += a.c2 * val;
      else if (a.maskmode == MASK_ZERO) a.y[idx] = 0.0;
    } else if (EPI == EPI_CHEB) {
      double dn = 0.0, xn = 0.0;
      if (!m) {
        dn = a.c2 * (first ? pdinv : a.dinv[idx]) * ((first ? pb : a.b[idx]) - val);
        if (a.c1 != 0.0) dn += a.c1 * (first ? pd : a.d[idx]);
        xn = (first ? px : x[idx]) + dn;
      } else if (a.ident) {
        xn = first ? pb : a.b[idx];
      }
      a.d[idx] = dn;
      a.y[idx] = xn;
    } else {
      if (m) val = (a.maskmode == MASK_IDENTITY) ? x[idx] : 0.0;
      else val *= a.c2;
      a.y[idx] = val;
      if (a.y2) {
        const double dn = m ? 0.0 : a.c1 * a.dinv[idx] * val;
        a.d[idx] = dn;
        a.y2[idx] = dn;
      }
    }
  }
}

// --- SELL-64 variant: one wavefront per slice of 64 rows, lane r walks row r through the
// column-major slice (coalesced column / value loads, one entry of 64 different rows per pass).
// On the parity-class numberings of the structured meshes the 64 column ids of a pass are
// consecutive, so the x gather of the wave is ONE contiguous run (64 NV doubles) instead of the
// ~50 scattered cache lines a CSR pass over 64 consecutive nonzeros of 2-3 rows touches.  No LDS,
// no cross-lane reduction; the epilogues are those of the other kernels.  Same ascending order of
// a row's entries as the CSR kernels.
template <int NV, int EPI, int U, int PIPE>
__global__ __launch_bounds__(256) void k_spmv_sell(int n_rows, int n_slices, int n_wg,
                                                   const int32_t* __restrict__ sptr,
                                                   const int32_t* __restrict__ scol,
                                                   const double* __restrict__ sval, SpmvArgs a,
                                                   XcdSplit xs) {
  // workgroups b, b + 8, ... run on the same XCD (round-robin dispatch): XCD x walks its own
  // contiguous, nonzero-balanced range of workgroups, so its L2 sees one spatial slab of x
  int wg;
  if (xs.s[8] > 0) {
    const int xcd = blockIdx.x & 7;
    wg = xs.s[xcd] + (int)(blockIdx.x >> 3);
    if (wg >= xs.s[xcd + 1]) return;
  } else {
    wg = blockIdx.x;
    if (wg >= n_wg) return;
    if (wg >= a.skip0) wg += a.skipn;
  }
  const int slice = wg * 4 + ((int)threadIdx.x >> 6);
  if (slice >= n_slices) return;
  const int lane = threadIdx.x & 63;
  const int row = slice * 64 + lane;
  const bool live = row < n_rows;
  const int base = sptr[slice];
  const int w = (sptr[slice + 1] - base) >> 6;           // wave-uniform
  const double* __restrict__ x = a.x;
  double acc[NV];
#pragma unroll
  for (int o = 0; o < NV; ++o) acc[o] = 0.0;
  const int32_t* __restrict__ cp = scol + (size_t)base + lane;
  const double* __restrict__ vp = sval + (size_t)base + lane;
  // groups of U entries per lane: U column ids + U values (coalesced, non-temporal), then the U x
  // gathers, all independent -> U (1 + NV) loads in flight per lane.  PIPE: the column ids / values
  // of the next group are requested before the gathers of the current one are consumed.
  // branch-free: entries past the slice width re-read its last entry with a zero weight (the
  // select is wave-uniform); vmcnt retires loads in issue order, so the next group's column /
  // value loads are issued AFTER the gathers of the current group and stay in flight while the
  // gathers are consumed
  int c[U], cn[U];
  double v[U], vn[U];
  auto fetch = [&](int k0, int* cc, double* vv) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = k0 + u < w ? k0 + u : w - 1;
      cc[u] = __builtin_nontemporal_load(cp + (size_t)kk * 64);
      const double t = __builtin_nontemporal_load(vp + (size_t)kk * 64);
      vv[u] = k0 + u < w ? t : 0.0;
    }
  };
  if (w > 0) {
    fetch(0, c, v);
    for (int k = 0; k < w; k += U) {
      double xv[U][NV];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < NV; ++t) xv[u][t] = x[(size_t)c[u] * NV + t];
      if (PIPE && k + U < w) fetch(k + U, cn, vn);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < NV; ++t) acc[t] += v[u] * xv[u][t];
      if (PIPE) {
#pragma unroll
        for (int u = 0; u < U; ++u) { c[u] = cn[u]; v[u] = vn[u]; }
      } else if (k + U < w) {
        fetch(k + U, c, v);
      }
    }
  }
  // epilogue with coalesced vector traffic: the wave's 64 rows x NV outputs are the CONTIGUOUS
  // entries 64 NV slice .. 64 NV (slice + 1) of the vectors; in pass p lane l handles entry
  // e = 64 p + l, whose sum lives in lane e / NV (component e % NV) -- fetched with a shuffle, so
  // every load / store of b, dinv, d, x, y touches whole cache lines (a lane-per-row epilogue
  // writes 8-byte pieces 8 NV bytes apart: measured 2.4x the algorithmic write traffic)
  (void)live;
  (void)row;
  const size_t ebase = (size_t)slice * 64 * NV;
#pragma unroll
  for (int p = 0; p < NV; ++p) {
    const int e = p * 64 + lane;
    const int owner = e / NV, comp = e % NV;
    double val = 0.0;
#pragma unroll
    for (int o = 0; o < NV; ++o) {
      const double t = __shfl(acc[o], owner, 64);
      if (comp == o) val = t;
    }
    if (slice * 64 + owner >= n_rows) continue;
    const size_t idx = ebase + e;
    int m_ = (a.maskmode != MASK_NONE) ? a.mask[idx] : 0;
    if (m_ == 2 && a.ghost == 2) m_ = 0;
    const bool m = m_ != 0;
    if (m_ == 2) {
      if (EPI == EPI_CHEB) {
        a.d[idx] = 0.0;
        a.y[idx] = a.ghost == 1 ? x[idx] : 0.0;
      } else {
        a.y[idx] = 0.0;
        if (EPI == EPI_STORE && a.y2) a.d[idx] = a.y2[idx] = 0.0;
      }
    } else if (EPI == EPI_RESID) {
      if (m) val = (a.maskmode == MASK_IDENTITY) ? a.b[idx] - x[idx] : 0.0;
      else val = a.b[idx] - val;
      a.y[idx] = val;
    } else if (EPI == EPI_ACCUM) {
      if (!m) a.y[idx] += a.c2 * val;
      else if (a.maskmode == MASK_ZERO) a.y[idx] = 0.0;
    } else if (EPI == EPI_CHEB) {
      double dn = 0.0, xn = 0.0;
      if (!m) {
        dn = a.c2 * a.dinv[idx] * (a.b[idx] - val);
        if (a.c1 != 0.0) dn += a.c1 * a.d[idx];
        xn = x[idx] + dn;
      } else if (a.ident) {
        xn = a.b[idx];
      }
      a.d[idx] = dn;
      a.y[idx] = xn;
    } else {
      if (m) val = (a.maskmode == MASK_IDENTITY) ? x[idx] : 0.0;
      else val *= a.c2;
      a.y[idx] = val;
      if (a.y2) {
        const double dn = m ? 0.0 : a.c1 * a.dinv[idx] * val;
        a.d[idx] = dn;
        a.y2[idx] = dn;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_sell_fill(int64_t len, const int32_t* __restrict__ src,
                                                   const double* __restrict__ csr,
                                                   double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int sidx = src[i];
    out[i] = sidx >= 0 ? csr[sidx] : 0.0;
  }
}

// SELL-64 layout of a pattern (host, once): kept only when the zero padding stays below 10 % of
// the nonzeros, i.e. when consecutive rows have (nearly) equal length -- the parity-class
// numberings of structured meshes and every P1 lattice operator
void build_sell(Pattern& p, hipStream_t s) {
  p.n_slices = 0;
  p.sell_len = 0;
  static const int enabled = [] {
    // measured (round 2, MI355X): on the parity-class numbering SELL-64 beats the CSR-stream kernel
    // by 6 % in 3D (261 vs 278 us in situ at n = 64: the whole step 26.4 vs 28.3 ms) and loses 15 %
    // in 2D (68 vs 59 us cache-cold at n = 512).  Default (1): patterns with >= 20 entries per row
    // on average, i.e. the tetrahedral P2 operators; 2: every pattern that qualifies; 0: never
    const char* e = std::getenv("NSFEM_SELL");
    return e ? std::atoi(e) : 1;
  }();
  if (!enabled || p.n_rows < 2048 || p.h_rowptr.empty()) return;
  if (enabled == 1 && (double)p.nnz < 20.0 * p.n_rows) return;
  const int ns = (p.n_rows + 63) / 64;
  std::vector<int32_t> ptr((size_t)ns + 1, 0);
  int64_t total = 0;
  for (int sl = 0; sl < ns; ++sl) {
    int w = 0;
    for (int r = sl * 64; r < std::min(p.n_rows, sl * 64 + 64); ++r)
      w = std::max(w, p.h_rowptr[r + 1] - p.h_rowptr[r]);
    total += (int64_t)w * 64;
    if (total > 0x7fffff00LL) return;                       // 32-bit entry offsets
    ptr[sl + 1] = (int32_t)total;
  }
  if ((double)total > 1.10 * (double)p.nnz) return;
  std::vector<int32_t> col((size_t)total), src((size_t)total);
  for (int sl = 0; sl < ns; ++sl) {
    const int w = (ptr[sl + 1] - ptr[sl]) / 64;
    for (int lane = 0; lane < 64; ++lane) {
      const int r = sl * 64 + lane;
      const int s0 = r < p.n_rows ? p.h_rowptr[r] : 0, len = r < p.n_rows ? p.h_rowptr[r + 1] - s0 : 0;
      // padding entries reference a column the row (or, past the end, the last row) reads anyway
      const int safe = r < p.n_rows ? (len > 0 ? p.h_col[s0] : 0) : 0;
      for (int k = 0; k < w; ++k) {
        const size_t pos = (size_t)ptr[sl] + (size_t)k * 64 + lane;
        col[pos] = k < len ? p.h_col[s0 + k] : safe;
        src[pos] = k < len ? s0 + k : -1;
      }
    }
  }
  p.sell_ptr.upload(ptr, s);
  p.sell_col.upload(col, s);
  p.sell_src.upload(src, s);
  p.n_slices = ns;
  p.sell_len = total;
  p.sell_w0 = p.sell_w1 = 0;
  // nonzero-balanced contiguous workgroup ranges for the 8 XCDs
  const int nwg = (ns + 3) / 4;
  p.sell_xcd[0] = 0;
  int wgi = 0;
  for (int x = 1; x <= 8; ++x) {
    const int64_t target = total * x / 8;
    while (wgi < nwg && (int64_t)ptr[std::min(ns, (wgi + 1) * 4)] <= target) ++wgi;
    p.sell_xcd[x] = x == 8 ? nwg : wgi;
  }
}

// ---------------------------------------------------------------- stencil dictionary
// One row per lane; lane l of wave w handles row 64 w + l.  A workgroup (256 consecutive rows) uses
// only a handful of dictionary entries (lattice numberings: 2 to ~12): they are copied into LDS
// first, every row then needs ONE byte (its entry's position in the workgroup's list).  Per
// nonzero: offset + value from LDS (lanes of the same class read the same address: broadcast) and
// the NV-wide gather of x at row + offset (consecutive lanes, consecutive nodes: coalesced).  The
// vector operands of the epilogue are requested BEFORE the gather loop -- one row per lane means
// one latency chain per wave, so everything independent is put in flight at once.  No matrix
// stream at all: a launch moves 1 byte per row plus the vectors.  Epilogue as in the SELL kernel
// (shuffle-coalesced).
constexpr int kDictLocal = 32;       // dictionary entries a workgroup may use
template <int NV, int EPI, int U>
__device__ __forceinline__ void spmv_dict_body(int n_rows, int n_wg, const uint8_t* __restrict__ lid,
                                               const int32_t* __restrict__ wg_ptr,
                                               const int32_t* __restrict__ wg_list,
                                               const int32_t* __restrict__ slen,
                                               const int32_t* __restrict__ soff,
                                               const double* __restrict__ sval,
                                               const int32_t* __restrict__ sdpos, int lmax,
                                               int max_local, const SpmvArgs& a) {
  extern __shared__ double sh_dict[];
  double* __restrict__ lv = sh_dict;                                   // [max_local * lmax] values
  double* __restrict__ ldi = sh_dict + (size_t)max_local * lmax;       // [max_local] 1 / diagonal
  int* __restrict__ lo = reinterpret_cast<int*>(ldi + max_local);      // offsets
  int* __restrict__ ll = lo + max_local * lmax;                        // lengths
  // XCD x (workgroups b = x mod 8) walks its own contiguous row range
  const int per = gridDim.x >> 3;
  int wg = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (wg >= n_wg) return;                                              // (whole workgroup)
  if (wg >= a.skip0) wg += a.skipn;            // partitioned product split into interior / halo-adjacent groups
  const int l0 = wg_ptr[wg], nl = wg_ptr[wg + 1] - l0;
  for (int t = threadIdx.x; t < nl * lmax; t += 256) {
    const int j = t / lmax, k = t - j * lmax;
    const size_t g = (size_t)wg_list[l0 + j] * lmax + k;
    lv[t] = sval[g];
    lo[t] = soff[g];
  }
  if ((int)threadIdx.x < nl) {
    const int sg = wg_list[l0 + threadIdx.x];
    ll[threadIdx.x] = slen[sg];
    if (EPI == EPI_CHEB) {
      // Jacobi scaling of the smoother: 1 / diagonal of the row's dictionary entry (rows flagged in the
      // mask never use it) -- no dinv vector is streamed
      const int dp = sdpos[sg];
      ldi[threadIdx.x] = dp >= 0 ? 1.0 / sval[(size_t)sg * lmax + dp] : 0.0;
    }
  }
  const int wave = wg * 4 + ((int)threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int row = wave * 64 + lane;
  const bool live = row < n_rows;
  const double* __restrict__ x = a.x;
  const int st = live ? lid[row] : 0;
  // epilogue operands of this lane's NV output entries (see below for the entry <-> lane map)
  const size_t ebase = (size_t)wave * 64 * NV;
  int pm[NV];
  double pb[NV], pdi[NV], pd[NV], px[NV];
#pragma unroll
  for (int p = 0; p < NV; ++p) {
    const int e = p * 64 + lane;
    const bool in = wave * 64 + e / NV < n_rows;
    const size_t idx = ebase + e;
    pm[p] = (in && a.maskmode != MASK_NONE) ? a.mask[idx] : 0;
    pb[p] = (in && EPI != EPI_STORE) ? a.b[idx] : 0.0;
    pdi[p] = 0.0;
    pd[p] = (in && EPI == EPI_CHEB && a.c1 != 0.0) ? a.d[idx] : 0.0;
    px[p] = in ? x[idx] : 0.0;
  }
  __syncthreads();
  double acc[NV];
#pragma unroll
  for (int o = 0; o < NV; ++o) acc[o] = 0.0;
  const int L = live ? ll[st] : 0;
  const int* __restrict__ op = lo + st * lmax;
  const double* __restrict__ vp = lv + st * lmax;
  // (fused node gather: the run's bounds are requested before the stencil loop)
  int g0 = 0, g1 = 0;
  if (EPI == EPI_STORE && a.gbuf && live) {
    g0 = a.gptr[row];
    g1 = a.gptr[row + 1];
  }
  for (int k = 0; k < L; k += U) {
    int c[U];
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = k + u < L ? k + u : L - 1;        // past the end: last entry again, weight 0
      c[u] = row + op[kk];
      const double t = vp[kk];
      v[u] = k + u < L ? t : 0.0;
    }
    double xv[U][NV];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < NV; ++t) xv[u][t] = x[(size_t)c[u] * NV + t];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < NV; ++t) acc[t] += v[u] * xv[u][t];
  }
  if (EPI == EPI_STORE && a.gbuf) {
    // y = A x (stored value, c2 = 1) + the node's element contributions in ascending order: the sums of the
    // separate gather kernel; rows flagged in the mask ignore the accumulator below
    const double* __restrict__ gb = a.gbuf;
    int k = g0;
    for (; k + 2 <= g1; k += 2) {
      double w0[NV], w1[NV];
#pragma unroll
      for (int t = 0; t < NV; ++t) { w0[t] = gb[(size_t)k * NV + t]; w1[t] = gb[(size_t)(k + 1) * NV + t]; }
#pragma unroll
      for (int t = 0; t < NV; ++t) { acc[t] += w0[t]; acc[t] += w1[t]; }
    }
    for (; k < g1; ++k)
#pragma unroll
      for (int t = 0; t < NV; ++t) acc[t] += gb[(size_t)k * NV + t];
  }
  // the wave's 64 rows x NV outputs are the CONTIGUOUS entries 64 NV wave .. of the vectors; in
  // pass p lane l handles entry e = 64 p + l, whose sum lives in lane e / NV (component e % NV)
#pragma unroll
  for (int p = 0; p < NV; ++p) {
    const int e = p * 64 + lane;
    const int owner = e / NV, comp = e % NV;
    double val = 0.0;
#pragma unroll
    for (int o = 0; o < NV; ++o) {
      const double t = __shfl(acc[o], owner, 64);
      if (comp == o) val = t;
    }
    if (EPI == EPI_CHEB) pdi[p] = ldi[__shfl(st, owner, 64)];
    if (wave * 64 + owner >= n_rows) continue;
    const size_t idx = ebase + e;
    int m_ = pm[p];
    if (m_ == 2 && a.ghost == 2) m_ = 0;
    const bool m = m_ != 0;
    if (m_ == 2) {
      if (EPI == EPI_CHEB) {
        a.d[idx] = 0.0;
        a.y[idx] = a.ghost == 1 ? px[p] : 0.0;
      } else {
        a.y[idx] = 0.0;
      }
    } else if (EPI == EPI_RESID) {
      if (m) val = (a.maskmode == MASK_IDENTITY) ? pb[p] - px[p] : 0.0;
      else val = pb[p] - val;
      a.y[idx] = val;
    } else if (EPI == EPI_CHEB) {
      double dn = 0.0, xn = 0.0;
      if (!m) {
        dn = a.c2 * pdi[p] * (pb[p] - val);
        if (a.c1 != 0.0) dn += a.c1 * pd[p];
        xn = px[p] + dn;
      } else if (a.ident) {
        xn = pb[p];
      }
      a.d[idx] = dn;
      a.y[idx] = xn;
    } else {
      if (m) val = (a.maskmode == MASK_IDENTITY) ? px[p] : 0.0;
      else val *= a.c2;
      a.y[idx] = val;
    }
  }
}

// two launch shapes of the same body: 4 waves per SIMD with 4 entries in flight per lane (102
// VGPRs; best for 3 interleaved components: 110 vs 129 us), and 8 waves per SIMD with 2 entries in
// flight (<= 64 VGPRs; best for 1 - 2 components: 26.7 vs 28.3 us at n = 512)
template <int NV, int EPI>
__global__ __launch_bounds__(256) void k_spmv_dict(int n_rows, int n_wg, const uint8_t* __restrict__ lid,
                                                   const int32_t* __restrict__ wg_ptr,
                                                   const int32_t* __restrict__ wg_list,
                                                   const int32_t* __restrict__ slen,
                                                   const int32_t* __restrict__ soff,
                                                   const double* __restrict__ sval,
                                                   const int32_t* __restrict__ sdpos, int lmax,
                                                   int max_local, SpmvArgs a) {
  spmv_dict_body<NV, EPI, 4>(n_rows, n_wg, lid, wg_ptr, wg_list, slen, soff, sval, sdpos, lmax, max_local, a);
}
template <int NV, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_spmv_dict_w8(int n_rows, int n_wg, const uint8_t* __restrict__ lid,
                    const int32_t* __restrict__ wg_ptr, const int32_t* __restrict__ wg_list,
                    const int32_t* __restrict__ slen, const int32_t* __restrict__ soff,
                    const double* __restrict__ sval, const int32_t* __restrict__ sdpos, int lmax,
                    int max_local, SpmvArgs a) {
  spmv_dict_body<NV, EPI, 2>(n_rows, n_wg, lid, wg_ptr, wg_list, slen, soff, sval, sdpos, lmax, max_local, a);
}

// Rectangular block operators between the P2 and P1 numberings (divergence 1 x dim blocks, its
// transpose dim x 1): same scheme, the column of entry k is cbase[row] + offset_k, a table entry
// holds BR * BC values.  EPI_STORE: y = c2 A x; EPI_ACCUM: y += c2 A x on rows not flagged.
template <int BR, int BC, int EPI>
__global__ __launch_bounds__(256) void k_spmv_dict_blk(int n_rows, int n_wg, const uint8_t* __restrict__ lid,
                                                       const int32_t* __restrict__ cbase,
                                                       const int32_t* __restrict__ wg_ptr,
                                                       const int32_t* __restrict__ wg_list,
                                                       const int32_t* __restrict__ slen,
                                                       const int32_t* __restrict__ soff,
                                                       const double* __restrict__ sval, int lmax,
                                                       int max_local, SpmvArgs a) {
  constexpr int BS = BR * BC;
  extern __shared__ double sh_dict[];
  double* __restrict__ lv = sh_dict;                                          // [max_local * lmax * BS]
  int* __restrict__ lo = reinterpret_cast<int*>(sh_dict + (size_t)max_local * lmax * BS);
  int* __restrict__ ll = lo + max_local * lmax;
  const int per = gridDim.x >> 3;
  const int wg = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (wg >= n_wg) return;
  const int l0 = wg_ptr[wg], nl = wg_ptr[wg + 1] - l0;
  for (int t = threadIdx.x; t < nl * lmax; t += 256) {
    const int j = t / lmax, k = t - j * lmax;
    const size_t g = (size_t)wg_list[l0 + j] * lmax + k;
    lo[t] = soff[g];
#pragma unroll
    for (int q = 0; q < BS; ++q) lv[(size_t)t * BS + q] = sval[g * BS + q];
  }
  if ((int)threadIdx.x < nl) ll[threadIdx.x] = slen[wg_list[l0 + threadIdx.x]];
  const int wave = wg * 4 + ((int)threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int row = wave * 64 + lane;
  const bool live = row < n_rows;
  const double* __restrict__ x = a.x;
  const int st = live ? lid[row] : 0;
  const int cb = live ? cbase[row] : 0;
  __syncthreads();
  double acc[BR];
#pragma unroll
  for (int o = 0; o < BR; ++o) acc[o] = 0.0;
  const int L = live ? ll[st] : 0;
  const int* __restrict__ op = lo + st * lmax;
  const double* __restrict__ vp = lv + (size_t)st * lmax * BS;
  constexpr int U = 4;
  for (int k = 0; k < L; k += U) {
    double xv[U][BC];
    int kk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      kk[u] = k + u < L ? k + u : L - 1;
      const int c = cb + op[kk[u]];
#pragma unroll
      for (int t = 0; t < BC; ++t) xv[u][t] = x[(size_t)c * BC + t];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double w = k + u < L ? 1.0 : 0.0;
#pragma unroll
      for (int r = 0; r < BR; ++r)
#pragma unroll
        for (int t = 0; t < BC; ++t) acc[r] += w * vp[(size_t)kk[u] * BS + r * BC + t] * xv[u][t];
    }
  }
  const size_t ebase = (size_t)wave * 64 * BR;
#pragma unroll
  for (int p = 0; p < BR; ++p) {
    const int e = p * 64 + lane;
    const int owner = e / BR, comp = e % BR;
    double val = 0.0;
#pragma unroll
    for (int o = 0; o < BR; ++o) {
      const double t = __shfl(acc[o], owner, 64);
      if (comp == o) val = t;
    }
    if (wave * 64 + owner >= n_rows) continue;
    const size_t idx = ebase + e;
    int m_ = (a.maskmode != MASK_NONE) ? a.mask[idx] : 0;
    if (m_ == 2 && a.ghost == 2) m_ = 0;
    if (m_ == 2) {
      a.y[idx] = 0.0;
    } else if (EPI == EPI_ACCUM) {
      if (!m_) a.y[idx] += a.c2 * val;
      else if (a.maskmode == MASK_ZERO) a.y[idx] = 0.0;
    } else {
      a.y[idx] = m_ ? 0.0 : a.c2 * val;
    }
  }
}

// ------------------------------------------------------------- multi-step lattice smoother
// S Chebyshev-Jacobi steps of a scalar lattice operator (2D lexicographic numbering, stencil
// dictionary with (dj, di) offsets) on NV interleaved components in ONE launch.  A workgroup owns
// an output tile T of TX x TY lattice nodes and keeps the iterate on the extended tile
// E = T (+) G, G = R * Mv (R = stencil reach, Mv = operator applications of the launch), in LDS:
// stage m computes the new iterate on T (+) R (Mv - m) -- the halo shrinks by the reach per stage,
// its nodes are computed redundantly by the neighbouring tiles -- from one LDS buffer into the
// other (one barrier per stage).  b, d and the new iterate of a thread's nodes stay in registers;
// the Jacobi scaling is 1 / diagonal of the row's dictionary entry.  Per launch the vectors are
// read and written once instead of once per step, and the 9 - 19 gathers per row and step come from
// LDS instead of the L1 / L2 path that bounds the one-step kernel (k_spmv_dict).
//
// Layout (what the measurements on the MI355X forced):
//   * E is 64 nodes wide and stored SPLIT BY PARITY CLASS (i & 1, j & 1): four planes of 32-word rows.
//     A wavefront covers two full plane rows, lane l at plane word base + l: its 16-byte reads of a
//     neighbour are 64 consecutive words (the only conflict-free pattern of ds_read_b128), and its
//     64 nodes belong to ONE class, i.e. away from the domain boundary to ONE dictionary entry;
//   * every node of E is owned by one (thread, slot): staging E is the owners' own loads, node
//     coordinates are shifts and masks of the lane / wave / slot ids (no integer division);
//   * all slots of a wave belong to the same class, so the entry's values and LDS offsets are
//     loaded ONCE per wave through the scalar cache (wave-uniform addresses of `const __restrict__`
//     kernel arguments) and stay in SGPRs for all rows of all stages; waves that touch the domain
//     boundary (several entries) fall back to per-lane table loads.
//   from_zero : the start vector is zero -- step 0 is the pointwise  d = x = c2 dinv b  (on E)
//   r_out     : additionally r = b - A x_S on T (pre-smoothing + residual of a V-cycle leg)
//   d_in/out  : Chebyshev direction carried across launches of one smoothing sequence
// Row masks as in the one-step kernel (flag 1: x = d = 0, or x = b on the last step when `ident`);
// partitioned levels (ghost rows) keep the one-step kernels.
struct LatticeArgs {
  int W, H;                 // lattice: row = j * W + i
  int TX, TY, ntx, ntiles;  // output tile (even dimensions), tiles per lattice line, tiles
  int R, S, Mv, G, Ge;      // reach, steps, operator applications, halo = R * Mv, halo rounded up to even
  int EHh;                  // rows of a class plane (= (TY + 2 Ge) / 2); a plane row has 32 words
  int from_zero, ident;
  int lp, n_st;             // table stride (longest stencil rounded up to a multiple of 4), entries
  int dbg;                  // knock-out build only (NSFEM_KNOCKOUTS, wrong results): 1 no stages, 4 no row products
  const double *x_in, *b, *d_in;
  double *x_out, *d_out, *r_out;
  const uint8_t *sid, *mask;
  const uint8_t* sidm;      // optional: per row  entry | (mask of component c) << (6 + c)  in ONE byte
  // transfers of a multigrid cycle fused into the staging (lattice hierarchies: the coarse lattice is the
  // even-even sublattice, odd nodes interpolate their two neighbours along x, y or the (1,1) diagonal):
  const double* xc;         // start vector = [x_in +] P xc  (prolongation of the coarse correction; Wc = (W + 1) / 2)
  const double* rf;         // right-hand side = R rf  (restriction of the finer level's vector, lattice 2 W - 1 wide),
  double* b_out;            //   stored to b_out on the output tile for the later launches of the level
  int Wc, Wf, Hf;
  // partitioned strips, relaxed halo mode: the first gh_lo and the last gh_hi lattice lines are ghost rows -- their
  // iterate is FROZEN through the stages of the launch (block-Jacobi across the ranks, as the one-step kernels do
  // with SpmvArgs::ghost = 1); gh_zero: the launch ends a smoothing sequence, the ghost rows are stored as zeros
  int gh_lo, gh_hi, gh_zero;
  // dictionary entries (bit e) whose stencil has the CANONICAL interior shape of the wave's parity class on a
  // right-diagonal lattice -- 0: none known; the stages of such waves read their neighbours at compile-time LDS
  // offsets (lattice_stages_fixed).  fixed_shape: 1 = P2 operator (19 / 9 / 9 / 9 entries), 2 = P1 (7-point)
  int fixed_shape;
  unsigned long long fixed_mask[4];
  double c1[8], c2[8];     // (up to 7 steps on reach-1 operators: 6 applications after the pointwise first step)
};

// per-slot state of a thread: LDS word, lattice row, packed {ring : 8 | mask bits : 8 | entry : 8}
// (ring = distance to the output tile, 255: no node), b, d and the newest iterate
// (slot q of a thread sits 128 LDS words and 8 lattice lines after slot 0; the newest iterate is not kept
// in registers -- every smoothing stage writes it to LDS, the final store reads it back)
template <int NV, int K, int WPC>
struct LatticeSlots {
  int self0, grow0, gstep;       // LDS word / lattice row of slot 0, lattice-row stride between slots (8 W)
  int info[K];
  double bq[K][NV], dq[K][NV];
  __device__ __forceinline__ int self(int q) const { return self0 + 64 * WPC * q; }
  __device__ __forceinline__ size_t grow(int q) const { return (size_t)(grow0 + q * gstep); }
};
#define LAT_RING(q) ((st.info[q] >> 16) & 255)
#define LAT_MK(q) ((st.info[q] >> 8) & 255)
#define LAT_ST(q) (st.info[q] & 255)

// stages of a wave whose nodes share ONE dictionary entry: its LP values and LDS offsets (zero padded
// tables, LP >= the entry's length) are scalar loads issued once, before the first stage
template <int NV, int K, int LP, int WPC, bool GH>
__device__ __forceinline__ void lattice_stages_uniform(const LatticeArgs& a, LatticeSlots<NV, K, WPC>& st,
                                                       double* __restrict__ xs0, double* __restrict__ xs1,
                                                       const double* __restrict__ vp,
                                                       const int32_t* __restrict__ op, double di) {
  typedef double vec __attribute__((ext_vector_type(2)));
  double v[LP];
  int o[LP];
#pragma unroll
  for (int k = 0; k < LP; ++k) {
    v[k] = vp[k];
    o[k] = op[k];
  }
  const int n_smooth = a.S - a.from_zero;
  for (int m = 1; m <= a.Mv; ++m) {
    const int lim = a.R * (a.Mv - m);
    const bool smoothing = m <= n_smooth;
    const int kstep = m - 1 + a.from_zero;
    const double c1 = smoothing ? a.c1[kstep] : 0.0, c2 = smoothing ? a.c2[kstep] : 0.0;
    const bool last = smoothing && kstep == a.S - 1;
    const double* __restrict__ src = (m & 1) ? xs0 : xs1;
    double* __restrict__ dst = (m & 1) ? xs1 : xs0;
#pragma unroll
    for (int q = 0; q < K; ++q) {
      const bool on = LAT_RING(q) <= lim;
      if (__ballot(on) == 0) continue;               // (wave-uniform)
      if (!on) continue;
      double acc[NV];
#pragma unroll
      for (int c = 0; c < NV; ++c) acc[c] = 0.0;
      if (!NSFEM_KO(a.dbg & 4)) {
        if (NV == 2) {
          const vec* __restrict__ xv = reinterpret_cast<const vec*>(src) + st.self(q);
#pragma unroll
          for (int k0 = 0; k0 < LP; k0 += 4) {
            vec x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = xv[o[k0 + u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              acc[0] += v[k0 + u] * x[u].x;
              acc[NV - 1] += v[k0 + u] * x[u].y;
            }
          }
        } else {
          const double* __restrict__ xv = src + st.self(q);
#pragma unroll
          for (int k0 = 0; k0 < LP; k0 += 4) {
            double x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = xv[o[k0 + u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[0] += v[k0 + u] * x[u];
          }
        }
      }
      if (smoothing) {
#pragma unroll
        for (int c = 0; c < NV; ++c) {
          double dn = 0.0, xn = 0.0;
          if (!((LAT_MK(q) >> c) & 1)) {
            dn = c2 * di * (st.bq[q][c] - acc[c]);
            if (c1 != 0.0) dn += c1 * st.dq[q][c];
            xn = src[(size_t)st.self(q) * NV + c] + dn;
          } else if ((a.ident && last) || (GH && (LAT_MK(q) & 4))) {    // identity row on the last step; ghost row
            xn = st.bq[q][c];                                             // of a strip: frozen (start value in bq)
          }
          st.dq[q][c] = dn;
          dst[(size_t)st.self(q) * NV + c] = xn;
        }
      } else {                          // residual of the smoothed iterate (ring 0 only)
#pragma unroll
        for (int c = 0; c < NV; ++c)
          a.r_out[(size_t)st.grow(q) * NV + c] = ((LAT_MK(q) >> c) & 1) ? 0.0 : st.bq[q][c] - acc[c];
      }
    }
    if (m < a.Mv) __syncthreads();
  }
}

// Stages of a wave whose nodes share ONE dictionary entry of canonical shape: the neighbour reads carry their LDS
// offsets as instruction immediates (ds_read_b128 ... offset:) -- no address arithmetic and no offset table; the
// values are scalar loads issued once, as in lattice_stages_uniform.  
template <int NV, int K, int WPC, bool GH, int SHAPE, int CLS>
__device__ __forceinline__ void lattice_stages_fixed(const LatticeArgs& a, LatticeSlots<NV, K, WPC>& st,
                                                     double* __restrict__ xs0, double* __restrict__ xs1,
                                                     const double* __restrict__ vp, double di) {
  typedef double vec __attribute__((ext_vector_type(2)));
  typedef LatShape<SHAPE, CLS> SH;
  constexpr int N = SH::N, EHH = 2 * K * WPC;
  double v[N];
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = vp[k];
  const int n_smooth = a.S - a.from_zero;
  for (int m = 1; m <= a.Mv; ++m) {
    const int lim = a.R * (a.Mv - m);
    const bool smoothing = m <= n_smooth;
    const int kstep = m - 1 + a.from_zero;
    const double c1 = smoothing ? a.c1[kstep] : 0.0, c2 = smoothing ? a.c2[kstep] : 0.0;
    const bool last = smoothing && kstep == a.S - 1;
    const double* __restrict__ src = (m & 1) ? xs0 : xs1;
    double* __restrict__ dst = (m & 1) ? xs1 : xs0;
#pragma unroll
    for (int q = 0; q < K; ++q) {
      const bool on = LAT_RING(q) <= lim;
      if (__ballot(on) == 0) continue;               // (wave-uniform)
      if (!on) continue;
      double acc[NV];
#pragma unroll
      for (int c = 0; c < NV; ++c) acc[c] = 0.0;
      if (NV == 2) {
        const vec* __restrict__ xv = reinterpret_cast<const vec*>(src) + st.self(q);
#pragma unroll
        for (int k0 = 0; k0 < N; k0 += 4) {
          vec x[4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (k0 + u < N) x[u] = xv[lat_fixed_off(CLS, SH::d[k0 + u < N ? k0 + u : 0][0], SH::d[k0 + u < N ? k0 + u : 0][1], EHH)];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (k0 + u < N) {
              acc[0] += v[k0 + u] * x[u].x;
              acc[NV - 1] += v[k0 + u] * x[u].y;
            }
        }
      } else {
        const double* __restrict__ xv = src + st.self(q);
#pragma unroll
        for (int k0 = 0; k0 < N; k0 += 4) {
          double x[4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (k0 + u < N) x[u] = xv[lat_fixed_off(CLS, SH::d[k0 + u < N ? k0 + u : 0][0], SH::d[k0 + u < N ? k0 + u : 0][1], EHH)];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (k0 + u < N) acc[0] += v[k0 + u] * x[u];
        }
      }
      if (smoothing) {
#pragma unroll
        for (int c = 0; c < NV; ++c) {
          double dn = 0.0, xn = 0.0;
          if (!((LAT_MK(q) >> c) & 1)) {
            dn = c2 * di * (st.bq[q][c] - acc[c]);
            if (c1 != 0.0) dn += c1 * st.dq[q][c];
            xn = src[(size_t)st.self(q) * NV + c] + dn;
          } else if ((a.ident && last) || (GH && (LAT_MK(q) & 4))) {
            xn = st.bq[q][c];
          }
          st.dq[q][c] = dn;
          dst[(size_t)st.self(q) * NV + c] = xn;
        }
      } else {
#pragma unroll
        for (int c = 0; c < NV; ++c)
          a.r_out[(size_t)st.grow(q) * NV + c] = ((LAT_MK(q) >> c) & 1) ? 0.0 : st.bq[q][c] - acc[c];
      }
    }
    if (m < a.Mv) __syncthreads();
  }
}

// the same for a wave whose nodes use several entries (tiles at the domain boundary): per-lane loads
template <int NV, int K, int WPC, bool GH>
__device__ __forceinline__ void lattice_stages_general(const LatticeArgs& a, LatticeSlots<NV, K, WPC>& st,
                                                       double* __restrict__ xs0, double* __restrict__ xs1,
                                                       const double* __restrict__ tval,
                                                       const int32_t* __restrict__ toff,
                                                       const int32_t* __restrict__ tlen,
                                                       const double* __restrict__ tdinv, int cls) {
  typedef double vec __attribute__((ext_vector_type(2)));
  const int n_smooth = a.S - a.from_zero;
  for (int m = 1; m <= a.Mv; ++m) {
    const int lim = a.R * (a.Mv - m);
    const bool smoothing = m <= n_smooth;
    const int kstep = m - 1 + a.from_zero;
    const double c1 = smoothing ? a.c1[kstep] : 0.0, c2 = smoothing ? a.c2[kstep] : 0.0;
    const bool last = smoothing && kstep == a.S - 1;
    const double* __restrict__ src = (m & 1) ? xs0 : xs1;
    double* __restrict__ dst = (m & 1) ? xs1 : xs0;
#pragma unroll
    for (int q = 0; q < K; ++q) {
      if (LAT_RING(q) > lim) continue;
      const int sq = LAT_ST(q);
      const int L = tlen[sq];
      const double* __restrict__ vp = tval + (size_t)sq * a.lp;
      const int32_t* __restrict__ op = toff + ((size_t)sq * 4 + cls) * a.lp;
      const double di = tdinv[sq];
      double acc[NV];
#pragma unroll
      for (int c = 0; c < NV; ++c) acc[c] = 0.0;
      for (int k0 = 0; k0 < L && !NSFEM_KO(a.dbg & 4); k0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double v = vp[k0 + u];
          const int o = op[k0 + u];
          if (NV == 2) {
            const vec xv = reinterpret_cast<const vec*>(src)[st.self(q) + o];
            acc[0] += v * xv.x;
            acc[NV - 1] += v * xv.y;
          } else {
            acc[0] += v * src[st.self(q) + o];
          }
        }
      }
      if (smoothing) {
#pragma unroll
        for (int c = 0; c < NV; ++c) {
          double dn = 0.0, xn = 0.0;
          if (!((LAT_MK(q) >> c) & 1)) {
            dn = c2 * di * (st.bq[q][c] - acc[c]);
            if (c1 != 0.0) dn += c1 * st.dq[q][c];
            xn = src[(size_t)st.self(q) * NV + c] + dn;
          } else if ((a.ident && last) || (GH && (LAT_MK(q) & 4))) {    // identity row on the last step; ghost row
            xn = st.bq[q][c];                                             // of a strip: frozen (start value in bq)
          }
          st.dq[q][c] = dn;
          dst[(size_t)st.self(q) * NV + c] = xn;
        }
      } else {
#pragma unroll
        for (int c = 0; c < NV; ++c)
          a.r_out[(size_t)st.grow(q) * NV + c] = ((LAT_MK(q) >> c) & 1) ? 0.0 : st.bq[q][c] - acc[c];
      }
    }
    if (m < a.Mv) __syncthreads();
  }
}

// The dictionary tables come as separate `const __restrict__` kernel arguments: only then may the compiler
// read them through the scalar cache (s_load) at wave-uniform addresses.  tval[st * lp + k] values (0 past
// the stencil's end), toff[(st * 4 + class) * lp + k] LDS offset of the k-th neighbour, tlen[st],
// tdinv[st] = 1 / diagonal.  512 threads: waves 2c, 2c + 1 own class c; slot q of wave w covers the plane
// rows 2 (2 q + (w & 1)) and the next.
template <int NV, int K, int WPE, int WPC, bool GH>
__global__ __launch_bounds__(256 * WPC) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void k_cheb_lattice(LatticeArgs a, const double* __restrict__ tval, const int32_t* __restrict__ toff,
                    const int32_t* __restrict__ tlen, const double* __restrict__ tdinv) {
  extern __shared__ double sh_lat[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD x (workgroups b = x mod 8) walks its own contiguous range of tiles
  const int per = gridDim.x >> 3;
  const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (tile >= a.ntiles) return;                                   // (whole workgroup)
  const int ty = tile / a.ntx, tx = tile - ty * a.ntx;
  const int i0 = tx * a.TX, j0 = ty * a.TY;
  const int i1 = min(i0 + a.TX, a.W), j1 = min(j0 + a.TY, a.H);
  const int ox = i0 - a.Ge, oy = j0 - a.Ge;                       // even: local parity = lattice parity
  const int plane = 32 * a.EHh;
  double* __restrict__ xs0 = sh_lat;                              // [4 planes][EHh][32][NV], two buffers
  double* __restrict__ xs1 = sh_lat + (size_t)4 * plane * NV;
  const int cls = w / WPC, pi = cls & 1, pj = cls >> 1;      // (WPC waves per parity class: 2 or 4)
  const int gi = ox + 2 * (lane & 31) + pi;
  const bool in_x = gi >= 0 && gi < a.W;
  const int ex = max(max(i0 - gi, gi - (i1 - 1)), 0);
  const int Go = a.R * max(a.Mv - 1, 0);
  const int need = (a.from_zero || a.xc) ? a.G : Go;              // nodes whose b / entry / mask are used
  LatticeSlots<NV, K, WPC> st;
  unsigned long long differs = 0;
  int stu = -1;
  // Staging in three passes, so that a wave has ONE memory round trip per pass instead of one per load (loads
  // behind lane-dependent branches are issued and waited for one by one -- 10 to 30 dependent round trips per
  // launch, which is what a launch on the small levels of the hierarchy cost): every load is unconditional at an
  // index clamped into the lattice, the conditions select the loaded values afterwards.
  //   pass 1  rings, clamped node indices, the byte entry | masks of every slot
  //   pass 2  the vector operands of every slot (b or the 7 fine values of R rf; d; x, P xc)
  //   pass 3  masks and conditions applied, slot state filled, start iterate stored to LDS
  int ringq[K], smq[K], gjcq[K];
  size_t gcl[K];
  const int gic = min(max(gi, 0), a.W - 1);
#pragma unroll
  for (int q = 0; q < K; ++q) {
    const int pr = 2 * (WPC * q + (w & (WPC - 1))) + (lane >> 5);
    const int gj = oy + 2 * pr + pj;
    const bool in = in_x && gj >= 0 && gj < a.H;
    const int ey = max(max(j0 - gj, gj - (j1 - 1)), 0);
    ringq[q] = in ? min(max(ex, ey), 255) : 255;
    if (q == 0) {
      st.self0 = cls * plane + pr * 32 + (lane & 31);
      st.grow0 = gj * a.W + gi;
      st.gstep = 4 * WPC * a.W;
    }
    const int gjc = min(max(gj, 0), a.H - 1);
    gjcq[q] = gjc;
    gcl[q] = (size_t)gjc * a.W + gic;
    smq[q] = a.sidm[gcl[q]];
  }
  double braw[K][NV], draw[K][NV], xraw[K][NV];
  // (one loop per kernel-uniform case: the loads of all slots sit in one basic block and are issued together)
  if (a.rf) {
#pragma unroll
    for (int q = 0; q < K; ++q) {
      // b = R rf: the node's own fine value + half of its six fine neighbours (ascending fine index, the
      // order of the CSR row of R = P^T); absent neighbours: the own value with weight 0
      const int fi = 2 * gic, fj = 2 * gjcq[q];
      const size_t fb = (size_t)fj * a.Wf + fi;
      const bool l_ = fi > 0, r_ = fi < a.Wf - 1, d_ = fj > 0, u_ = fj < a.Hf - 1;
      const size_t n0 = (d_ && l_) ? fb - a.Wf - 1 : fb, n1 = d_ ? fb - a.Wf : fb, n2 = l_ ? fb - 1 : fb;
      const size_t n4 = r_ ? fb + 1 : fb, n5 = u_ ? fb + a.Wf : fb, n6 = (u_ && r_) ? fb + a.Wf + 1 : fb;
      const double w0 = (d_ && l_) ? 0.5 : 0.0, w1 = d_ ? 0.5 : 0.0, w2 = l_ ? 0.5 : 0.0;
      const double w4 = r_ ? 0.5 : 0.0, w5 = u_ ? 0.5 : 0.0, w6 = (u_ && r_) ? 0.5 : 0.0;
      double f[7][NV];
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        f[0][c] = a.rf[n0 * NV + c]; f[1][c] = a.rf[n1 * NV + c]; f[2][c] = a.rf[n2 * NV + c];
        f[3][c] = a.rf[fb * NV + c];
        f[4][c] = a.rf[n4 * NV + c]; f[5][c] = a.rf[n5 * NV + c]; f[6][c] = a.rf[n6 * NV + c];
      }
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        double v = 0.0;
        v += w0 * f[0][c];
        v += w1 * f[1][c];
        v += w2 * f[2][c];
        v += f[3][c];
        v += w4 * f[4][c];
        v += w5 * f[5][c];
        v += w6 * f[6][c];
        braw[q][c] = v;
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < K; ++q)
#pragma unroll
      for (int c = 0; c < NV; ++c) braw[q][c] = a.b[gcl[q] * NV + c];
  }
  if (!a.from_zero && a.d_in) {
#pragma unroll
    for (int q = 0; q < K; ++q)
#pragma unroll
      for (int c = 0; c < NV; ++c) draw[q][c] = a.d_in[gcl[q] * NV + c];
  } else {
#pragma unroll
    for (int q = 0; q < K; ++q)
#pragma unroll
      for (int c = 0; c < NV; ++c) draw[q][c] = 0.0;
  }
  if (!a.from_zero && a.xc) {
    // start vector = [x_in +] P xc: even-even nodes copy their coarse node, the others average the two
    // coarse nodes (I, J) and (I + pi, J + pj)
    double xin[K][NV];
    if (a.x_in) {
#pragma unroll
      for (int q = 0; q < K; ++q)
#pragma unroll
        for (int c = 0; c < NV; ++c) xin[q][c] = a.x_in[gcl[q] * NV + c];
    }
    const size_t cmax = (size_t)a.Wc * ((a.H + 1) >> 1) - 1;
    const double pw0 = cls != 0 ? 0.5 : 1.0, pw1 = cls != 0 ? 0.5 : 0.0;
#pragma unroll
    for (int q = 0; q < K; ++q) {
      const size_t c0 = (size_t)(gjcq[q] >> 1) * a.Wc + (gic >> 1);
      const size_t c1 = min(c0 + pi + (size_t)pj * a.Wc, cmax);
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        const double v0 = a.xc[c0 * NV + c], v1 = a.xc[c1 * NV + c];
        xraw[q][c] = pw0 * v0 + pw1 * v1;            // (1 v0 + 0 v1 on the even-even class: the copy, exactly)
      }
    }
    if (a.x_in) {
#pragma unroll
      for (int q = 0; q < K; ++q)
#pragma unroll
        for (int c = 0; c < NV; ++c) xraw[q][c] = xin[q][c] + xraw[q][c];
    }
  } else if (!a.from_zero) {
#pragma unroll
    for (int q = 0; q < K; ++q)
#pragma unroll
      for (int c = 0; c < NV; ++c) xraw[q][c] = a.x_in[gcl[q] * NV + c];
  } else {
#pragma unroll
    for (int q = 0; q < K; ++q)
#pragma unroll
      for (int c = 0; c < NV; ++c) xraw[q][c] = 0.0;
  }
  double diq[K];
#pragma unroll
  for (int q = 0; q < K; ++q) diq[q] = a.from_zero ? tdinv[smq[q] & 63] : 0.0;
#pragma unroll
  for (int q = 0; q < K; ++q) {
    const int ring = ringq[q];
    const bool counts = ring <= need;
    const int stq = counts ? (smq[q] & 63) : 0, mkq = counts ? ((smq[q] >> 6) | ((GH && ((smq[q] >> 6) & 1) && (gjcq[q] < a.gh_lo || gjcq[q] >= a.H - a.gh_hi)) ? 4 : 0)) : 0;
    // (bit 2: ghost row of a partitioned strip -- whole lattice lines at the bottom / top, flagged in the mask as well)
    double xv[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const bool mk = (mkq >> c) & 1;
      double bv = counts ? braw[q][c] : 0.0;
      if (a.rf) {
        if (mk) bv = 0.0;
        if (counts && ring == 0) a.b_out[st.grow(q) * NV + c] = bv;
      }
      st.dq[q][c] = (counts && !a.from_zero && a.d_in && ring <= Go) ? draw[q][c] : 0.0;
      double v = 0.0;
      if (!a.from_zero && ring <= a.G) v = (a.xc && mk) ? 0.0 : xraw[q][c];
      xv[c] = v;
      st.bq[q][c] = (GH && (mkq & 4)) ? v : bv;        // (ghost rows carry their frozen iterate in place of b)
    }
    st.info[q] = (ring << 16) | (mkq << 8) | stq;
    // do the wave's nodes share their dictionary entry?
    const unsigned long long who = __ballot(counts);
    if (who != 0) {
      if (stu < 0) stu = __builtin_amdgcn_readfirstlane(__shfl(stq, __ffsll((long long)who) - 1, 64));
      differs |= __ballot(counts && stq != stu);
    }
    if (a.from_zero) {                    // step 0 from a zero start: pointwise
      const double di = counts ? diq[q] : 0.0;
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        double v = 0.0;
        if (counts) {
          if (!((mkq >> c) & 1)) v = a.c2[0] * di * st.bq[q][c];
          else if (a.ident && a.S == 1 && !(GH && (mkq & 4))) v = st.bq[q][c];
        }
        xv[c] = v;
        st.dq[q][c] = (mkq >> c) & 1 ? 0.0 : v;
      }
    }
#pragma unroll
    for (int c = 0; c < NV; ++c) xs0[(size_t)st.self(q) * NV + c] = xv[c];
  }
  __syncthreads();                       // E is staged
  if (!NSFEM_KO(a.dbg & 1)) {
    if (differs == 0 && stu >= 0) {
      const int L = tlen[stu];
      const double* __restrict__ vp = tval + (size_t)stu * a.lp;
      const int32_t* __restrict__ op = toff + ((size_t)stu * 4 + cls) * a.lp;
      const double di = tdinv[stu];
      bool done = false;
      if (WPE == 4 && a.fixed_shape != 0 && ((a.fixed_mask[cls] >> stu) & 1ull)) {
        // (cls is wave-uniform: one of the eight instantiations runs)
        done = true;
        switch (a.fixed_shape * 4 + cls) {
          case 4: lattice_stages_fixed<NV, K, WPC, GH, 1, 0>(a, st, xs0, xs1, vp, di); break;
          case 5: lattice_stages_fixed<NV, K, WPC, GH, 1, 1>(a, st, xs0, xs1, vp, di); break;
          case 6: lattice_stages_fixed<NV, K, WPC, GH, 1, 2>(a, st, xs0, xs1, vp, di); break;
          case 7: lattice_stages_fixed<NV, K, WPC, GH, 1, 3>(a, st, xs0, xs1, vp, di); break;
          case 8: lattice_stages_fixed<NV, K, WPC, GH, 2, 0>(a, st, xs0, xs1, vp, di); break;
          case 9: lattice_stages_fixed<NV, K, WPC, GH, 2, 1>(a, st, xs0, xs1, vp, di); break;
          case 10: lattice_stages_fixed<NV, K, WPC, GH, 2, 2>(a, st, xs0, xs1, vp, di); break;
          case 11: lattice_stages_fixed<NV, K, WPC, GH, 2, 3>(a, st, xs0, xs1, vp, di); break;
          default: done = false;
        }
      }
      if (done) {
      } else
      if (L <= 8 && a.lp >= 8) lattice_stages_uniform<NV, K, 8, WPC, GH>(a, st, xs0, xs1, vp, op, di);
      else if (L <= 12 && a.lp >= 12) lattice_stages_uniform<NV, K, 12, WPC, GH>(a, st, xs0, xs1, vp, op, di);
      else if (L <= 20 && a.lp >= 20) lattice_stages_uniform<NV, K, 20, WPC, GH>(a, st, xs0, xs1, vp, op, di);
      else lattice_stages_general<NV, K, WPC, GH>(a, st, xs0, xs1, tval, toff, tlen, tdinv, cls);
    } else {
      lattice_stages_general<NV, K, WPC, GH>(a, st, xs0, xs1, tval, toff, tlen, tdinv, cls);
    }
  }
  // the newest iterate sits in the buffer the last smoothing stage wrote (stage m writes buffer m & 1)
  const int n_smooth = NSFEM_KO(a.dbg & 1) ? 0 : a.S - a.from_zero;
  const double* __restrict__ fin = (n_smooth & 1) ? xs1 : xs0;
#pragma unroll
  for (int q = 0; q < K; ++q)
    if (LAT_RING(q) == 0) {
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        a.x_out[st.grow(q) * NV + c] = (GH && a.gh_zero && (LAT_MK(q) & 4)) ? 0.0 : fin[(size_t)st.self(q) * NV + c];
        if (a.d_out) a.d_out[st.grow(q) * NV + c] = st.dq[q][c];
      }
    }
}
#undef LAT_RING
#undef LAT_MK
#undef LAT_ST

// can smoothing steps of this operator run in the lattice kernel?
// tuning / test switches, re-read whenever a context is created (tests switch them between contexts)
static bool g_lattice_on = true, g_lattice_transfers_on = true;
void refresh_env_switches() {
  const char* e = std::getenv("NSFEM_LATTICE");
  g_lattice_on = e ? std::atoi(e) != 0 : true;
  e = std::getenv("NSFEM_LATTICE_TRANSFERS");
  g_lattice_transfers_on = e ? std::atoi(e) != 0 : true;
}
bool lattice_transfers_enabled() { return g_lattice_transfers_on; }
bool lattice_smoother_available(const BlockMat& A, int nv) {
  const bool on = g_lattice_on;
  return on && A.dict_ready && A.dict && A.dict->lat_w >= 8 && A.br == 1 && A.bc == 1 && (nv == 1 || nv == 2);
}
bool lattice_tables_available(const BlockMat& A, int nv) {
  return g_lattice_on && A.dict && A.dict->n_stencils > 0 && A.dict->n_stencils <= 64 && A.dict->lat_w > 0 &&
         A.br == 1 && A.bc == 1 && (nv == 1 || nv == 2) && A.lat_vals.p != nullptr && A.dict_dinv.p != nullptr &&
         (A.dict_ready || A.dict->tables_only);
}
int lattice_smoother_max_steps(const BlockMat& A, bool from_zero, bool with_resid) {
  // operator applications per launch: at most 3 (reach 2: halo 6) / 6 (reach 1: the same halo) -- the truncated
  // velocity cycle's coarse solve (7 steps from zero on a P1 level) is ONE launch
  const int mv_max = A.dict->lat_r >= 2 ? 3 : 6;
  return std::min(A.dict->lat_r >= 2 ? 4 : 7, mv_max + (from_zero ? 1 : 0) - (with_resid ? 1 : 0));
}

// LDS offsets of the dictionary's neighbours in the class-split tile of plane dimensions (ewh, ehh):
// the neighbour (dj, di) of a node of class (pi, pj) lies in the plane of class ((pi + di) & 1,
// (pj + dj) & 1) at the plane position shifted by (floor((pj + dj) / 2), floor((pi + di) / 2))
static const int32_t* lattice_offsets(hipStream_t s, const StencilDict& d, int ewh, int ehh) {
  for (auto& e : d.loff_cache)
    if (e.ewh == ewh && e.ehh == ehh) return e.buf.p;
  const int plane = ewh * ehh, lp = (d.lmax + 3) & ~3;
  std::vector<int32_t> t((size_t)d.n_stencils * 4 * lp, 0);
  auto fl2 = [](int v) { return v >= 0 ? v / 2 : -((1 - v) / 2); };
  for (int st = 0; st < d.n_stencils; ++st)
    for (int c = 0; c < 4; ++c) {
      const int pi = c & 1, pj = (c >> 1) & 1;
      for (int k = 0; k < d.h_len[st]; ++k) {
        const int q = d.h_pack[(size_t)st * d.lmax + k], dj = (q >> 5) - 8, di = (q & 31) - 8;
        const int c2 = ((pi + di) & 1) | (((pj + dj) & 1) << 1);
        t[((size_t)st * 4 + c) * lp + k] = (c2 - c) * plane + fl2(pj + dj) * ewh + fl2(pi + di);
      }
    }
  d.loff_cache.emplace_back();
  StencilDict::LatticeOffsets& e = d.loff_cache.back();
  e.ewh = ewh;
  e.ehh = ehh;
  e.buf.upload(t, s);
  return e.buf.p;
}

// which dictionary entries have the canonical interior stencil shape of a parity class (StencilDict::fixed_mask)
void ensure_fixed_masks(const StencilDict& d) {
  if (d.fixed_shape != 0) return;
  d.fixed_shape = -1;
  const int shape = d.lat_r == 2 ? 1 : (d.lat_r == 1 ? 2 : 0);
  if (shape == 0 || d.lat_w <= 0 || (int)d.h_len.size() != d.n_stencils) return;
  bool any = false;
  for (int c = 0; c < 4; ++c) {
    d.fixed_mask[c] = 0;
    int n = 0;
    const int (*sh)[2] = nullptr;
    if (shape == 1) {
      if (c == 0) { n = LatShape<1, 0>::N; sh = LatShape<1, 0>::d; }
      else if (c == 1) { n = LatShape<1, 1>::N; sh = LatShape<1, 1>::d; }
      else if (c == 2) { n = LatShape<1, 2>::N; sh = LatShape<1, 2>::d; }
      else { n = LatShape<1, 3>::N; sh = LatShape<1, 3>::d; }
    } else { n = LatShape<2, 0>::N; sh = LatShape<2, 0>::d; }
    for (int e = 0; e < d.n_stencils && e < 64; ++e) {
      if (d.h_len[e] != n) continue;
      bool same = true;
      for (int k = 0; k < n && same; ++k)
        same = d.h_pack[(size_t)e * d.lmax + k] == (sh[k][0] + 8) * 32 + (sh[k][1] + 8);
      if (same) { d.fixed_mask[c] |= 1ull << e; any = true; }
    }
  }
  if (any) d.fixed_shape = shape;
}

// `steps` (<= lattice_smoother_max_steps) Chebyshev-Jacobi steps with the coefficients c1[k], c2[k]
// supplied by the caller; x_in == nullptr: zero start
void launch_cheb_lattice(hipStream_t s, const BlockMat& A, int nv, const double* x_in, const double* b,
                         const double* d_in, double* x_out, double* d_out, double* r_out,
                         const uint8_t* mask, int steps, const double* c1, const double* c2, int ident,
                         const uint8_t* sidm, const double* xc, const double* rf, double* b_out, int gh_lo, int gh_hi,
                         int gh_zero) {
  const StencilDict& d = *A.dict;
  NSFEM_REQUIRE(steps >= 1 && steps <= 8, "lattice smoother: 1..8 steps per launch");
  NSFEM_REQUIRE(gh_lo >= 0 && gh_hi >= 0 && gh_lo + gh_hi < d.lat_h && !((gh_lo || gh_hi) && (xc || rf)),
                "lattice smoother: bad ghost lines (strips run without fused transfers)");
  NSFEM_REQUIRE(x_out != x_in, "lattice smoother works out of place");
  NSFEM_REQUIRE(!rf || b_out, "fused restriction needs a place to keep the right-hand side");
  LatticeArgs a;
  a.W = d.lat_w; a.H = d.lat_h; a.R = d.lat_r; a.S = steps;
  a.from_zero = (x_in || xc) ? 0 : 1;
  a.xc = xc; a.rf = rf; a.b_out = b_out;
  a.gh_lo = gh_lo; a.gh_hi = gh_hi; a.gh_zero = gh_zero;
  // entries of canonical interior shape (compile-time LDS offsets in the stages; NSFEM_LATTICE_FIXED=0: off)
  ensure_fixed_masks(d);
  static const bool fixed_on = [] { const char* e = std::getenv("NSFEM_LATTICE_FIXED"); return e ? std::atoi(e) != 0 : true; }();
  a.fixed_shape = (fixed_on && d.fixed_shape > 0) ? d.fixed_shape : 0;
  for (int c = 0; c < 4; ++c) a.fixed_mask[c] = a.fixed_shape ? d.fixed_mask[c] : 0ull;
  a.Wc = (d.lat_w + 1) / 2; a.Wf = 2 * d.lat_w - 1; a.Hf = 2 * d.lat_h - 1;
  a.Mv = steps - a.from_zero + (r_out ? 1 : 0);
  NSFEM_REQUIRE(a.Mv >= 0 && a.Mv * a.R <= 8, "lattice smoother: halo too wide");
  a.G = a.R * a.Mv;
  a.Ge = (a.G + 1) & ~1;
  // extended tile: 64 nodes wide, 32 / 24 / 16 lines high (4 / 3 / 2 slots per thread); the output tile is
  // what the halo leaves, balanced over the lattice so that the last tile of a line is not a sliver
  const int64_t nn = (int64_t)a.W * a.H;
  // (measured, 2D cavity n = 512, 3 steps on the P2 lattice: 24 lines 43 us warm / 49 us cold, 32 lines
  // 48 / 56 us -- 21 spilled registers under the 128-VGPR cap and 1040 tiles on 512 workgroup slots)
  static const int64_t eh24_from = [] { const char* e = std::getenv("NSFEM_LATTICE_EH24_FROM"); return e ? std::atoll(e) : 20000ll; }();
  int eh = nn >= eh24_from ? 24 : 16;
  static const int force_eh = [] { const char* e = std::getenv("NSFEM_LATTICE_EH"); return e ? std::atoi(e) : 0; }();
  // 48 lines: ONE workgroup of 1024 threads per CU (four waves per parity class, the same three slots per thread and
  // the same 4 waves per SIMD as two 24-line workgroups): the halo is shared by twice the output lines -- 1.64
  // instead of 2.46 staged nodes per output node at a 6-line halo, 1.20 instead of 1.45 row-steps per useful one.
  // Measured (round 4): n = 512 (1.05 M nodes) 1.94 vs 1.925 ms/step with 24 lines, n = 1024 (4.2 M nodes) 7.21 vs
  // 7.29 -- the launch is bound by the latency chain of a workgroup, not by the halo; 16-wave barriers cost what the
  // halo saves.  On from 2 M lattice nodes.
  static const int tall_from = [] { const char* e = std::getenv("NSFEM_LATTICE_TALL_FROM"); return e ? std::atoi(e) : 2000000; }();
  if (nn >= tall_from && a.H >= 96) eh = 48;
  if (force_eh == 16 || force_eh == 24 || force_eh == 32 || force_eh == 48) eh = force_eh;
  while (eh < 32 && eh - 2 * a.Ge < 8) eh += 8;
  const int tmx = 64 - 2 * a.Ge, tmy = eh - 2 * a.Ge;
  NSFEM_REQUIRE(tmx >= 2 && tmy >= 2, "lattice smoother: halo too wide for the tile");
  const int ncx = (a.W + tmx - 1) / tmx, ncy = (a.H + tmy - 1) / tmy;
  a.TX = std::min(tmx, (((a.W + ncx - 1) / ncx) + 1) & ~1);
  a.TY = std::min(tmy, (((a.H + ncy - 1) / ncy) + 1) & ~1);
  a.ntx = (a.W + a.TX - 1) / a.TX;
  a.ntiles = a.ntx * ((a.H + a.TY - 1) / a.TY);
  a.EHh = eh / 2;
  a.ident = ident;
  a.x_in = x_in; a.b = b; a.d_in = d_in; a.x_out = x_out; a.d_out = d_out; a.r_out = r_out;
  a.sid = d.sid8.p; a.mask = mask;
  // entry | masks as ONE byte per row: callers outside the multigrid (test hook, timing) get it built here
  NSFEM_REQUIRE(d.n_stencils <= 64 && nv <= 2, "the lattice kernel needs <= 64 dictionary entries and <= 2 components");
  if (!sidm) {
    // (per dictionary: contexts on other devices or streams bring their own operator, hence their own buffer)
    if (d.sidm_scratch.n != (size_t)d.n_rows) d.sidm_scratch.alloc((size_t)d.n_rows);
    launch_lattice_sidm(s, A, nv, mask, d.sidm_scratch.p);
    sidm = d.sidm_scratch.p;
  }
  a.sidm = sidm;
  const int32_t* toff = lattice_offsets(s, d, 32, a.EHh);
  a.lp = (d.lmax + 3) & ~3;
  a.n_st = d.n_stencils;
  NSFEM_REQUIRE(A.lat_vals.n == (size_t)d.n_stencils * a.lp, "lattice value table missing");
#if NSFEM_KNOCKOUTS
  static const int dbg = [] { const char* e = std::getenv("NSFEM_LATTICE_DBG"); return e ? std::atoi(e) : 0; }();
  a.dbg = dbg;
#else
  a.dbg = 0;
#endif
  for (int k = 0; k < 8; ++k) { a.c1[k] = k < steps ? c1[k] : 0.0; a.c2[k] = k < steps ? c2[k] : 0.0; }
  const size_t lds = (size_t)2 * 4 * 32 * a.EHh * nv * 8;
  const int grid = (a.ntiles + 7) & ~7;
  // launch shape (tuning switch NSFEM_LATTICE_SHAPE): 0 = 4 waves per SIMD (<= 128 VGPRs: two workgroups per
  // CU), 1 = 2 waves per SIMD (no register cap: one workgroup per CU)
  static const int shape = [] { const char* e = std::getenv("NSFEM_LATTICE_SHAPE"); return e ? std::atoi(e) : 0; }();
#define NSFEM_LAT(NV, KK, WPE, WPC, GH)                                                                \
  do {                                                                                                 \
    static bool attr_dev[64];                                                                          \
    int dev_ = 0;                                                                                      \
    NSFEM_HIP(hipGetDevice(&dev_));                                                                    \
    bool& attr_set = attr_dev[dev_ & 63];                                                              \
    if (!attr_set) {                                                                                   \
      NSFEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cheb_lattice<NV, KK, WPE, WPC, GH>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));         \
      attr_set = true;                                                                                 \
    }                                                                                                  \
    hipLaunchKernelGGL((k_cheb_lattice<NV, KK, WPE, WPC, GH>), dim3(grid), dim3(256 * WPC), lds, s, a, \
                       (const double*)A.lat_vals.p, toff, (const int32_t*)d.len.p,                     \
                       (const double*)A.dict_dinv.p);                                                  \
  } while (0)
#define NSFEM_LAT_K(NV, WPE, GH)                        \
  do {                                                  \
    if (eh == 48) NSFEM_LAT(NV, 3, 4, 4, GH);           \
    else if (eh == 32) NSFEM_LAT(NV, 4, WPE, 2, GH);    \
    else if (eh == 24) NSFEM_LAT(NV, 3, WPE, 2, GH);    \
    else NSFEM_LAT(NV, 2, WPE, 2, GH);                  \
  } while (0)
  // (the frozen ghost lines of partitioned strips are a template flag: the single-context launches do not pay for
  // them -- 811 vs 758 VALU instructions per wave; strips always use the default launch shape)
  const bool gh = gh_lo > 0 || gh_hi > 0;
  if (gh && nv == 2) NSFEM_LAT_K(2, 4, true);
  else if (gh) NSFEM_LAT_K(1, 4, true);
  else if (nv == 2 && shape == 1) NSFEM_LAT_K(2, 2, false);
  else if (nv == 2 && shape == 2) NSFEM_LAT_K(2, 6, false);
  else if (nv == 2) NSFEM_LAT_K(2, 4, false);
  else if (shape == 1) NSFEM_LAT_K(1, 2, false);
  else if (shape == 2) NSFEM_LAT_K(1, 6, false);
  else NSFEM_LAT_K(1, 4, false);
#undef NSFEM_LAT_K
#undef NSFEM_LAT
  NSFEM_HIP(hipGetLastError());
}

// entry | mask bits in one byte per row (the lattice kernel reads one byte instead of 1 + NV)
__global__ __launch_bounds__(256) void k_lattice_sidm(int n, int nv, const uint8_t* __restrict__ sid,
                                                      const uint8_t* __restrict__ mask, uint8_t* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int v = sid[i];
    for (int c = 0; c < nv; ++c)
      if (mask && mask[(size_t)i * nv + c]) v |= 64 << c;
    out[i] = (uint8_t)v;
  }
}
void launch_lattice_sidm(hipStream_t s, const BlockMat& A, int nv, const uint8_t* mask, uint8_t* out) {
  const StencilDict& d = *A.dict;
  NSFEM_REQUIRE(d.n_stencils <= 64 && nv <= 2, "combined entry / mask byte needs <= 64 entries and <= 2 components");
  hipLaunchKernelGGL(k_lattice_sidm, dim3(std::min((d.n_rows + 255) / 256, 2048)), dim3(256), 0, s, d.n_rows, nv,
                     d.sid8.p, mask, out);
  NSFEM_HIP(hipGetLastError());
}

// b_c = R r on a lattice hierarchy (coarse lattice = even-even sublattice, R = P^T of the lattice interpolation):
// the coarse node's own fine value + half of its six fine neighbours, summed in ascending fine index (the order of
// the CSR row of R); rows flagged in the coarse mask get 0.  One thread per coarse node, the seven NV-wide loads
// unconditional at clamped indices (absent neighbours: the own value with weight 0).  Replaces the CSR product with
// R on the restriction chain of a cycle without pre-smoothing: no matrix stream, no index loads.
template <int NV>
__global__ __launch_bounds__(256) void k_restrict_lattice(int Wc, int Hc, int Wf, int Hf,
                                                          const double* __restrict__ rf,
                                                          const uint8_t* __restrict__ mask,
                                                          double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Wc * Hc) return;
  const int J = t / Wc, I = t - J * Wc;
  const int fi = 2 * I, fj = 2 * J;
  const size_t fb = (size_t)fj * Wf + fi;
  const bool l_ = fi > 0, r_ = fi < Wf - 1, d_ = fj > 0, u_ = fj < Hf - 1;
  const size_t n0 = (d_ && l_) ? fb - Wf - 1 : fb, n1 = d_ ? fb - Wf : fb, n2 = l_ ? fb - 1 : fb;
  const size_t n4 = r_ ? fb + 1 : fb, n5 = u_ ? fb + Wf : fb, n6 = (u_ && r_) ? fb + Wf + 1 : fb;
  const double w0 = (d_ && l_) ? 0.5 : 0.0, w1 = d_ ? 0.5 : 0.0, w2 = l_ ? 0.5 : 0.0;
  const double w4 = r_ ? 0.5 : 0.0, w5 = u_ ? 0.5 : 0.0, w6 = (u_ && r_) ? 0.5 : 0.0;
  double f[7][NV];
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    f[0][c] = rf[n0 * NV + c]; f[1][c] = rf[n1 * NV + c]; f[2][c] = rf[n2 * NV + c];
    f[3][c] = rf[fb * NV + c];
    f[4][c] = rf[n4 * NV + c]; f[5][c] = rf[n5 * NV + c]; f[6][c] = rf[n6 * NV + c];
  }
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    double v = 0.0;
    v += w0 * f[0][c];
    v += w1 * f[1][c];
    v += w2 * f[2][c];
    v += f[3][c];
    v += w4 * f[4][c];
    v += w5 * f[5][c];
    v += w6 * f[6][c];
    if (mask && mask[(size_t)t * NV + c]) v = 0.0;
    out[(size_t)t * NV + c] = v;
  }
}
bool launch_restrict_lattice(hipStream_t s, int nv, int Wc, int Hc, int Wf, int Hf, const double* rf,
                             const uint8_t* mask, double* out) {
  if (nv < 1 || nv > 2 || Wf != 2 * Wc - 1 || Hf != 2 * Hc - 1) return false;
  const dim3 grid((Wc * Hc + 255) / 256), block(256);
  if (nv == 1) hipLaunchKernelGGL(k_restrict_lattice<1>, grid, block, 0, s, Wc, Hc, Wf, Hf, rf, mask, out);
  else hipLaunchKernelGGL(k_restrict_lattice<2>, grid, block, 0, s, Wc, Hc, Wf, Hf, rf, mask, out);
  NSFEM_HIP(hipGetLastError());
  return true;
}

// Two restrictions of a chain in one launch:  b1 = R1 rf  and  b2 = R2 b1  (both stored: the levels' later smoothing
// launches read them).  A workgroup owns 16 x 16 nodes of the coarsest of the three lattices: it forms the 33 x 33
// nodes of the middle lattice around them in LDS (its own 32 x 32 are stored; the one-node ring is recomputed by the
// neighbours), then its 256 coarse nodes from LDS.  Same sums in the same order as two launches of
// k_restrict_lattice.
template <int NV>
__global__ __launch_bounds__(256) void k_restrict_lattice2(int W2, int H2, int W1, int H1, int Wf, int Hf,
                                                           const double* __restrict__ rf,
                                                           const uint8_t* __restrict__ mask1,
                                                           const uint8_t* __restrict__ mask2,
                                                           double* __restrict__ b1, double* __restrict__ b2) {
  __shared__ double sh[33 * 33 * NV];
  const int ntx = (W2 + 15) >> 4;
  const int ty = blockIdx.x / ntx, tx = blockIdx.x - ty * ntx;
  const int i10 = 32 * tx - 1, j10 = 32 * ty - 1;
  for (int t = threadIdx.x; t < 33 * 33; t += 256) {
    const int lj = t / 33, li = t - lj * 33;
    const int i1 = i10 + li, j1 = j10 + lj;
    const bool in = i1 >= 0 && i1 < W1 && j1 >= 0 && j1 < H1;
    const int fi = 2 * min(max(i1, 0), W1 - 1), fj = 2 * min(max(j1, 0), H1 - 1);
    const size_t fb = (size_t)fj * Wf + fi;
    const bool l_ = fi > 0, r_ = fi < Wf - 1, d_ = fj > 0, u_ = fj < Hf - 1;
    const size_t n0 = (d_ && l_) ? fb - Wf - 1 : fb, n1 = d_ ? fb - Wf : fb, n2 = l_ ? fb - 1 : fb;
    const size_t n4 = r_ ? fb + 1 : fb, n5 = u_ ? fb + Wf : fb, n6 = (u_ && r_) ? fb + Wf + 1 : fb;
    const double w0 = (d_ && l_) ? 0.5 : 0.0, w1 = d_ ? 0.5 : 0.0, w2 = l_ ? 0.5 : 0.0;
    const double w4 = r_ ? 0.5 : 0.0, w5 = u_ ? 0.5 : 0.0, w6 = (u_ && r_) ? 0.5 : 0.0;
    double f[7][NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      f[0][c] = rf[n0 * NV + c]; f[1][c] = rf[n1 * NV + c]; f[2][c] = rf[n2 * NV + c];
      f[3][c] = rf[fb * NV + c];
      f[4][c] = rf[n4 * NV + c]; f[5][c] = rf[n5 * NV + c]; f[6][c] = rf[n6 * NV + c];
    }
    const size_t g1 = (size_t)min(max(j1, 0), H1 - 1) * W1 + min(max(i1, 0), W1 - 1);
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      double v = 0.0;
      v += w0 * f[0][c];
      v += w1 * f[1][c];
      v += w2 * f[2][c];
      v += f[3][c];
      v += w4 * f[4][c];
      v += w5 * f[5][c];
      v += w6 * f[6][c];
      if (!in || (mask1 && mask1[g1 * NV + c])) v = 0.0;
      sh[t * NV + c] = v;
      if (in && li >= 1 && lj >= 1) b1[g1 * NV + c] = v;
    }
  }
  __syncthreads();
  const int I2 = 16 * tx + (threadIdx.x & 15), J2 = 16 * ty + (threadIdx.x >> 4);
  if (I2 >= W2 || J2 >= H2) return;
  const int fi = 2 * I2, fj = 2 * J2;                       // position on the middle lattice
  const bool l_ = fi > 0, r_ = fi < W1 - 1, d_ = fj > 0, u_ = fj < H1 - 1;
  const int cb = (2 * (int)(threadIdx.x >> 4) + 1) * 33 + 2 * (int)(threadIdx.x & 15) + 1;
  const double w0 = (d_ && l_) ? 0.5 : 0.0, w1 = d_ ? 0.5 : 0.0, w2 = l_ ? 0.5 : 0.0;
  const double w4 = r_ ? 0.5 : 0.0, w5 = u_ ? 0.5 : 0.0, w6 = (u_ && r_) ? 0.5 : 0.0;
  const size_t g2 = (size_t)J2 * W2 + I2;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    double v = 0.0;
    v += w0 * sh[(cb - 33 - 1) * NV + c];
    v += w1 * sh[(cb - 33) * NV + c];
    v += w2 * sh[(cb - 1) * NV + c];
    v += sh[cb * NV + c];
    v += w4 * sh[(cb + 1) * NV + c];
    v += w5 * sh[(cb + 33) * NV + c];
    v += w6 * sh[(cb + 33 + 1) * NV + c];
    if (mask2 && mask2[g2 * NV + c]) v = 0.0;
    b2[g2 * NV + c] = v;
  }
}
bool launch_restrict_lattice2(hipStream_t s, int nv, int W2, int H2, int W1, int H1, int Wf, int Hf,
                              const double* rf, const uint8_t* mask1, const uint8_t* mask2, double* b1, double* b2) {
  if (nv < 1 || nv > 2 || Wf != 2 * W1 - 1 || Hf != 2 * H1 - 1 || W1 != 2 * W2 - 1 || H1 != 2 * H2 - 1) return false;
  const dim3 grid(((W2 + 15) / 16) * ((H2 + 15) / 16)), block(256);
  if (nv == 1) hipLaunchKernelGGL(k_restrict_lattice2<1>, grid, block, 0, s, W2, H2, W1, H1, Wf, Hf, rf, mask1, mask2, b1, b2);
  else hipLaunchKernelGGL(k_restrict_lattice2<2>, grid, block, 0, s, W2, H2, W1, H1, Wf, Hf, rf, mask1, mask2, b1, b2);
  NSFEM_HIP(hipGetLastError());
  return true;
}

// value table of the lattice kernel: rows of lp = lmax rounded up to a multiple of 4, zero padded
__global__ __launch_bounds__(256) void k_dict_pad(int n_st, int lmax, int lp, const double* __restrict__ vals,
                                                  double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_st * lp) return;
  const int st = t / lp, k = t - st * lp;
  out[t] = k < lmax ? vals[(size_t)st * lmax + k] : 0.0;
}

// 1 / diagonal of every dictionary entry (the smoother kernels' Jacobi scaling)
__global__ __launch_bounds__(256) void k_dict_dinv(int n_st, int lmax, const int32_t* __restrict__ dpos,
                                                   const double* __restrict__ vals, double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_st) return;
  const int dp = dpos[t];
  out[t] = dp >= 0 ? 1.0 / vals[(size_t)t * lmax + dp] : 0.0;
}

__global__ __launch_bounds__(256) void k_dict_fill(int64_t len, int bsz, const int32_t* __restrict__ src,
                                                   const double* __restrict__ csr,
                                                   double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len * bsz;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int sidx = src[i / bsz];
    out[i] = sidx >= 0 ? csr[(size_t)sidx * bsz + i % bsz] : 0.0;
  }
}

bool build_stencil_dict(hipStream_t s, const Pattern& p, const double* dev_a, const double* dev_b,
                        StencilDict& d, int bsz, bool rect, int min_rows_lattice) {
  d.n_stencils = 0;
  d.max_local = 0;
  d.tables_only = false;
  const char* env = std::getenv("NSFEM_DICT");        // (read per context: tests switch it)
  const bool enabled = env ? std::atoi(env) != 0 : true;
  // (>= 1024 rows: lattice levels down to 33 x 33 nodes get a dictionary; NSFEM_DICT_MIN_ROWS overrides)
  const char* env_min = std::getenv("NSFEM_DICT_MIN_ROWS");
  const int min_rows_full = env_min ? std::atoi(env_min) : 1024;
  // (smaller patterns: accepted only as 2D lattices, for the fused multigrid legs -- see the end)
  const bool small = p.n_rows < min_rows_full;
  const int min_rows = (min_rows_lattice > 0 && !rect && bsz == 1) ? std::min(min_rows_lattice, min_rows_full) : min_rows_full;
  if (!enabled || p.n_rows < min_rows || p.h_rowptr.empty() || (!rect && p.n_rows != p.n_cols)) return false;
  const int n = p.n_rows;
  const size_t nval = (size_t)p.nnz * bsz;
  std::vector<double> va(nval), vb;
  NSFEM_HIP(hipMemcpyAsync(va.data(), dev_a, sizeof(double) * nval, hipMemcpyDeviceToHost, s));
  if (dev_b) {
    vb.resize(nval);
    NSFEM_HIP(hipMemcpyAsync(vb.data(), dev_b, sizeof(double) * nval, hipMemcpyDeviceToHost, s));
  }
  NSFEM_HIP(hipStreamSynchronize(s));
  double sa = 0.0, sb = 0.0;
  for (double v : va) sa = std::max(sa, std::fabs(v));
  for (double v : vb) sb = std::max(sb, std::fabs(v));
  if (!(sa > 0.0)) return false;
  const double qa = std::ldexp(1.0, 41) / sa, qb = sb > 0.0 ? std::ldexp(1.0, 41) / sb : 0.0;
  const double ta = std::ldexp(sa, -40), tb = std::ldexp(sb, -40);
  std::unordered_map<uint64_t, std::vector<int>> seen;
  seen.reserve(1 << 14);
  std::vector<int32_t> sid((size_t)n), rep;
  auto mix = [](uint64_t h, uint64_t v) { return (h ^ v) * 0x100000001b3ULL; };
  int lmax = 0;
  int limit = std::max(64, n / 8);                  // more distinct rows than that: not a lattice
  bool exact = true;                                // every row equals its representative bit for bit
  // Pass 0 merges rows that agree to the tolerance.  When that gives a SMALL dictionary that is not bit-exact (rows
  // of different boundary sides whose values differ in the last bit: the gradient block at h = 1 / 512), pass 1
  // repeats the classification with bitwise comparison: a few entries more, and the copy IS the matrix -- every
  // product may use it.  (Meshes without a binary spacing produce thousands of nearly equal rows: pass 1 gives up at
  // 64 entries and the tolerant dictionary stays.)
  // column offsets are taken relative to the row itself (square operators) or to the row's first
  // column (rectangular blocks between the P2 and P1 numberings)
  auto origin = [&](int r) { return rect ? (p.h_rowptr[r + 1] > p.h_rowptr[r] ? p.h_col[p.h_rowptr[r]] : 0) : r; };
  std::vector<int32_t> sid0, rep0;
  int lmax0 = 0;
  for (int strict = 0; strict < 2; ++strict) {
  if (strict) {
    if (exact || (int)rep.size() > 48) break;
    sid0 = sid; rep0 = rep; lmax0 = lmax;
    seen.clear(); rep.clear(); lmax = 0; exact = true; limit = 64;
  }
  bool gave_up = false;
  for (int r = 0; r < n; ++r) {
    const int b = p.h_rowptr[r], e = p.h_rowptr[r + 1], o_r = origin(r);
    uint64_t h = mix(0xcbf29ce484222325ULL, (uint64_t)(e - b));
    for (int k = b; k < e; ++k) {
      h = mix(h, (uint64_t)(uint32_t)(p.h_col[k] - o_r));
      for (int j = 0; j < bsz; ++j) {
        h = mix(h, (uint64_t)std::llround(va[(size_t)k * bsz + j] * qa));
        if (dev_b) h = mix(h, (uint64_t)std::llround(vb[(size_t)k * bsz + j] * qb));
      }
    }
    std::vector<int>& cand = seen[h];
    int found = -1;
    for (int c : cand) {
      const int rr = rep[c], bb = p.h_rowptr[rr], o_rr = origin(rr);
      if (p.h_rowptr[rr + 1] - bb != e - b) continue;
      bool same = true, bitwise = true;
      for (int k = 0; k < e - b && same; ++k) {
        same = p.h_col[bb + k] - o_rr == p.h_col[b + k] - o_r;
        for (int j = 0; j < bsz && same; ++j) {
          const size_t i0 = (size_t)(bb + k) * bsz + j, i1 = (size_t)(b + k) * bsz + j;
          same = std::fabs(va[i0] - va[i1]) <= ta && (!dev_b || std::fabs(vb[i0] - vb[i1]) <= tb);
          bitwise = bitwise && va[i0] == va[i1] && (!dev_b || vb[i0] == vb[i1]);
        }
      }
      if (same && strict && !bitwise) continue;     // (pass 1: only rows equal bit for bit share an entry)
      if (same) {
        found = c;
        if (exact && !bitwise && std::getenv("NSFEM_DEBUG_SETUP")) {
          std::fprintf(stderr, "[stencil dictionary] %d x %d%s: row %d joins the entry of row %d without being equal bit for bit:",
                       n, p.n_cols, rect ? " (rect)" : "", r, rep[c]);
          for (int k = 0; k < e - b; ++k)
            for (int j = 0; j < bsz; ++j) {
              const size_t i0 = (size_t)(p.h_rowptr[rep[c]] + k) * bsz + j, i1 = (size_t)(b + k) * bsz + j;
              if (va[i0] != va[i1]) std::fprintf(stderr, " [%d.%d] %.17g vs %.17g", k, j, va[i0], va[i1]);
            }
          std::fprintf(stderr, "\n");
        }
        exact = exact && bitwise;
        break;
      }
    }
    if (found < 0) {
      found = (int)rep.size();
      if (found >= limit) {
        if (!strict) return false;
        gave_up = true;
        break;
      }
      rep.push_back(r);
      cand.push_back(found);
      lmax = std::max(lmax, e - b);
    }
    sid[r] = found;
  }
  if (strict && gave_up) {                 // (keep the tolerant classification)
    sid = sid0; rep = rep0; lmax = lmax0; exact = false;
  }
  }
  const int ns = (int)rep.size();
  std::vector<int32_t> len((size_t)ns), off((size_t)ns * lmax, 0), src((size_t)ns * lmax, -1);
  for (int c = 0; c < ns; ++c) {
    const int rr = rep[c], bb = p.h_rowptr[rr];
    len[c] = p.h_rowptr[rr + 1] - bb;
    for (int k = 0; k < len[c]; ++k) {
      off[(size_t)c * lmax + k] = p.h_col[bb + k] - origin(rr);
      src[(size_t)c * lmax + k] = bb + k;
    }
  }
  // per workgroup of 256 rows: the dictionary entries it uses and every row's position in that list
  const int n_wg = (n + 255) / 256;
  std::vector<int32_t> wg_ptr((size_t)n_wg + 1, 0), wg_list;
  std::vector<uint8_t> lid((size_t)n);
  for (int w = 0; w < n_wg; ++w) {
    const size_t first = wg_list.size();
    for (int r = w * 256; r < std::min(n, w * 256 + 256); ++r) {
      size_t j = first;
      while (j < wg_list.size() && wg_list[j] != sid[r]) ++j;
      if (j == wg_list.size()) {
        if (j - first >= (size_t)kDictLocal) return false;      // no locality: not a lattice numbering
        wg_list.push_back(sid[r]);
      }
      lid[r] = (uint8_t)(j - first);
    }
    wg_ptr[w + 1] = (int32_t)wg_list.size();
    d.max_local = std::max(d.max_local, (int)(wg_list.size() - first));
  }
  d.bsz = bsz;
  d.rect = rect;
  if (rect) {
    std::vector<int32_t> cb((size_t)n);
    for (int r = 0; r < n; ++r) cb[r] = origin(r);
    d.cbase.upload(cb, s);
  }
  d.n_rows = n;
  d.n_stencils = ns;
  d.lmax = lmax;
  d.exact = exact;
  if (std::getenv("NSFEM_DEBUG_SETUP"))
    std::fprintf(stderr, "[stencil dictionary] %d x %d%s: %d entries, longest %d, %s\n", n, p.n_cols, rect ? " (rect)" : "",
                 ns, lmax, exact ? "bit-exact" : "NOT bit-exact");
  d.lid.upload(lid, s);
  d.wg_ptr.upload(wg_ptr, s);
  d.wg_list.upload(wg_list, s);
  d.sid.upload(sid, s);
  d.len.upload(len, s);
  d.off.upload(off, s);
  d.src.upload(src, s);
  // position of the diagonal entry of every stencil (square operators)
  std::vector<int32_t> dpos((size_t)ns, -1);
  if (!rect)
    for (int c = 0; c < ns; ++c)
      for (int k = 0; k < len[c]; ++k)
        if (off[(size_t)c * lmax + k] == 0) dpos[c] = k;
  d.dpos.upload(dpos, s);
  // ---- 2D lattice structure (see StencilDict): the smallest positive offset beyond the in-line
  // neighbours is (dj, di) = (1, -r), r <= 2, so the line length is one of three candidates; a
  // candidate is accepted when every offset decomposes and every ROW's entries stay inside the box
  d.lat_w = d.lat_h = d.lat_r = 0;
  if (!rect && bsz == 1 && ns <= 64 && lmax <= 32) {
    int omin = 0;
    for (int c = 0; c < ns; ++c)
      for (int k = 0; k < len[c]; ++k) {
        const int o = off[(size_t)c * lmax + k];
        if (o > 2 && (omin == 0 || o < omin)) omin = o;
      }
    for (int Wc = omin; omin > 0 && Wc <= omin + 2 && d.lat_w == 0; ++Wc) {
      if (Wc < 3 || n % Wc != 0 || n / Wc < 3) continue;
      const int Hc = n / Wc;
      std::vector<int32_t> pk((size_t)ns * lmax, 8 * 32 + 8);
      bool ok = true;
      int reach = 0;
      for (int c = 0; c < ns && ok; ++c)
        for (int k = 0; k < len[c] && ok; ++k) {
          const int o = off[(size_t)c * lmax + k];
          const int dj = (int)std::floor((double)o / Wc + 0.5), di = o - dj * Wc;
          ok = std::abs(di) <= 2 && std::abs(dj) <= 2;
          reach = std::max(reach, std::max(std::abs(di), std::abs(dj)));
          pk[(size_t)c * lmax + k] = (dj + 8) * 32 + (di + 8);
        }
      for (int r = 0; r < n && ok; ++r) {
        const int c = sid[r], i = r % Wc, j = r / Wc;
        for (int k = 0; k < len[c] && ok; ++k) {
          const int q = pk[(size_t)c * lmax + k], dj = (q >> 5) - 8, di = (q & 31) - 8;
          ok = i + di >= 0 && i + di < Wc && j + dj >= 0 && j + dj < Hc;
        }
      }
      if (!ok || reach < 1) continue;
      d.lat_w = Wc;
      d.lat_h = Hc;
      d.lat_r = reach;
      d.pack.upload(pk, s);
      d.h_pack = pk;
      d.h_len.assign(len.begin(), len.end());
      std::vector<uint8_t> s8((size_t)n);
      for (int r = 0; r < n; ++r) s8[r] = (uint8_t)sid[r];
      d.sid8.upload(s8, s);
    }
  }
  NSFEM_HIP(hipStreamSynchronize(s));
  if (small) {
    if (d.lat_w == 0) {            // a small pattern that is no lattice: no dictionary (as before)
      d.n_stencils = 0;
      return false;
    }
    d.tables_only = true;
  }
  return true;
}

void BlockMat::sell_update(hipStream_t s) {
  dict_ready = false;
  if (dict && dict->n_stencils > 0 && pat && br * bc == dict->bsz && dict->n_rows == pat->n_rows) {
    const int64_t len = (int64_t)dict->n_stencils * dict->lmax;
    if (dict_vals.n != (size_t)(len * dict->bsz)) dict_vals.alloc((size_t)(len * dict->bsz));
    hipLaunchKernelGGL(k_dict_fill, dim3((int)std::min<int64_t>((len * dict->bsz + 255) / 256, 4096)), dim3(256),
                       0, s, len, dict->bsz, dict->src.p, vals.p, dict_vals.p);
    NSFEM_HIP(hipGetLastError());
    if (!dict->rect && dict->bsz == 1) {
      if (dict_dinv.n != (size_t)dict->n_stencils) dict_dinv.alloc((size_t)dict->n_stencils);
      hipLaunchKernelGGL(k_dict_dinv, dim3((dict->n_stencils + 255) / 256), dim3(256), 0, s, dict->n_stencils,
                         dict->lmax, dict->dpos.p, dict_vals.p, dict_dinv.p);
      NSFEM_HIP(hipGetLastError());
      if (dict->lat_w > 0) {
        const int lp = (dict->lmax + 3) & ~3;
        if (lat_vals.n != (size_t)dict->n_stencils * lp) lat_vals.alloc((size_t)dict->n_stencils * lp);
        hipLaunchKernelGGL(k_dict_pad, dim3((dict->n_stencils * lp + 255) / 256), dim3(256), 0, s, dict->n_stencils,
                           dict->lmax, lp, dict_vals.p, lat_vals.p);
        NSFEM_HIP(hipGetLastError());
      }
    }
    dict_ready = !dict->tables_only;
  }
  sell_ready = false;
  if (!pat || pat->n_slices == 0 || br != 1 || bc != 1) return;
  if (sell_vals.n != (size_t)pat->sell_len) sell_vals.alloc((size_t)pat->sell_len);
  const int grid = (int)std::min<int64_t>((pat->sell_len + 255) / 256, 4096);
  hipLaunchKernelGGL(k_sell_fill, dim3(grid), dim3(256), 0, s, pat->sell_len, pat->sell_src.p, vals.p,
                     sell_vals.p);
  NSFEM_HIP(hipGetLastError());
  sell_ready = true;
}

// greedy row blocks with <= kStreamNnz nonzeros (host, once per pattern); patterns with a
// longer row keep n_rblk = 0 and use the lane-group kernel
void build_rowblocks(Pattern& p, hipStream_t s) {
  // (1020 = 1024 - 4: the kernel reads a 16-byte aligned window of 256 column quads over the chunk)
  constexpr int kLimit = kStreamNnz - 4;
  std::vector<int32_t> blk;
  blk.push_back(0);
  int start = 0;
  p.n_rblk = 0;
  bool ok = true;
  for (int r = 0; r < p.n_rows && ok; ++r) {
    const int len = p.h_rowptr[r + 1] - p.h_rowptr[r];
    if (len > kLimit) ok = false;
    if (p.h_rowptr[r + 1] - p.h_rowptr[start] > kLimit) {
      blk.push_back(r);
      start = r;
    }
  }
  if (ok) {
    blk.push_back(p.n_rows);
    p.n_rblk = (int)blk.size() - 1;
    std::vector<int32_t> info((size_t)p.n_rblk * 4);
    for (int b = 0; b < p.n_rblk; ++b) {
      info[4 * b] = blk[b];
      info[4 * b + 1] = blk[b + 1];
      info[4 * b + 2] = p.h_rowptr[blk[b]];
      info[4 * b + 3] = p.h_rowptr[blk[b + 1]];
    }
    p.rblk.upload(info, s);
    p.rblk1.upload(blk, s);
    p.max_chunk_rows = 0;
    for (int b = 0; b < p.n_rblk; ++b) p.max_chunk_rows = std::max(p.max_chunk_rows, blk[b + 1] - blk[b]);
    p.h_rblk.swap(blk);
  }
  p.int_b0 = p.int_b1 = 0;
  build_sell(p, s);
}

void mark_interior_blocks(Pattern& p, const std::vector<uint8_t>& ghost_cols) {
  p.int_b0 = p.int_b1 = 0;
  p.wg_w0 = p.wg_w1 = 0;
  if ((int)ghost_cols.size() == p.n_cols && !p.h_rowptr.empty()) {
    // groups of 256 consecutive rows (workgroups of the stencil-dictionary kernel)
    const int nwg = (p.n_rows + 255) / 256;
    int b0 = 0, b1 = 0, r0 = 0;
    for (int b = 0; b <= nwg; ++b) {
      bool dirty = b == nwg;
      if (!dirty)
        for (int k = p.h_rowptr[std::min(p.n_rows, b * 256)]; k < p.h_rowptr[std::min(p.n_rows, b * 256 + 256)] && !dirty; ++k)
          dirty = ghost_cols[p.h_col[k]] != 0;
      if (dirty) {
        if (b - r0 > b1 - b0) { b0 = r0; b1 = b; }
        r0 = b + 1;
      }
    }
    p.wg_w0 = b0;
    p.wg_w1 = b1;
  }
  if (p.n_rblk == 0 || (int)ghost_cols.size() != p.n_cols) return;
  int best0 = 0, best1 = 0, run0 = 0;
  for (int b = 0; b <= p.n_rblk; ++b) {
    bool dirty = b == p.n_rblk;
    if (!dirty)
      for (int k = p.h_rowptr[p.h_rblk[b]]; k < p.h_rowptr[p.h_rblk[b + 1]] && !dirty; ++k)
        dirty = ghost_cols[p.h_col[k]] != 0;
    if (dirty) {
      if (b - run0 > best1 - best0) { best0 = run0; best1 = b; }
      run0 = b + 1;
    }
  }
  p.int_b0 = best0;
  p.int_b1 = best1;
  // the same for the SELL workgroups (4 slices = 256 rows each)
  p.sell_w0 = p.sell_w1 = 0;
  if (p.n_slices > 0) {
    const int nwg = (p.n_slices + 3) / 4;
    int b0 = 0, b1 = 0, r0 = 0;
    for (int b = 0; b <= nwg; ++b) {
      bool dirty = b == nwg;
      if (!dirty)
        for (int k = p.h_rowptr[std::min(p.n_rows, b * 256)]; k < p.h_rowptr[std::min(p.n_rows, b * 256 + 256)] && !dirty; ++k)
          dirty = ghost_cols[p.h_col[k]] != 0;
      if (dirty) {
        if (b - r0 > b1 - b0) { b0 = r0; b1 = b; }
        r0 = b + 1;
      }
    }
    p.sell_w0 = b0;
    p.sell_w1 = b1;
  }
}

template <int EPI>
static void spmv_dispatch(hipStream_t s, const BlockMat& A, int nv, const SpmvArgs& a_in) {
  const Pattern& p = *A.pat;
  static const int use_stream = [] {
    const char* e = std::getenv("NSFEM_SPMV_STREAM");
    return e ? std::atoi(e) : -1;          // -1: per-shape default
  }();
  if (A.dict_ready && A.dict->rect && (a_in.dict_ok || A.dict->exact) && a_in.phase == 0 && nv == 1 &&
      (EPI == EPI_STORE || EPI == EPI_ACCUM) && !a_in.y2) {
    const StencilDict& d = *A.dict;
    const int n_wg = (d.n_rows + 255) / 256;
    const int grid = (n_wg + 7) & ~7;
    const size_t lds = (size_t)d.max_local * d.lmax * (8 * d.bsz + 4) + (size_t)d.max_local * 4 + 8;
#define NSFEM_DICTB(BR, BC)                                                                              \
  hipLaunchKernelGGL((k_spmv_dict_blk<BR, BC, EPI>), dim3(grid), dim3(256), lds, s, d.n_rows, n_wg,     \
                     d.lid.p, d.cbase.p, d.wg_ptr.p, d.wg_list.p, d.len.p, d.off.p, A.dict_vals.p,      \
                     d.lmax, d.max_local, a_in)
    bool done = true;
    if (A.br == 3 && A.bc == 1) NSFEM_DICTB(3, 1);
    else if (A.br == 1 && A.bc == 3) NSFEM_DICTB(1, 3);
    else if (A.br == 2 && A.bc == 1) NSFEM_DICTB(2, 1);
    else if (A.br == 1 && A.bc == 2) NSFEM_DICTB(1, 2);
    else done = false;
#undef NSFEM_DICTB
    if (done) {
      NSFEM_HIP(hipGetLastError());
      return;
    }
  }
  if (A.dict_ready && !A.dict->rect && A.br == 1 && A.bc == 1 && (a_in.dict_ok || A.dict->exact) && EPI != EPI_ACCUM && !(EPI == EPI_STORE && a_in.y2) && nv >= 1 && nv <= 3) {
    const StencilDict& d = *A.dict;
    const int n_all = (d.n_rows + 255) / 256;
    int n_wg = n_all;
    SpmvArgs a = a_in;
    a.skip0 = n_all;
    a.skipn = 0;
    if (a.phase == 1) {                    // interior workgroups only: shift the logical index
      n_wg = p.wg_w1 - p.wg_w0;
      a.skip0 = 0;
      a.skipn = p.wg_w0;
    } else if (a.phase == 2) {
      n_wg = n_all - (p.wg_w1 - p.wg_w0);
      a.skip0 = p.wg_w0;
      a.skipn = p.wg_w1 - p.wg_w0;
    }
    if (n_wg <= 0) return;
    const int grid = (n_wg + 7) & ~7;
    const size_t lds = (size_t)d.max_local * d.lmax * 12 + (size_t)d.max_local * 12;
#define NSFEM_DICT_LAUNCH(KERNEL, NV)                                                                  \
  hipLaunchKernelGGL((KERNEL<NV, EPI>), dim3(grid), dim3(256), lds, s, d.n_rows, n_wg, d.lid.p,        \
                     d.wg_ptr.p, d.wg_list.p, d.len.p, d.off.p, A.dict_vals.p, d.dpos.p, d.lmax,       \
                     d.max_local, a)
    if (nv == 1) NSFEM_DICT_LAUNCH(k_spmv_dict_w8, 1);
    else if (nv == 2) NSFEM_DICT_LAUNCH(k_spmv_dict_w8, 2);
    else NSFEM_DICT_LAUNCH(k_spmv_dict, 3);
#undef NSFEM_DICT_LAUNCH
    NSFEM_HIP(hipGetLastError());
    return;
  }
  if (A.sell_ready && p.n_slices > 0 && A.br == 1 && A.bc == 1 && nv >= 1 && nv <= 3) {
    SpmvArgs a = a_in;
    const int nwg_all = (p.n_slices + 3) / 4;
    int nwg = nwg_all;
    a.skip0 = nwg_all;
    a.skipn = 0;
    if (a.phase == 1) {                    // interior workgroups only: shift the logical index
      nwg = p.sell_w1 - p.sell_w0;
      a.skip0 = 0;
      a.skipn = p.sell_w0;
    } else if (a.phase == 2) {
      nwg = nwg_all - (p.sell_w1 - p.sell_w0);
      a.skip0 = p.sell_w0;
      a.skipn = p.sell_w1 - p.sell_w0;
    }
    if (nwg <= 0) return;
    XcdSplit xs;
    for (int x = 0; x < 9; ++x) xs.s[x] = 0;
    int grid = (nwg + 7) & ~7;
    static const int balance = [] {
      const char* e = std::getenv("NSFEM_SELL_BALANCE");
      return e ? std::atoi(e) : 1;
    }();
    if (a.phase == 0 && balance) {          // whole operator: balanced contiguous XCD ranges
      int most = 0;
      for (int x = 0; x < 9; ++x) xs.s[x] = p.sell_xcd[x];
      for (int x = 0; x < 8; ++x) most = std::max(most, xs.s[x + 1] - xs.s[x]);
      grid = 8 * most;
    }
#define NSFEM_SELL_LAUNCH(NV, U, PIPE)                                                               \
  hipLaunchKernelGGL((k_spmv_sell<NV, EPI, U, PIPE>), dim3(grid), dim3(256), 0, s, p.n_rows,         \
                     p.n_slices, nwg, p.sell_ptr.p, p.sell_col.p, A.sell_vals.p, a, xs)
    // tuning switch (smoother epilogue only): NSFEM_SELL_VARIANT = 0 (groups of 4 entries, 8 waves
    // per SIMD: default -- measured 247-253 us on the 3D n = 64 smoothing launch), 1 (groups of 8:
    // 264 us), 2 (8, software pipelined: 255-263 us), 3 (12 / 10 pipelined: 256-267 us)
    static const int variant = [] {
      const char* e = std::getenv("NSFEM_SELL_VARIANT");
      return e ? std::atoi(e) : 0;
    }();
    const int var = EPI == EPI_CHEB ? variant : 0;
#define NSFEM_SELL_NV(NV, UBIG)                        \
  do {                                                 \
    if (var == 0) NSFEM_SELL_LAUNCH(NV, 4, 0);         \
    else if (var == 1) NSFEM_SELL_LAUNCH(NV, 8, 0);    \
    else if (var == 3) NSFEM_SELL_LAUNCH(NV, UBIG, 1); \
    else NSFEM_SELL_LAUNCH(NV, 8, 1);                  \
  } while (0)
    if (nv == 1) NSFEM_SELL_NV(1, 12);
    else if (nv == 2) NSFEM_SELL_NV(2, 12);
    else NSFEM_SELL_NV(3, 10);
#undef NSFEM_SELL_NV
#undef NSFEM_SELL_LAUNCH
    NSFEM_HIP(hipGetLastError());
    return;
  }
  const bool shape22 = A.br == 2 && A.bc == 2;
  // measured (n = 512): the stream kernel wins on every operator except the short-row P2 x P1
  // gradient (4.6 entries per row), which keeps the lane-group kernel
  const bool long_rows = (double)p.nnz >= 6.0 * p.n_rows;
  const bool stream = p.n_rblk > 0 &&
                      (use_stream == 1 || (use_stream == -1 && kStreamDefault(shape22) && long_rows));
  if (stream) {
    // interior / halo-adjacent split of a partitioned product (see product_with_halo)
    SpmvArgs a = a_in;
    int nb = p.n_rblk;
    const int4* rb = reinterpret_cast<const int4*>(p.rblk.p);
    a.skip0 = nb;
    a.skipn = 0;
    if (a.phase == 1) {
      nb = p.int_b1 - p.int_b0;
      rb += p.int_b0;
      a.skip0 = nb;
    } else if (a.phase == 2) {
      nb = p.n_rblk - (p.int_b1 - p.int_b0);
      a.skip0 = p.int_b0;
      a.skipn = p.int_b1 - p.int_b0;
    }
    if (nb <= 0) return;
    const int grid = (nb + 7) & ~7;
    static const int version = [] {
      // measured in situ (round 2): v1 44-46 us / 278 us (2D n = 512 / 3D n = 64 finest-level smoothing
      // launch), v2 45-47 us / 303 us -- v2 only wins cache-cold in 2D (49-52 vs 54-58 us)
      const char* e = std::getenv("NSFEM_STREAM_V");
      return e ? std::atoi(e) : 1;
    }();
    const int32_t* rb1 = p.rblk1.p + (a.phase == 1 ? p.int_b0 : 0);
#define NSFEM_STREAM(BR, BC, NV)                                                              \
  do {                                                                                        \
    if (version == 1)                                                                         \
      hipLaunchKernelGGL((k_spmv_stream_v1<BR, BC, NV, EPI>), dim3(grid), dim3(256), 0, s, nb, \
                         rb1, p.rowptr.p, p.col.p, A.vals.p, a);                              \
    else                                                                                      \
      hipLaunchKernelGGL((k_spmv_stream<BR, BC, NV, EPI>), dim3(grid), dim3(256),              \
                         (size_t)kStreamNnz * (BR * NV) * 8 + (size_t)(p.max_chunk_rows + 2) * 4, s, \
                         nb, rb, p.rowptr.p, p.col.p, A.vals.p, a);                           \
  } while (0)
    if (A.br == 2 && A.bc == 2 && nv == 1) NSFEM_STREAM(2, 2, 1);
    else if (A.br == 1 && A.bc == 1 && nv == 2) NSFEM_STREAM(1, 1, 2);
    else if (A.br == 1 && A.bc == 1 && nv == 1) NSFEM_STREAM(1, 1, 1);
    else if (A.br == 1 && A.bc == 2 && nv == 1) NSFEM_STREAM(1, 2, 1);
    else if (A.br == 2 && A.bc == 1 && nv == 1) NSFEM_STREAM(2, 1, 1);
    else if (A.br == 3 && A.bc == 3 && nv == 1) NSFEM_STREAM(3, 3, 1);
    else if (A.br == 1 && A.bc == 1 && nv == 3) NSFEM_STREAM(1, 1, 3);
    else if (A.br == 1 && A.bc == 3 && nv == 1) NSFEM_STREAM(1, 3, 1);
    else if (A.br == 3 && A.bc == 1 && nv == 1) NSFEM_STREAM(3, 1, 1);
    else throw Error(NSFEM_ERR_ARG, "unsupported block shape in spmv");
#undef NSFEM_STREAM
    NSFEM_HIP(hipGetLastError());
    return;
  }
  // (the lane-group kernel has no row-block split: the interior phase is empty, the halo-adjacent
  // phase computes every row)
  if (a_in.phase == 1) return;
  const SpmvArgs& a = a_in;
  // lanes per block row: 4 for short rows (P1 7-point, P2 x P1), 8 otherwise; the
  // NSFEM_SPMV_G environment variable overrides it (tuning experiments only)
  static const int forced = [] {
    const char* e = std::getenv("NSFEM_SPMV_G");
    return e ? std::atoi(e) : 0;
  }();
  // measured on MI355X (scripts/gpu_spmv_sweep.py, n = 512): rows with <= ~200 B of matrix data
  // (scalar P2 / P1 operators, grad) run 15-30 % faster with 4 lanes per row, the 2x2
  // Jacobian (414 B/row) and div (460 B/row) with 8
  const double row_bytes = (double)p.nnz / (p.n_rows > 0 ? p.n_rows : 1) * (8.0 * A.br * A.bc + 4.0);
  int G = forced ? forced : (row_bytes <= 200.0 ? 4 : (row_bytes <= 1600.0 ? 8 : 16));
  // interpolation-type operators (<= 2 entries per row, e.g. the P2 <- P1 prolongation): two
  // lanes per row are enough and halve the idle lanes (needs G >= outputs per row)
  if (!forced && (double)p.nnz <= 2.2 * p.n_rows && A.br * nv <= 2) G = 2;
  if (G != 2 && G != 4 && G != 8 && G != 16) G = 8;
  if (G < A.br * nv) G = 4;
  const int rpb = 256 / G;
  int grid = (p.n_rows + rpb - 1) / rpb;
  grid = (grid + 7) & ~7;
#define NSFEM_SPMV(BR, BC, NV)                                                                \
  do {                                                                                        \
    if (G == 2 && BR * NV <= 2)                                                               \
      hipLaunchKernelGGL((k_spmv<BR, BC, NV, (BR * NV <= 2 ? 2 : 4), EPI>), dim3(grid), dim3(256), 0, s, \
                         p.n_rows, p.rowptr.p, p.col.p, A.vals.p, a);                         \
    else if (G == 4)                                                                          \
      hipLaunchKernelGGL((k_spmv<BR, BC, NV, 4, EPI>), dim3(grid), dim3(256), 0, s, p.n_rows,  \
                         p.rowptr.p, p.col.p, A.vals.p, a);                                   \
    else if (G == 8)                                                                          \
      hipLaunchKernelGGL((k_spmv<BR, BC, NV, 8, EPI>), dim3(grid), dim3(256), 0, s, p.n_rows,  \
                         p.rowptr.p, p.col.p, A.vals.p, a);                                   \
    else                                                                                      \
      hipLaunchKernelGGL((k_spmv<BR, BC, NV, 16, EPI>), dim3(grid), dim3(256), 0, s, p.n_rows, \
                         p.rowptr.p, p.col.p, A.vals.p, a);                                   \
  } while (0)
  if (A.br == 2 && A.bc == 2 && nv == 1) NSFEM_SPMV(2, 2, 1);
  else if (A.br == 1 && A.bc == 1 && nv == 2) NSFEM_SPMV(1, 1, 2);
  else if (A.br == 1 && A.bc == 1 && nv == 1) NSFEM_SPMV(1, 1, 1);
  else if (A.br == 1 && A.bc == 2 && nv == 1) NSFEM_SPMV(1, 2, 1);
  else if (A.br == 2 && A.bc == 1 && nv == 1) NSFEM_SPMV(2, 1, 1);
  else if (A.br == 3 && A.bc == 3 && nv == 1) NSFEM_SPMV(3, 3, 1);
  else if (A.br == 1 && A.bc == 1 && nv == 3) NSFEM_SPMV(1, 1, 3);
  else if (A.br == 1 && A.bc == 3 && nv == 1) NSFEM_SPMV(1, 3, 1);
  else if (A.br == 3 && A.bc == 1 && nv == 1) NSFEM_SPMV(3, 1, 1);
  else throw Error(NSFEM_ERR_ARG, "unsupported block shape in spmv");
#undef NSFEM_SPMV
  NSFEM_HIP(hipGetLastError());
}

static SpmvArgs make_args(const double* x, const double* b, double* y, const uint8_t* mask,
                          int maskmode) {
  SpmvArgs a;
  a.x = x; a.b = b; a.y = y; a.mask = mask;
  a.maskmode = mask ? maskmode : MASK_NONE;
  a.dinv = nullptr; a.d = nullptr; a.c1 = 0.0; a.c2 = 1.0;
  static const int nt = [] {
    // measured (n = 512, in-situ smoother launches, 3 runs each): 52.8 us vs 53.6 us without
    const char* e = std::getenv("NSFEM_SPMV_NT");
    return e ? std::atoi(e) : 1;
  }();
  a.nt = nt;
  a.ghost = 0;
  a.skip0 = 0x7fffffff;
  a.skipn = 0;
  a.phase = 0;
  a.ident = 0;
  a.dict_ok = 0;
  a.y2 = nullptr;
#if NSFEM_KNOCKOUTS
  static const int dbg = [] {
    const char* e = std::getenv("NSFEM_SPMV_DEBUG");
    return e ? std::atoi(e) : 0;
  }();
  a.dbg = dbg;
#else
  a.dbg = 0;
#endif
  a.gptr = nullptr;
  a.gbuf = nullptr;
  return a;
}

// y = A x + (node-sorted element contributions gbuf, run gptr[r] .. gptr[r + 1] of row r) with identity rows on
// the rows flagged in rowmask -- ONE launch of the stencil-dictionary kernel.  false: the matrix has no usable
// dictionary copy (the caller multiplies and gathers separately).
bool launch_spmv_with_gather(hipStream_t s, const BlockMat& A, int nv, const double* x, double* y,
                             const uint8_t* rowmask, int maskmode, const int32_t* gptr, const double* gbuf) {
  static const bool on = [] { const char* e = std::getenv("NSFEM_FUSED_GATHER"); return e ? std::atoi(e) != 0 : true; }();
  if (!on || !A.dict_ready || A.dict->rect || A.br != 1 || A.bc != 1 || nv < 1 || nv > 3) return false;
  SpmvArgs a = make_args(x, nullptr, y, rowmask, maskmode);
  a.dict_ok = 1;
  a.gptr = gptr;
  a.gbuf = gbuf;
  spmv_dispatch<EPI_STORE>(s, A, nv, a);
  return true;
}

void launch_spmv(hipStream_t s, const BlockMat& A, int nv, const double* x, double* y,
                 const uint8_t* rowmask, int maskmode, int ghost, int phase, int dict_ok) {
  SpmvArgs a = make_args(x, nullptr, y, rowmask, maskmode);
  a.ghost = ghost;
  a.phase = phase;
  a.dict_ok = dict_ok;
  spmv_dispatch<EPI_STORE>(s, A, nv, a);
}
// y = A x (rows flagged in rowmask -> 0) and, fused, the first Chebyshev-Jacobi step from a zero
// start on y:  d = x1 = c2 dinv y
void launch_spmv_cheb_first(hipStream_t s, const BlockMat& A, int nv, const double* x, double* y,
                            const uint8_t* rowmask, const double* dinv, double c2, double* d,
                            double* x1) {
  SpmvArgs a = make_args(x, nullptr, y, rowmask, MASK_ZERO);
  a.dinv = dinv; a.d = d; a.c1 = c2; a.y2 = x1;
  spmv_dispatch<EPI_STORE>(s, A, nv, a);
}
void launch_residual(hipStream_t s, const BlockMat& A, int nv, const double* x, const double* b,
                     double* y, const uint8_t* rowmask, int maskmode, int phase) {
  SpmvArgs a = make_args(x, b, y, rowmask, maskmode);
  a.phase = phase;
  spmv_dispatch<EPI_RESID>(s, A, nv, a);
}
void launch_spmv_accumulate(hipStream_t s, const BlockMat& A, int nv, const double* x, double* y,
                            const uint8_t* rowmask, int ghost) {
  SpmvArgs a = make_args(x, nullptr, y, rowmask, MASK_ZERO);
  a.ghost = ghost;
  spmv_dispatch<EPI_ACCUM>(s, A, nv, a);
}
void launch_spmv_scaled(hipStream_t s, const BlockMat& A, int nv, double scale, const double* x,
                        double* y, int dict_ok) {
  SpmvArgs a = make_args(x, nullptr, y, nullptr, MASK_NONE);
  a.c2 = scale;
  a.dict_ok = dict_ok;
  spmv_dispatch<EPI_STORE>(s, A, nv, a);
}
void launch_spmv_axpy(hipStream_t s, const BlockMat& A, int nv, double scale, const double* x,
                      double* y, const uint8_t* skipmask, int dict_ok) {
  SpmvArgs a = make_args(x, nullptr, y, skipmask, MASK_IDENTITY);
  a.c2 = scale;
  a.dict_ok = dict_ok;
  spmv_dispatch<EPI_ACCUM>(s, A, nv, a);
}
void launch_cheb_step(hipStream_t s, const BlockMat& A, int nv, const double* x, const double* b,
                      const double* dinv, double* d, double c1, double c2, double* xout,
                      const uint8_t* rowmask, int ghost, int phase, int ident) {
  SpmvArgs a = make_args(x, b, xout, rowmask, MASK_ZERO);
  a.dinv = dinv; a.d = d; a.c1 = c1; a.c2 = c2;
  a.ghost = ghost;
  a.phase = phase;
  a.ident = ident;
  a.dict_ok = 1;
  spmv_dispatch<EPI_CHEB>(s, A, nv, a);
}

// ---------------------------------------------------------- reductions helpers
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// block-wide sum of per-thread values; result valid in every thread
__device__ __forceinline__ double block_sum(double v, double* sh /* [4] */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// fixed-order re-reduction of the kParts partial sums of one dot product
__device__ __forceinline__ double sum_parts(const double* __restrict__ parts, double* sh) {
  static_assert(kParts == 2 * kBlock, "sum_parts assumes kParts == 2 * blockDim");
  const double v = parts[threadIdx.x] + parts[threadIdx.x + kBlock];
  return block_sum(v, sh);
}

#define GRID_STRIDE(i, n) \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------- generic vectors
__global__ __launch_bounds__(256) void k_axpby(int64_t n, double a, const double* __restrict__ x,
                                               double b, const double* __restrict__ y,
                                               double* __restrict__ z) {
  GRID_STRIDE(i, n) z[i] = a * x[i] + (b != 0.0 ? b * y[i] : 0.0);
}
__global__ __launch_bounds__(256) void k_lincomb3(int64_t n, double a, const double* __restrict__ x,
                                                  double b, const double* __restrict__ y, double c,
                                                  const double* __restrict__ z,
                                                  double* __restrict__ out) {
  GRID_STRIDE(i, n) out[i] = a * x[i] + b * y[i] + c * z[i];
}
__global__ __launch_bounds__(256) void k_dot(int64_t n, const double* __restrict__ x,
                                             const double* __restrict__ y,
                                             double* __restrict__ parts) {
  __shared__ double sh[4];
  double v = 0.0;
  GRID_STRIDE(i, n) v += x[i] * y[i];
  v = block_sum(v, sh);
  if (threadIdx.x == 0) parts[blockIdx.x] = v;
}
// Right-hand side and start residual of the velocity-correction mass solve in one pass (one rank, fused step):
//   rhs = M u* + a t  (rhs holds M u* on entry, t = G (p - p_old), a = -k / alpha0)
//   r0  = a t on free rows, 0 on Dirichlet rows  (= rhs - M u0 for the start vector u0 = u*: the M u* parts cancel)
// and the partial sums of |rhs|^2 and |r0|^2 -- instead of an axpby, a residual product with M and two dot launches
__global__ __launch_bounds__(256) void k_correction_setup(int64_t n, double a, double* __restrict__ rhs,
                                                          const double* __restrict__ t,
                                                          const uint8_t* __restrict__ mask, double* __restrict__ r0,
                                                          double* __restrict__ parts_r, double* __restrict__ parts_b) {
  __shared__ double sh[4];
  double sr = 0.0, sb = 0.0;
  GRID_STRIDE(i, n) {
    const double at = a * t[i];
    const double b = rhs[i] + at;
    const double r = mask[i] ? 0.0 : at;
    rhs[i] = b;
    r0[i] = r;
    sb += b * b;
    sr += r * r;
  }
  sr = block_sum(sr, sh);
  sb = block_sum(sb, sh);
  if (threadIdx.x == 0) {
    parts_r[blockIdx.x] = sr;
    parts_b[blockIdx.x] = sb;
  }
}
// Dirichlet rows of a Newton residual AND its squared norm in one launch (one rank): b[dof_j] = x[dof_j] - g_j on the
// nbc constrained dofs (their rows carry mask != 0 and are skipped by the sum over the free rows, so the two parts
// of the kernel touch disjoint entries), parts = sum over all rows of b^2
__global__ __launch_bounds__(256) void k_bc_residual_norm(int64_t n, double* __restrict__ b, const uint8_t* __restrict__ mask,
                                                          int nbc, const int32_t* __restrict__ dofs,
                                                          const double* __restrict__ g, const double* __restrict__ x,
                                                          double* __restrict__ parts) {
  __shared__ double sh[4];
  double v = 0.0;
  GRID_STRIDE(i, n) {
    const double bi = mask[i] ? 0.0 : b[i];
    v += bi * bi;
  }
  GRID_STRIDE(j, nbc) {
    const int d = dofs[j];
    const double r = x[d] - g[j];
    b[d] = r;
    v += r * r;
  }
  v = block_sum(v, sh);
  if (threadIdx.x == 0) parts[blockIdx.x] = v;
}
__global__ __launch_bounds__(256) void k_set_bc_residual(int nbc, const int32_t* __restrict__ dofs,
                                                         const double* __restrict__ g,
                                                         const double* __restrict__ x,
                                                         double* __restrict__ b) {
  GRID_STRIDE(i, nbc) {
    const int d = dofs[i];
    b[d] = x[d] - g[i];
  }
}
__global__ __launch_bounds__(256) void k_set_values(int nbc, const int32_t* __restrict__ dofs,
                                                    const double* __restrict__ g,
                                                    double* __restrict__ x) {
  GRID_STRIDE(i, nbc) x[dofs[i]] = g[i];
}
__global__ __launch_bounds__(256) void k_fill_mask(int nbc, const int32_t* __restrict__ dofs,
                                                   uint8_t* __restrict__ mask) {
  GRID_STRIDE(i, nbc) mask[dofs[i]] = 1;
}
__global__ __launch_bounds__(256) void k_mask_zero(int64_t n, const uint8_t* __restrict__ mask,
                                                   double* __restrict__ x) {
  GRID_STRIDE(i, n) if (mask[i]) x[i] = 0.0;
}
__global__ __launch_bounds__(256) void k_add_scalar(int64_t n, double a, double* __restrict__ x) {
  GRID_STRIDE(i, n) x[i] += a;
}
// x -= mean(x) with the sum given as partials (sum over all n entries)
__global__ __launch_bounds__(256) void k_sub_mean(int64_t n, int64_t n_global,
                                                  const double* __restrict__ parts,
                                                  double* __restrict__ x) {
  __shared__ double sh[4];
  const double mean = sum_parts(parts, sh) / (double)n_global;
  GRID_STRIDE(i, n) x[i] -= mean;
}
__global__ __launch_bounds__(256) void k_sum(int64_t n, const double* __restrict__ x,
                                             double* __restrict__ parts) {
  __shared__ double sh[4];
  double v = 0.0;
  GRID_STRIDE(i, n) v += x[i];
  v = block_sum(v, sh);
  if (threadIdx.x == 0) parts[blockIdx.x] = v;
}
__global__ __launch_bounds__(256) void k_inv_diag(int n_rows, int br, int bc, int nv,
                                                  const int32_t* __restrict__ diag,
                                                  const double* __restrict__ vals,
                                                  const uint8_t* __restrict__ mask,
                                                  double* __restrict__ dinv) {
  const int no = br * nv;
  GRID_STRIDE(i, (int64_t)n_rows * no) {
    const int row = (int)(i / no), o = (int)(i % no);
    const int r = (nv == 1) ? o : 0;
    double d = vals[(size_t)diag[row] * (br * bc) + r * bc + r];
    if (mask && mask[i]) d = 1.0;
    dinv[i] = 1.0 / d;
  }
}

static inline int vgrid(int64_t n) {
  int64_t g = (n + kBlock - 1) / kBlock;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}
#define LAUNCH(kern, grid, s, ...)                                        \
  do {                                                                    \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, s, __VA_ARGS__); \
    NSFEM_HIP(hipGetLastError());                                         \
  } while (0)

// rotating frame: out = g * (-u_y, u_x) per node (2D cross product e_z x u)
__global__ __launch_bounds__(256) void k_rot90(int64_t n_nodes, double g, const double* __restrict__ u,
                                               double* __restrict__ out) {
  GRID_STRIDE(i, n_nodes) {
    const double2 v = reinterpret_cast<const double2*>(u)[i];
    reinterpret_cast<double2*>(out)[i] = make_double2(-g * v.y, g * v.x);
  }
}
// J[s] += g * M[s] * [[0,-1],[1,0]]  (Coriolis block of the velocity Jacobian)
__global__ __launch_bounds__(256) void k_jac_add_skew(int64_t nnz, double g, const double* __restrict__ M,
                                                      double* __restrict__ J) {
  GRID_STRIDE(s, nnz) {
    const double m = g * M[s];
    J[4 * s + 1] -= m;
    J[4 * s + 2] += m;
  }
}
// 3D: out = g x u per node ; J[s] += M[s] [g]_x  (Coriolis block with the vector g = 2 c_cor Omega)
__global__ __launch_bounds__(256) void k_cross3(int64_t n_nodes, double gx, double gy, double gz,
                                                const double* __restrict__ u, double* __restrict__ out) {
  GRID_STRIDE(i, n_nodes) {
    const double a = u[3 * i], b = u[3 * i + 1], c = u[3 * i + 2];
    out[3 * i] = gy * c - gz * b;
    out[3 * i + 1] = gz * a - gx * c;
    out[3 * i + 2] = gx * b - gy * a;
  }
}
__global__ __launch_bounds__(256) void k_jac_add_skew3(int64_t nnz, double gx, double gy, double gz,
                                                       const double* __restrict__ M, double* __restrict__ J) {
  GRID_STRIDE(s, nnz) {
    const double m = M[s];
    double* b = J + 9 * s;
    b[1] -= m * gz; b[2] += m * gy;
    b[3] += m * gz; b[5] -= m * gx;
    b[6] -= m * gy; b[7] += m * gx;
  }
}
void launch_cross3(hipStream_t s, int64_t n_nodes, const double g[3], const double* u, double* out) {
  LAUNCH(k_cross3, vgrid(n_nodes), s, n_nodes, g[0], g[1], g[2], u, out);
}
void launch_jac_add_skew3(hipStream_t s, int64_t nnz, const double g[3], const double* M, double* J) {
  LAUNCH(k_jac_add_skew3, vgrid(nnz), s, nnz, g[0], g[1], g[2], M, J);
}
void launch_rot90(hipStream_t s, int64_t n_nodes, double g, const double* u, double* out) {
  LAUNCH(k_rot90, vgrid(n_nodes), s, n_nodes, g, u, out);
}
void launch_jac_add_skew(hipStream_t s, int64_t nnz, double g, const double* M, double* J) {
  LAUNCH(k_jac_add_skew, vgrid(nnz), s, nnz, g, M, J);
}

void launch_axpby(hipStream_t s, int64_t n, double a, const double* x, double b, const double* y,
                  double* z) {
  LAUNCH(k_axpby, vgrid(n), s, n, a, x, b, y ? y : x, z);
}
void launch_lincomb3(hipStream_t s, int64_t n, double a, const double* x, double b,
                     const double* y, double c, const double* z, double* out) {
  LAUNCH(k_lincomb3, vgrid(n), s, n, a, x, b, y, c, z, out);
}
void launch_scale_combine(hipStream_t s, int nnz, double a, const double* A, double b,
                          const double* B, double* C) {
  LAUNCH(k_axpby, vgrid(nnz), s, (int64_t)nnz, a, A, b, B, C);
}
void launch_correction_setup(hipStream_t s, int64_t n, double a, double* rhs, const double* t, const uint8_t* mask,
                             double* r0, double* parts_r, double* parts_b) {
  LAUNCH(k_correction_setup, kParts, s, n, a, rhs, t, mask, r0, parts_r, parts_b);
}
// x -= mean(x) (the compatible right-hand side of a singular Neumann problem); `parts`: one partial-sum slot
void launch_sum_sub_mean(hipStream_t s, int64_t n, double* x, double* parts) {
  LAUNCH(k_sum, kParts, s, n, x, parts);
  LAUNCH(k_sub_mean, vgrid(n), s, n, n, parts, x);
}
void launch_sum(hipStream_t s, int64_t n, const double* x, double* parts) { LAUNCH(k_sum, kParts, s, n, x, parts); }
void launch_sub_mean(hipStream_t s, int64_t n, int64_t count, const double* parts, double* x) {
  LAUNCH(k_sub_mean, vgrid(n), s, n, count, parts, x);
}
__global__ __launch_bounds__(256) void k_gather_diag(int n, const int32_t* __restrict__ diag, const double* __restrict__ vals,
                                                     double* __restrict__ out) {
  GRID_STRIDE(i, n) out[i] = diag[i] >= 0 ? vals[diag[i]] : 0.0;
}
void launch_gather_diag(hipStream_t s, int n, const int32_t* diag, const double* vals, double* out) {
  LAUNCH(k_gather_diag, vgrid(n), s, n, diag, vals, out);
}
void launch_dot(hipStream_t s, int64_t n, const double* x, const double* y, double* parts) {
  LAUNCH(k_dot, kParts, s, n, x, y, parts);
}
void launch_bc_residual_norm(hipStream_t s, int64_t n, double* b, const uint8_t* mask, int nbc, const int32_t* dofs,
                             const double* g, const double* x, double* parts) {
  LAUNCH(k_bc_residual_norm, kParts, s, n, b, mask, nbc, dofs, g, x, parts);
}
void launch_set_bc_residual(hipStream_t s, int nbc, const int32_t* dofs, const double* g,
                            const double* x, double* b) {
  if (nbc) LAUNCH(k_set_bc_residual, vgrid(nbc), s, nbc, dofs, g, x, b);
}
void launch_set_values(hipStream_t s, int nbc, const int32_t* dofs, const double* g, double* x) {
  if (nbc) LAUNCH(k_set_values, vgrid(nbc), s, nbc, dofs, g, x);
}
void launch_fill_mask(hipStream_t s, int nbc, const int32_t* dofs, uint8_t* mask) {
  if (nbc) LAUNCH(k_fill_mask, vgrid(nbc), s, nbc, dofs, mask);
}
__global__ __launch_bounds__(256) void k_copy_at(int n, const int32_t* __restrict__ dofs,
                                                 const double* __restrict__ r,
                                                 double* __restrict__ z) {
  GRID_STRIDE(i, n) {
    const int d = dofs[i];
    z[d] = r[d];
  }
}
void launch_copy_at(hipStream_t s, int n, const int32_t* dofs, const double* r, double* z) {
  if (n) LAUNCH(k_copy_at, vgrid(n), s, n, dofs, r, z);
}
__global__ __launch_bounds__(256) void k_zero_ghost(int64_t n, const uint8_t* __restrict__ mask,
                                                    double* __restrict__ x) {
  GRID_STRIDE(i, n) if (mask[i] == 2) x[i] = 0.0;
}
__global__ __launch_bounds__(256) void k_overlay_ghost(int64_t n, const uint8_t* __restrict__ ghost,
                                                       uint8_t* __restrict__ mask) {
  GRID_STRIDE(i, n) if (ghost[i]) mask[i] = 2;
}
void launch_overlay_ghost(hipStream_t s, int64_t n, const uint8_t* ghost, uint8_t* mask) {
  if (ghost) LAUNCH(k_overlay_ghost, vgrid(n), s, n, ghost, mask);
}
void launch_zero_ghost(hipStream_t s, int64_t n, const uint8_t* mask, double* x) {
  if (mask) LAUNCH(k_zero_ghost, vgrid(n), s, n, mask, x);
}
void launch_mask_zero(hipStream_t s, int64_t n, const uint8_t* mask, double* x) {
  LAUNCH(k_mask_zero, vgrid(n), s, n, mask, x);
}
void launch_add_scalar(hipStream_t s, int64_t n, double a, double* x) {
  LAUNCH(k_add_scalar, vgrid(n), s, n, a, x);
}
void launch_inv_diag(hipStream_t s, const BlockMat& A, int nv, const uint8_t* rowmask,
                     double* dinv) {
  const Pattern& p = *A.pat;
  NSFEM_REQUIRE(p.diag.n == (size_t)p.n_rows, "inv_diag needs a square pattern");
  LAUNCH(k_inv_diag, vgrid((int64_t)p.n_rows * A.br * nv), s, p.n_rows, A.br, A.bc, nv, p.diag.p,
         A.vals.p, rowmask, dinv);
}

// ------------------------------------------------------------ Krylov scratch
void KrylovWork::ensure(int64_t n_) {
  if (n_ <= n) return;
  clear_graphs();                // the work vectors move: captured pointers are stale
  n = n_;
  for (DevBuf<double>* b : {&r, &rhat, &p, &v, &s, &t, &phat, &shat, &z, &q}) b->alloc((size_t)n);
  if (!parts.p) {
    parts.alloc((size_t)kPartSlots * kParts);
    scal.alloc(16);
    NSFEM_HIP(hipHostMalloc((void**)&h_parts, sizeof(double) * kPartSlots * kParts));
    NSFEM_HIP(hipHostMalloc((void**)&h_res, sizeof(HostResult), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(h_res, 0, sizeof(HostResult));
  }
}
KrylovWork::~KrylovWork() {
  if (h_parts) (void)hipHostFree(h_parts);
  if (h_res) (void)hipHostFree(h_res);
  clear_graphs();
}

void KrylovWork::clear_graphs() {
  for (auto& g : graphs) (void)hipGraphExecDestroy(g.second);
  graphs.clear();
  seen.clear();
}

bool KrylovWork::graphs_enabled(const LinOp& op) const {
  static const bool on = [] {
    // Measured on the MI355X (scripts/micro/launchgap.hip): a dependent kernel boundary costs 2.7 - 2.8 us between
    // eager launches and 1.75 us inside a replayed graph -- but every Krylov iteration ends in a host check, and a
    // replay starts 10 - 16 us after the host calls it where the first eager launch starts after ~3: n = 512 IPCS
    // steps 2.54 ms with every iteration body replayed vs 2.50 ms eager (round 4, same box; round 1: 17.01 vs 16.97).
    // Replay stays opt-in (NSFEM_GRAPHS=1).
    const char* e = std::getenv("NSFEM_GRAPHS");
    return e ? std::atoi(e) != 0 : false;
  }();
  return on && !graphs_off && !graphs_suspended && op.comm == nullptr;      // (RCCL calls are not captured)
}

void KrylovWork::replay(hipStream_t s, const GraphKey& key, const std::function<void()>& body) {
  if (key.epoch != epoch) {             // operators / smoother data changed: baked arguments stale
    if (!graphs.empty()) {
      short_epochs = replays_in_epoch < 64 ? short_epochs + 1 : 0;
      if (short_epochs >= 3) graphs_off = true;     // captures do not pay: eager launches from now on
    }
    clear_graphs();
    epoch = key.epoch;
    replays_in_epoch = 0;
    if (graphs_off) { body(); return; }
  }
  ++replays_in_epoch;
  hipGraphExec_t exec = nullptr;
  for (auto& g : graphs)
    if (g.first == key) { exec = g.second; break; }
  if (!exec) {
    bool known = false;
    for (const GraphKey& k : seen) known = known || k == key;
    if (!known) {                       // first run of this body: eager (first-use set-up inside it may synchronise)
      seen.push_back(key);
      body();
      return;
    }
    hipGraph_t graph = nullptr;
    NSFEM_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    try {
      body();
    } catch (...) {
      (void)hipStreamEndCapture(s, &graph);
      if (graph) (void)hipGraphDestroy(graph);
      throw;
    }
    NSFEM_HIP(hipStreamEndCapture(s, &graph));
    NSFEM_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    if (graphs.size() >= 16) clear_graphs();
    graphs.emplace_back(key, exec);
  }
  NSFEM_HIP(hipGraphLaunch(exec, s));
}

// sums of up to four partial-sum slots -> coherent host memory, then the sequence number (system-scope release)
__global__ __launch_bounds__(256) void k_publish(const double* __restrict__ parts, int n, int s0, int s1, int s2, int s3,
                                                 KrylovWork::HostResult* __restrict__ out, uint64_t seq) {
  __shared__ double sh[4];
  const int slot[4] = {s0, s1, s2, s3};
  double r[4] = {0.0, 0.0, 0.0, 0.0};
  for (int k = 0; k < n; ++k) r[k] = sum_parts(parts + (size_t)slot[k] * kParts, sh);
  if (threadIdx.x == 0) {
    for (int k = 0; k < n; ++k) out->v[k] = r[k];
    __threadfence_system();
    __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// the host side: launch, spin on the sequence number (the stream is in order: everything queued before has finished
// when it arrives); a stream error ends the wait
static void publish_and_wait(hipStream_t s, KrylovWork& w, int n, const int slot[4], double out[4]) {
  static const bool copy_path = std::getenv("NSFEM_SYNC_COPY") != nullptr;      // (the round-3 path, for A/B timing)
  if (copy_path || !w.h_res) {
    int lo = slot[0], hi = slot[0];
    for (int k = 1; k < n; ++k) { lo = std::min(lo, slot[k]); hi = std::max(hi, slot[k]); }
    NSFEM_HIP(hipMemcpyAsync(w.h_parts + (size_t)lo * kParts, w.parts.p + (size_t)lo * kParts,
                             sizeof(double) * (size_t)(hi - lo + 1) * kParts, hipMemcpyDeviceToHost, s));
    NSFEM_HIP(hipStreamSynchronize(s));
    for (int k = 0; k < n; ++k) {
      double acc = 0.0;
      for (int i = 0; i < kParts; ++i) acc += w.h_parts[(size_t)slot[k] * kParts + i];
      out[k] = acc;
    }
    return;
  }
  const uint64_t seq = ++w.seq_no;
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(kBlock), 0, s, (const double*)w.parts.p, n, slot[0], slot[1], slot[2],
                     slot[3], w.h_res, seq);
  NSFEM_HIP(hipGetLastError());
  volatile uint64_t* flag = &w.h_res->seq;
  for (uint64_t spins = 0;; ++spins) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
    if ((spins & 0xffff) == 0xffff) {
      const hipError_t e = hipStreamQuery(s);
      if (e != hipSuccess && e != hipErrorNotReady)
        throw Error(NSFEM_ERR_HIP, std::string("stream error while waiting for a reduction: ") + hipGetErrorString(e));
      if (e == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
        // (the stream drained but the flag store is not visible yet: one more look after a full synchronisation)
        NSFEM_HIP(hipStreamSynchronize(s));
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq)
          throw Error(NSFEM_ERR_HIP, "reduction result never reached the host");
        break;
      }
    }
  }
  for (int k = 0; k < n; ++k) out[k] = w.h_res->v[k];
}

double host_sum_parts(hipStream_t s, KrylovWork& w, int which) {
  const int slot[4] = {which, which, which, which};
  double out[4];
  publish_and_wait(s, w, 1, slot, out);
  return out[0];
}

// two / three slots with ONE device -> host round trip
void host_sum_parts2(hipStream_t s, KrylovWork& w, int a, int b, double& ra, double& rb) {
  const int slot[4] = {a, b, a, a};
  double out[4];
  publish_and_wait(s, w, 2, slot, out);
  ra = out[0];
  rb = out[1];
}
void host_sum_parts3(hipStream_t s, KrylovWork& w, int a, int b, int c, double& ra, double& rb, double& rc) {
  const int slot[4] = {a, b, c, a};
  double out[4];
  publish_and_wait(s, w, 3, slot, out);
  ra = out[0];
  rb = out[1];
  rc = out[2];
}

// solves with rtol at or below this confirm the true residual on convergence (see bicgstab)
constexpr double kConfirmRtol = 1e-10;

// --------------------------------------------------------------- BiCGStab
// partial-sum slots
// (RHO, RR), (TS, TT) and the CG pairs (RZ, RR') are adjacent: one all-reduce per kernel
enum { P_RHO = 0, P_RR = 1, P_TS = 2, P_TT = 3, P_RTV = 4, P_PQ = 5, P_RZ0 = 6, P_RZ1 = 8, P_R0 = 12, P_B0 = 13 };
// (P_R0, P_B0: |r0|^2 and |b|^2 of the running solve, kept until the first convergence check reads them together
// with the current residual -- a solve without communicator starts without a device -> host round trip)
// device scalars
enum { S_RHO_OLD = 0, S_ALPHA = 1, S_OMEGA = 2, S_RHO = 3, S_RHAT2 = 4, S_RHAT2_NEXT = 5 };

// rhat = r ; parts[RHO] = parts[RR] = r.r
// (rcopy: the residual is `r` itself still in the caller's right-hand side -- zero start vector -- and is stored
// to the work vector here instead of by a copy launch of its own)
__global__ __launch_bounds__(256) void k_bicg_start(int64_t n, const double* __restrict__ r,
                                                    double* __restrict__ rhat,
                                                    double* __restrict__ parts,
                                                    double* __restrict__ scal, double* __restrict__ rcopy) {
  __shared__ double sh[4];
  double v = 0.0;
  GRID_STRIDE(i, n) {
    const double ri = r[i];
    rhat[i] = ri;
    if (rcopy) rcopy[i] = ri;
    v += ri * ri;
  }
  v = block_sum(v, sh);
  if (threadIdx.x == 0) {
    parts[P_RHO * kParts + blockIdx.x] = v;
    parts[P_RR * kParts + blockIdx.x] = v;
    parts[P_R0 * kParts + blockIdx.x] = v;
    if (rcopy) {                                               // zero start: |b|^2 = |r0|^2, no dot launch of its own
      parts[P_TS * kParts + blockIdx.x] = v;
      parts[P_B0 * kParts + blockIdx.x] = v;
    }
    if (blockIdx.x == 0) {
      scal[S_RHO_OLD] = 1.0;
      scal[S_ALPHA] = 1.0;
      scal[S_OMEGA] = 1.0;
    }
  }
}

// p = r + beta (p - omega v) ; phat = dinv * p.  When the shadow residual has become
// (numerically) orthogonal to r -- rho = rhat.r ~ 0, e.g. a start residual supported on the
// Dirichlet rows only, which every later residual vanishes on -- the recurrence breaks down:
// restart it from the current residual (rhat = r, p = r).  Every block takes the same decision
// from the same partial sums.
__global__ __launch_bounds__(256) void k_bicg_p(int64_t n, int first, const double* __restrict__ r,
                                                const double* __restrict__ v,
                                                const double* __restrict__ dinv,
                                                double* __restrict__ p, double* __restrict__ phat,
                                                double* __restrict__ rhat,
                                                const double* __restrict__ parts,
                                                double* __restrict__ scal) {
  __shared__ double sh[4];
  double rho = sum_parts(parts + P_RHO * kParts, sh);
  const double rr = sum_parts(parts + P_RR * kParts, sh);
  const double rho_old = scal[S_RHO_OLD], alpha = scal[S_ALPHA], omega = scal[S_OMEGA];
  double rhat2 = first ? rho : scal[S_RHAT2];
  const bool restart = !first && (rho * rho <= 1e-16 * rr * rhat2 || rho_old == 0.0 || omega == 0.0);
  double beta = 0.0;
  if (restart) {
    rho = rr;
    rhat2 = rr;
  } else if (!first) {
    beta = (rho / rho_old) * (alpha / omega);
  }
  GRID_STRIDE(i, n) {
    const double ri = r[i];
    const double pi = (first || restart) ? ri : ri + beta * (p[i] - omega * v[i]);
    p[i] = pi;
    if (restart) rhat[i] = ri;
    if (dinv) phat[i] = dinv[i] * pi;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    scal[S_RHO] = rho;
    scal[S_RHAT2_NEXT] = rhat2;
  }
}

// alpha = rho / (rhat.v) ; s = r - alpha v ; shat = dinv * s
__global__ __launch_bounds__(256) void k_bicg_s(int64_t n, const double* __restrict__ r,
                                                const double* __restrict__ v,
                                                const double* __restrict__ dinv,
                                                double* __restrict__ sv, double* __restrict__ shat,
                                                const double* __restrict__ parts,
                                                double* __restrict__ scal) {
  __shared__ double sh[4];
  const double rtv = sum_parts(parts + P_RTV * kParts, sh);
  const double rho = scal[S_RHO];
  const double alpha = (rtv != 0.0) ? rho / rtv : 0.0;
  GRID_STRIDE(i, n) {
    const double si = r[i] - alpha * v[i];
    sv[i] = si;
    if (dinv) shat[i] = dinv[i] * si;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    scal[S_ALPHA] = alpha;
    scal[S_RHAT2] = scal[S_RHAT2_NEXT];
  }
}

// the five sums the second half of an iteration needs, in adjacent slots (ONE all-reduce):
//   t.s, t.t (omega), rhat.s, rhat.t, s.s -- with r = s - omega t they give the new
//   rho = rhat.r = rhat.s - omega rhat.t  and  |r|^2 = s.s - 2 omega t.s + omega^2 t.t
// without a second reduction after the update
enum { P_RHS = P_TT + 1, P_RHT = P_TT + 2, P_SS = P_TT + 3 };
__global__ __launch_bounds__(256) void k_dot_ts_tt(int64_t n, const double* __restrict__ t,
                                                   const double* __restrict__ sv,
                                                   const double* __restrict__ rhat,
                                                   double* __restrict__ parts) {
  __shared__ double sh[4];
  double a = 0.0, b = 0.0, c = 0.0, d = 0.0, e = 0.0;
  GRID_STRIDE(i, n) {
    const double ti = t[i], si = sv[i], hi = rhat[i];
    a += ti * si;
    b += ti * ti;
    c += hi * si;
    d += hi * ti;
    e += si * si;
  }
  a = block_sum(a, sh);
  b = block_sum(b, sh);
  c = block_sum(c, sh);
  d = block_sum(d, sh);
  e = block_sum(e, sh);
  if (threadIdx.x == 0) {
    parts[P_TS * kParts + blockIdx.x] = a;
    parts[P_TT * kParts + blockIdx.x] = b;
    parts[P_RHS * kParts + blockIdx.x] = c;
    parts[P_RHT * kParts + blockIdx.x] = d;
    parts[P_SS * kParts + blockIdx.x] = e;
  }
}

// omega = ts/tt ; x += alpha phat + omega shat ; r = s - omega t ; rho and |r|^2 of the new
// residual from the five (already global) sums: block 0 stores them as the only non-zero partial
__global__ __launch_bounds__(256) void k_bicg_xr(int64_t n, const double* __restrict__ phat,
                                                 const double* __restrict__ shat,
                                                 const double* __restrict__ sv,
                                                 const double* __restrict__ t,
                                                 double* __restrict__ x, double* __restrict__ r,
                                                 double* __restrict__ parts,
                                                 double* __restrict__ scal, int x_is_zero) {
  __shared__ double sh[4];
  const double ts = sum_parts(parts + P_TS * kParts, sh);
  const double tt = sum_parts(parts + P_TT * kParts, sh);
  const double rhs = sum_parts(parts + P_RHS * kParts, sh);
  const double rht = sum_parts(parts + P_RHT * kParts, sh);
  const double ss = sum_parts(parts + P_SS * kParts, sh);
  const double omega = (tt > 0.0) ? ts / tt : 0.0;
  const double alpha = scal[S_ALPHA];
  const double rho = scal[S_RHO];
  if (x_is_zero) {                       // (first iteration from a zero start vector: x is not read -- the caller
    GRID_STRIDE(i, n) {                  // did not have to clear it; 0 + v == v, the same bits)
      x[i] = alpha * phat[i] + omega * shat[i];
      r[i] = sv[i] - omega * t[i];
    }
  } else {
    GRID_STRIDE(i, n) {
      x[i] += alpha * phat[i] + omega * shat[i];
      r[i] = sv[i] - omega * t[i];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double rr = fmax(ss - 2.0 * omega * ts + omega * omega * tt, 0.0);
    parts[P_RR * kParts + blockIdx.x] = blockIdx.x == 0 ? rr : 0.0;
    parts[P_RHO * kParts + blockIdx.x] = blockIdx.x == 0 ? rhs - omega * rht : 0.0;
    if (blockIdx.x == 0) {
      scal[S_OMEGA] = omega;
      scal[S_RHO_OLD] = rho;
    }
  }
}

// all-reduce `nslots` adjacent partial-sum slots over the ranks of a partitioned mesh
static inline void reduce_slots(const LinOp& op, hipStream_t s, double* parts, int slot, int nslots) {
  if (op.comm) op.comm->allreduce_sum(s, parts + (size_t)slot * kParts, (int64_t)nslots * kParts);
}

int bicgstab(hipStream_t s, KrylovWork& w, const LinOp& op, const double* b, double* x,
             const nsfem_krylov_opts& o, nsfem_solve_info& info) {
  const int64_t n = op.custom ? op.custom->n
                              : (int64_t)op.A->pat->n_rows * op.A->br * op.nv;
  w.ensure(n);
  ++w.touch;
  double* parts = w.parts.p;
  double* scal = w.scal.p;
  auto apply = [&](const double* in, double* out) {
    if (op.custom) {
      op.custom->apply(s, in, out);          // custom operators exchange their own inputs
    } else {
      product_with_halo(op.comm, op.halo, op.halo_width, s, in, op.A->pat, [&](int phase) {
        launch_spmv(s, *op.A, op.nv, in, out, op.rowmask, op.maskmode, 0, phase);
      });
    }
  };
  if (op.x_zero) {
    // zero start vector (Newton updates): r = b - A 0 = b, no operator application (A 0 = 0 on every row,
    // identity rows included); the start kernel below stores it
  } else if (op.custom) {
    op.custom->apply(s, x, w.r.p);
    launch_axpby(s, n, 1.0, b, -1.0, w.r.p, w.r.p);
  } else {
    launch_residual(s, *op.A, op.nv, x, b, w.r.p, op.rowmask, op.maskmode);
  }
  LAUNCH(k_bicg_start, kParts, s, n, op.x_zero ? b : w.r.p, w.rhat.p, parts, scal, op.x_zero ? w.r.p : (double*)nullptr);
  // |b| for the relative criterion goes into the slot next to (rho, |r0|^2): ONE all-reduce for
  // the three start-up sums and one read-back for the two the host needs
  static_assert(P_RR == P_RHO + 1 && P_TS == P_RHO + 2, "start-up slots must be adjacent");
  if (!op.x_zero) launch_dot(s, n, b, b, parts + (op.comm ? P_TS : P_B0) * kParts);
  reduce_slots(op, s, parts, P_RHO, 3);
  // The host needs |r0| and |b| only to form the target of the convergence checks.  Known to the caller (Newton:
  // |b| = the nonlinear residual norm just evaluated, zero start): no round trip.  Otherwise (one rank) the two
  // sums wait in their own slots and the FIRST check reads them together with the residual of that iteration.
  double rr = 0.0, bb = 0.0, r0 = 0.0, bnorm = 0.0, target = 0.0;
  bool deferred = false;
  if (op.x_zero && op.known_bnorm >= 0.0) {
    r0 = bnorm = op.known_bnorm;
  } else if (!op.comm && o.max_iter > 0) {
    deferred = true;
  } else {
    host_sum_parts2(s, w, P_RR, P_TS, rr, bb);
    r0 = std::sqrt(rr);
    bnorm = std::sqrt(bb);
  }
  auto set_target = [&] { target = std::max(o.atol, o.rtol * (bnorm > 0.0 ? bnorm : 1.0)); w.last_target = target; };
  set_target();
  info.residual0 = r0;
  info.residual = r0;
  info.iterations = 0;
  info.converged = deferred ? false : (r0 <= target);
  const int check = o.check_every > 0 ? o.check_every : 1;
  int it = 0;
  // (op.x_zero: the start vector is zero BY CONTRACT and need not be stored: the first update writes x, and a solve
  // that ends without an iteration clears it)
  bool x_unwritten = op.x_zero;
  auto body = [&](int first) {
    LAUNCH(k_bicg_p, kParts, s, n, first, w.r.p, w.v.p, op.prec ? nullptr : op.dinv,
           w.p.p, w.phat.p, w.rhat.p, parts, scal);
    if (op.prec) op.prec->apply(s, w.p.p, w.phat.p);
    apply(w.phat.p, w.v.p);
    launch_dot(s, n, w.rhat.p, w.v.p, parts + P_RTV * kParts);
    reduce_slots(op, s, parts, P_RTV, 1);
    LAUNCH(k_bicg_s, kParts, s, n, w.r.p, w.v.p, op.prec ? nullptr : op.dinv, w.s.p, w.shat.p,
           parts, scal);
    if (op.prec) op.prec->apply(s, w.s.p, w.shat.p);
    apply(w.shat.p, w.t.p);
    LAUNCH(k_dot_ts_tt, kParts, s, n, w.t.p, w.s.p, w.rhat.p, parts);
    reduce_slots(op, s, parts, P_TS, 5);
    LAUNCH(k_bicg_xr, kParts, s, n, w.phat.p, w.shat.p, w.s.p, w.t.p, x, w.r.p, parts, scal, x_unwritten ? 1 : 0);
  };
  // iterations >= 1 replay one captured HIP graph (same kernels, same arguments): removes the
  // host launch cost of the ~100 small multigrid kernels per iteration
  const GraphKey key{op.custom ? (const void*)op.custom : (const void*)op.A, (const void*)op.prec,
                     (const void*)x, (const void*)op.dinv, n, 0, op.graph_epoch};
  // The criterion above runs on RECURRENCE quantities (|r|^2 = s.s - 2 omega t.s + omega^2 t.t is
  // prone to cancellation).  Solves to direct-solver accuracy (rtol <= 1e-10: the parity settings)
  // confirm the TRUE residual b - A x once on reported convergence; if it misses the target the
  // iteration restarts from it (at most twice -- round-off may put a floor under the true residual).
  int confirmations = o.rtol <= kConfirmRtol ? 2 : 0;
  bool restart = false;
  for (;;) {
    while (!info.converged && it < o.max_iter) {
      const int first = (it == 0 || restart) ? 1 : 0;
      if (!w.graphs_enabled(op)) body(first);
      else {
        GraphKey k2 = key;
        k2.parity = first | (x_unwritten ? 2 : 0);
        w.replay(s, k2, [&] { body(first); });
      }
      x_unwritten = false;
      restart = false;
      ++it;
      if ((it >= o.first_check && it % check == 0) || it == o.max_iter) {
        if (deferred) {
          double rr0;
          host_sum_parts3(s, w, P_RR, P_R0, P_B0, rr, rr0, bb);
          r0 = std::sqrt(rr0);
          bnorm = std::sqrt(bb);
          set_target();
          info.residual0 = r0;
          deferred = false;
        } else {
          rr = host_sum_parts(s, w, P_RR);
        }
        if (!std::isfinite(rr) || rr > 1e20 * std::max(r0 * r0, bnorm * bnorm)) {   // NaN / diverging
          info.iterations = it;
          info.residual = rr;
          return NSFEM_ERR_BREAKDOWN;
        }
        info.residual = std::sqrt(rr);
        info.converged = info.residual <= target;
        static const bool dbg = std::getenv("NSFEM_DEBUG_KRYLOV") != nullptr;
        if (dbg) std::fprintf(stderr, "  bicgstab it %d |r| %.3e (target %.3e)\n", it, info.residual, target);
      }
    }
    if (!info.converged || confirmations == 0 || it == 0) break;
    --confirmations;
    if (op.custom) {
      op.custom->apply(s, x, w.r.p);
      launch_axpby(s, n, 1.0, b, -1.0, w.r.p, w.r.p);
    } else {
      product_with_halo(op.comm, op.halo, op.halo_width, s, x, op.A->pat, [&](int phase) {
        launch_residual(s, *op.A, op.nv, x, b, w.r.p, op.rowmask, op.maskmode, phase);
      });
    }
    LAUNCH(k_bicg_start, kParts, s, n, w.r.p, w.rhat.p, parts, scal, (double*)nullptr);
    reduce_slots(op, s, parts, P_RHO, 2);
    const double true_r = std::sqrt(host_sum_parts(s, w, P_RR));
    if (!std::isfinite(true_r)) return NSFEM_ERR_BREAKDOWN;
    info.residual = true_r;
    if (true_r <= target || confirmations == 0 || it >= o.max_iter) {
      info.converged = true_r <= 10.0 * target;     // (floor of the true residual: accept within 10x)
      break;
    }
    info.converged = false;                          // continue from the true residual
    restart = true;
  }
  if (x_unwritten) NSFEM_HIP(hipMemsetAsync(x, 0, sizeof(double) * (size_t)n, s));     // (no iteration ran)
  info.iterations = it;
  return info.converged ? NSFEM_OK : NSFEM_ERR_NOT_CONVERGED;
}

// ---------------------------------------------------------------------- CG
// z = dinv r ; p = z ; parts[rz] = r.z ; parts[RR] = r.r
__global__ __launch_bounds__(256) void k_cg_start(int64_t n, const double* __restrict__ r,
                                                  const double* __restrict__ dinv,
                                                  double* __restrict__ p,
                                                  double* __restrict__ parts, int rz_slot) {
  __shared__ double sh[4];
  double rz = 0.0, rr = 0.0;
  GRID_STRIDE(i, n) {
    const double ri = r[i];
    const double zi = dinv ? dinv[i] * ri : p[i];      // general preconditioner: p holds z
    p[i] = zi;
    rz += ri * zi;
    rr += ri * ri;
  }
  rz = block_sum(rz, sh);
  rr = block_sum(rr, sh);
  if (threadIdx.x == 0) {
    parts[rz_slot * kParts + blockIdx.x] = rz;
    parts[(rz_slot + 1) * kParts + blockIdx.x] = rr;
  }
}

// alpha = rz / p.q ; x += alpha p ; r -= alpha q ; z = dinv r ; new r.z, r.r
__global__ __launch_bounds__(256) void k_cg_update(int64_t n, const double* __restrict__ p,
                                                   const double* __restrict__ q,
                                                   const double* __restrict__ dinv,
                                                   double* __restrict__ x, double* __restrict__ r,
                                                   double* __restrict__ z,
                                                   double* __restrict__ parts, int rz_cur,
                                                   int rz_next) {
  __shared__ double sh[4];
  const double rzc = sum_parts(parts + rz_cur * kParts, sh);
  const double pq = sum_parts(parts + P_PQ * kParts, sh);
  const double alpha = (pq != 0.0) ? rzc / pq : 0.0;
  double rz = 0.0, rr = 0.0;
  GRID_STRIDE(i, n) {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * q[i];
    r[i] = ri;
    if (dinv) {
      const double zi = dinv[i] * ri;
      z[i] = zi;
      rz += ri * zi;
    }
    rr += ri * ri;
  }
  rz = block_sum(rz, sh);
  rr = block_sum(rr, sh);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (dinv) parts[rz_next * kParts + blockIdx.x] = rz;
    parts[(rz_next + 1) * kParts + blockIdx.x] = rr;
  }
}

// beta = rz_next / rz_cur ; p = z + beta p
__global__ __launch_bounds__(256) void k_cg_p(int64_t n, const double* __restrict__ z,
                                              double* __restrict__ p,
                                              const double* __restrict__ parts, int rz_cur,
                                              int rz_next) {
  __shared__ double sh[4];
  const double rzc = sum_parts(parts + rz_cur * kParts, sh);
  const double rzn = sum_parts(parts + rz_next * kParts, sh);
  const double beta = (rzc != 0.0) ? rzn / rzc : 0.0;
  GRID_STRIDE(i, n) p[i] = z[i] + beta * p[i];
}

// ---- single-reduction (Chronopoulos-Gear) CG for operator-preconditioned solves: per iteration
//   x += alpha p ; r -= alpha s ; u = M^-1 r ; w = A u ; (gamma, delta, |r|^2) = (r.u, w.u, r.r)
//   beta = gamma / gamma_old ; alpha = gamma / (delta - beta gamma / alpha_old)
//   p = u + beta p ; s = w + beta s
// ONE fused triple dot product (one all-reduce on partitioned meshes instead of two) and one fused
// vector update per iteration.  gamma_old / alpha_old live in device scalars, double buffered.
__global__ __launch_bounds__(256) void k_cgcg_dots(int64_t n, const double* __restrict__ r,
                                                   const double* __restrict__ u,
                                                   const double* __restrict__ wv,
                                                   double* __restrict__ parts, int slot, int keep_slot) {
  __shared__ double sh[4];
  double a = 0.0, b = 0.0, c = 0.0;
  GRID_STRIDE(i, n) {
    const double ri = r[i], ui = u[i];
    a += ri * ui;
    b += wv[i] * ui;
    c += ri * ri;
  }
  a = block_sum(a, sh);
  b = block_sum(b, sh);
  c = block_sum(c, sh);
  if (threadIdx.x == 0) {
    parts[(slot + 0) * kParts + blockIdx.x] = a;
    parts[(slot + 1) * kParts + blockIdx.x] = b;
    parts[(slot + 2) * kParts + blockIdx.x] = c;
    if (keep_slot >= 0) parts[keep_slot * kParts + blockIdx.x] = c;      // |r0|^2 of the solve (first check)
  }
}

__global__ __launch_bounds__(256) void k_cgcg_update(int64_t n, int first, int parity,
                                                     const double* __restrict__ u,
                                                     const double* __restrict__ wv,
                                                     double* __restrict__ p, double* __restrict__ sv,
                                                     double* __restrict__ x, double* __restrict__ r,
                                                     const double* __restrict__ parts, int slot,
                                                     double* __restrict__ scal) {
  __shared__ double sh[4];
  const double gamma = sum_parts(parts + (slot + 0) * kParts, sh);
  const double delta = sum_parts(parts + (slot + 1) * kParts, sh);
  const double gamma_old = scal[8 + 2 * parity], alpha_old = scal[9 + 2 * parity];
  const double beta = (first || gamma_old == 0.0) ? 0.0 : gamma / gamma_old;
  double den = delta;
  if (!first && alpha_old != 0.0) den -= beta * gamma / alpha_old;
  const double alpha = (den != 0.0) ? gamma / den : 0.0;
  GRID_STRIDE(i, n) {
    const double pi = first ? u[i] : u[i] + beta * p[i];        // (first: p, s hold stale data)
    const double si = first ? wv[i] : wv[i] + beta * sv[i];
    p[i] = pi;
    sv[i] = si;
    x[i] += alpha * pi;
    r[i] -= alpha * si;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    scal[8 + 2 * (parity ^ 1)] = gamma;
    scal[9 + 2 * (parity ^ 1)] = alpha;
  }
}

static int pcg_single_reduction(hipStream_t s, KrylovWork& w, const LinOp& op, const double* rhs,
                                double* x, const nsfem_krylov_opts& o, nsfem_solve_info& info) {
  const Pattern& pat = *op.A->pat;
  const int64_t n = (int64_t)pat.n_rows * op.A->br * op.nv;
  double* parts = w.parts.p;
  double* scal = w.scal.p;
  const int slot = P_RZ0;                       // gamma, delta, |r|^2 in slots 6, 7, 8 ; |b|^2 in 9
  auto precond_and_dots = [&](int keep = -1) {
    op.prec->apply(s, w.r.p, w.z.p);                                    // u
    product_with_halo(op.comm, op.halo, op.halo_width, s, w.z.p, op.A->pat, [&](int phase) {
      launch_spmv(s, *op.A, op.nv, w.z.p, w.q.p, op.rowmask, op.maskmode, 0, phase);   // w = A u
    });
    LAUNCH(k_cgcg_dots, kParts, s, n, w.r.p, w.z.p, w.q.p, parts, slot, keep);
  };
  launch_residual(s, *op.A, op.nv, x, rhs, w.r.p, op.rowmask, op.maskmode);
  // one rank: |r0|^2 and |b|^2 wait in their own slots for the first convergence check (see bicgstab)
  bool deferred = !op.comm && o.max_iter > 0;
  precond_and_dots(deferred ? P_R0 : -1);
  launch_dot(s, n, rhs, rhs, parts + (deferred ? P_B0 : slot + 3) * kParts);
  reduce_slots(op, s, parts, slot, 4);
  double rr = 0.0, bb = 0.0, r0 = 0.0, bnorm = 0.0, target = 0.0;
  if (!deferred) {
    host_sum_parts2(s, w, slot + 2, slot + 3, rr, bb);
    r0 = std::sqrt(rr);
    bnorm = std::sqrt(bb);
  }
  auto set_target = [&] { target = std::max(o.atol, o.rtol * (bnorm > 0.0 ? bnorm : 1.0)); w.last_target = target; };
  set_target();
  info.residual0 = info.residual = r0;
  info.iterations = 0;
  info.converged = deferred ? false : (r0 <= target);
  const int check = o.check_every > 0 ? o.check_every : 1;
  int it = 0;
  // (the recurrence residual r -= alpha s drifts from b - A x; tight solves confirm the true
  // residual on reported convergence and restart from it when it misses the target, see bicgstab)
  int confirmations = o.rtol <= kConfirmRtol ? 2 : 0;
  bool restart = false;
  for (;;) {
    while (!info.converged && it < o.max_iter) {
      const int first = (it == 0 || restart) ? 1 : 0, par = it & 1;
      auto body = [&] {
        LAUNCH(k_cgcg_update, kParts, s, n, first, par, w.z.p, w.q.p, w.p.p, w.s.p, x, w.r.p, parts, slot, scal);
        precond_and_dots();
      };
      if (!w.graphs_enabled(op)) body();
      else {
        const GraphKey key{(const void*)op.A, (const void*)op.prec, (const void*)x, (const void*)rhs, n,
                           16 + 2 * first + par, op.graph_epoch};
        w.replay(s, key, body);
      }
      restart = false;
      reduce_slots(op, s, parts, slot, 3);
      ++it;
      if ((it >= o.first_check && it % check == 0) || it == o.max_iter) {
        if (deferred) {
          double rr0;
          host_sum_parts3(s, w, slot + 2, P_R0, P_B0, rr, rr0, bb);
          r0 = std::sqrt(rr0);
          bnorm = std::sqrt(bb);
          set_target();
          info.residual0 = r0;
          deferred = false;
        } else {
          rr = host_sum_parts(s, w, slot + 2);
        }
        if (!std::isfinite(rr)) {
          info.iterations = it;
          info.residual = rr;
          return NSFEM_ERR_BREAKDOWN;
        }
        info.residual = std::sqrt(rr);
        info.converged = info.residual <= target;
      }
    }
    if (!info.converged || confirmations == 0 || it == 0) break;
    --confirmations;
    product_with_halo(op.comm, op.halo, op.halo_width, s, x, op.A->pat, [&](int phase) {
      launch_residual(s, *op.A, op.nv, x, rhs, w.r.p, op.rowmask, op.maskmode, phase);
    });
    precond_and_dots();                              // u = M^-1 r, w = A u and the three sums of r
    reduce_slots(op, s, parts, slot, 3);
    const double true_r = std::sqrt(host_sum_parts(s, w, slot + 2));
    if (!std::isfinite(true_r)) return NSFEM_ERR_BREAKDOWN;
    info.residual = true_r;
    if (true_r <= target || confirmations == 0 || it >= o.max_iter) {
      info.converged = true_r <= 10.0 * target;
      break;
    }
    info.converged = false;
    restart = true;                                  // next update starts a fresh recurrence (beta = 0)
  }
  info.iterations = it;
  return info.converged ? NSFEM_OK : NSFEM_ERR_NOT_CONVERGED;
}

int pcg(hipStream_t s, KrylovWork& w, const LinOp& op, const double* b, double* x,
        const nsfem_krylov_opts& o, nsfem_solve_info& info, bool project_mean) {
  const Pattern& pat = *op.A->pat;
  const int64_t n = (int64_t)pat.n_rows * op.A->br * op.nv;
  w.ensure(n);
  ++w.touch;
  double* parts = w.parts.p;
  const double* rhs = b;
  if (project_mean) {
    // make the right-hand side compatible with the constant null space
    NSFEM_HIP(hipMemcpyAsync(w.t.p, b, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
    LAUNCH(k_sum, kParts, s, n, w.t.p, parts + P_TS * kParts);
    reduce_slots(op, s, parts, P_TS, 1);
    LAUNCH(k_sub_mean, vgrid(n), s, n, op.n_global > 0 ? op.n_global : n, parts + P_TS * kParts,
           w.t.p);
    if (op.ghostmask) launch_zero_ghost(s, n, op.ghostmask, w.t.p);
    rhs = w.t.p;
  }
  static const bool single_reduction = std::getenv("NSFEM_CG_TWO_REDUCTIONS") == nullptr;
  if (op.prec && single_reduction) return pcg_single_reduction(s, w, op, rhs, x, o, info);
  launch_residual(s, *op.A, op.nv, x, rhs, w.r.p, op.rowmask, op.maskmode);
  int cur = P_RZ0, nxt = P_RZ1;
  if (op.prec) op.prec->apply(s, w.r.p, w.p.p);
  LAUNCH(k_cg_start, kParts, s, n, w.r.p, op.prec ? nullptr : op.dinv, w.p.p, parts, cur);
  // |b| next to (r.z, |r0|^2) -- the first slot of the other (r.z, |r|^2) pair, rewritten by the
  // first iteration: one all-reduce for the three start-up sums
  launch_dot(s, n, rhs, rhs, parts + (cur + 2) * kParts);
  reduce_slots(op, s, parts, cur, 3);
  double rr, bb;
  host_sum_parts2(s, w, cur + 1, cur + 2, rr, bb);
  const double r0 = std::sqrt(rr);
  const double bnorm = std::sqrt(bb);
  const double target = std::max(o.atol, o.rtol * (bnorm > 0.0 ? bnorm : 1.0));
  w.last_target = target;
  info.residual0 = r0;
  info.residual = r0;
  info.iterations = 0;
  info.converged = (r0 <= target);
  const int check = o.check_every > 0 ? o.check_every : 1;
  int it = 0;
  auto body = [&](int cur_, int nxt_) {
    product_with_halo(op.comm, op.halo, op.halo_width, s, w.p.p, op.A->pat, [&](int phase) {
      launch_spmv(s, *op.A, op.nv, w.p.p, w.q.p, op.rowmask, op.maskmode, 0, phase);
    });
    launch_dot(s, n, w.p.p, w.q.p, parts + P_PQ * kParts);
    reduce_slots(op, s, parts, P_PQ, 1);
    LAUNCH(k_cg_update, kParts, s, n, w.p.p, w.q.p, op.prec ? nullptr : op.dinv, x, w.r.p, w.z.p,
           parts, cur_, nxt_);
    if (op.prec) {
      op.prec->apply(s, w.r.p, w.z.p);
      launch_dot(s, n, w.r.p, w.z.p, parts + nxt_ * kParts);
    }
    reduce_slots(op, s, parts, nxt_, 2);
    LAUNCH(k_cg_p, kParts, s, n, w.z.p, w.p.p, parts, cur_, nxt_);
  };
  while (!info.converged && it < o.max_iter) {
    if (!w.graphs_enabled(op)) {
      body(cur, nxt);
    } else {
      const GraphKey key{(const void*)op.A, (const void*)op.prec, (const void*)x,
                         (const void*)op.dinv, n, cur, op.graph_epoch};
      const int c0 = cur, n0 = nxt;
      w.replay(s, key, [&] { body(c0, n0); });
    }
    std::swap(cur, nxt);
    ++it;
    if ((it >= o.first_check && it % check == 0) || it == o.max_iter) {
      rr = host_sum_parts(s, w, cur + 1);
      if (!std::isfinite(rr)) {
        info.iterations = it;
        info.residual = rr;
        return NSFEM_ERR_BREAKDOWN;
      }
      info.residual = std::sqrt(rr);
      info.converged = info.residual <= target;
    }
  }
  info.iterations = it;
  return info.converged ? NSFEM_OK : NSFEM_ERR_NOT_CONVERGED;
}

}  // namespace nsfem
