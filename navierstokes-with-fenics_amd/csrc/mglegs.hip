// Fused multi-level launches of the multigrid cycles on 2D lattice hierarchies ("legs").
//
// The reference solves the pressure Poisson problem and the Newton systems of the diffusion step with one sparse LU
// each (source/ns_ipcs_solver.py:171,205,198-208); here they are Krylov solves preconditioned by geometric multigrid
// (multigrid.hip).  Below the finest level a cycle is a chain of SMALL operations -- on lattices of 513 x 513 nodes
// and less -- that cost 7 - 13 us per launch whatever they compute (round 3: 75 launches = 23 % of a time step).
// This file runs such a chain in ONE launch: k_mg_leg interprets a short program of level operations from LDS.
//
//   * a workgroup owns a tile of T x T nodes of the coarsest level of the leg (the anchor) and the nodes of the finer
//     levels above them (own region of level l: the tile shifted left by the level distance);
//   * every operation computes its output on the own region widened by a halo -- what the later operations of the
//     program read beyond the own region, planned backwards on the host (LegPlan::finalize): a Chebyshev step needs
//     its input one stencil reach further out, a restriction 2 h + 1 fine nodes, a prolongation h / 2 + 1 coarse
//     nodes.  Halo nodes are computed redundantly by the neighbouring workgroups: no communication inside a launch;
//   * the iterate of a level lives in two LDS arrays (x_k, x_(k-1)), the right-hand side in a third; the Chebyshev
//     direction is d_k = x_k - x_(k-1), so the step  d' = c1 d + c2 D^-1 (b - A x),  x' = x + d'  of the one-step
//     kernels needs no array of its own; the new iterate overwrites x_(k-1) in place;
//   * operators are the levels' stencil dictionaries (StencilDict): one byte per node (entry | mask bits) and a dense
//     (2 R + 1)^2 value table per entry in LDS (absent neighbours: value 0, the arrays carry a zeroed guard ring), the
//     products run in the dictionary's (ascending column) order -- the same sums as k_cheb_lattice / k_spmv_dict;
//   * transfers are the lattice interpolation (even-even sublattice; Transfer::is_lattice) and its transpose with the
//     summation order of k_restrict_lattice.
// A single-workgroup launch (tile = the whole anchor lattice) runs the complete bottom of a V-cycle including the
// dense coarsest solve: the "tail".
#include "nsfem_internal.hpp"
#include <algorithm>

namespace nsfem {

// ------------------------------------------------------------------------------------------------ kernel
template <int NV, int R>
__device__ __forceinline__ void leg_product(const double* __restrict__ X, int base, int pitch,
                                            const double* __restrict__ tv, double (&acc)[NV]) {
  constexpr int D = 2 * R + 1;
#pragma unroll
  for (int c = 0; c < NV; ++c) acc[c] = 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) {
      const double v = tv[a * D + b];
      const int n = base + ((a - R) * pitch + (b - R)) * NV;
#pragma unroll
      for (int c = 0; c < NV; ++c) acc[c] += v * X[n + c];
    }
}

// b_c = R r at the coarse node whose fine position is `fb` (index of the fine node, `fp` nodes per fine line): the
// node's own fine value + half of its six fine neighbours in ascending fine index (k_restrict_lattice)
template <int NV>
__device__ __forceinline__ void leg_restrict7(const double* __restrict__ F, size_t fb, size_t fp, bool l_, bool r_,
                                              bool d_, bool u_, double (&out)[NV]) {
  const size_t n0 = (d_ && l_) ? fb - fp - 1 : fb, n1 = d_ ? fb - fp : fb, n2 = l_ ? fb - 1 : fb;
  const size_t n4 = r_ ? fb + 1 : fb, n5 = u_ ? fb + fp : fb, n6 = (u_ && r_) ? fb + fp + 1 : fb;
  const double w0 = (d_ && l_) ? 0.5 : 0.0, w1 = d_ ? 0.5 : 0.0, w2 = l_ ? 0.5 : 0.0;
  const double w4 = r_ ? 0.5 : 0.0, w5 = u_ ? 0.5 : 0.0, w6 = (u_ && r_) ? 0.5 : 0.0;
  double f[7][NV];
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    f[0][c] = F[n0 * NV + c]; f[1][c] = F[n1 * NV + c]; f[2][c] = F[n2 * NV + c];
    f[3][c] = F[fb * NV + c];
    f[4][c] = F[n4 * NV + c]; f[5][c] = F[n5 * NV + c]; f[6][c] = F[n6 * NV + c];
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
    out[c] = v;
  }
}

// nodes of the region [x0, x1) x [y0, y1): thread t takes the nodes t, t + nthr, ... of the flattened region
// (j = t / w through a float reciprocal: exact for the region sizes of a tile, (t + 0.5) / w is never within rounding
// of an integer)
#define LEG_NODES(i, j, x0, x1, y0, y1)                                                          \
  for (int w_ = (x1) - (x0), n_ = w_ * ((y1) - (y0)), t_ = tid, j = 0, i = 0;                    \
       t_ < n_ && (j = (int)__fdividef((float)t_ + 0.5f, (float)w_), i = t_ - j * w_ + (x0), j += (y0), true); t_ += nthr)

template <int NV>
__global__ __launch_bounds__(1024) void k_mg_leg(const LegPlanDev* __restrict__ P, const double* __restrict__ ext_in,
                                                 double* __restrict__ ext_out) {
  extern __shared__ double leg_lds[];
  double* __restrict__ lds = leg_lds;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int tile = blockIdx.x;
  const int ntx = P->ntx, T = P->T;
  const int tj = tile / ntx, ti = tile - tj * ntx;
  const int I0 = ti * T, J0 = tj * T;
  const int I1 = min(I0 + T, P->Wa), J1 = min(J0 + T, P->Ha);
  for (int t = tid; t < P->lds_doubles; t += nthr) lds[t] = 0.0;
  __syncthreads();
  // value tables and entry | mask bytes of every level
  const int nl = P->n_levels;
  for (int l = 0; l < nl; ++l) {
    const LegLevelDev L = P->lv[l];
    if (L.t_off >= 0) {
      double* __restrict__ tab = lds + L.t_off;
      const int stride = L.np + 1, D = 2 * L.R + 1;
      for (int t = tid; t < L.n_st * L.lmax; t += nthr) {
        const int e = t / L.lmax, k = t - e * L.lmax;
        if (k < L.len[e]) {
          const int q = L.pack[t], dj = (q >> 5) - 8, di = (q & 31) - 8;
          tab[e * stride + (dj + L.R) * D + di + L.R] = L.tval[e * L.lp + k];
        }
      }
      for (int t = tid; t < L.n_st; t += nthr) tab[t * stride + L.np] = L.dinv[t];
    }
    if (L.e_off >= 0) {
      uint8_t* __restrict__ S = reinterpret_cast<uint8_t*>(lds) + L.e_off;
      const int lx = I0 << L.shift, ly = J0 << L.shift;
      const int hx = min(I1 << L.shift, L.W), hy = min(J1 << L.shift, L.H);
      const int x0 = max(lx - L.e_g, 0), x1 = min(hx + L.e_g, L.W), y0 = max(ly - L.e_g, 0), y1 = min(hy + L.e_g, L.H);
      const int sb = (L.e_g - ly) * L.e_pitch + L.e_g - lx;
      LEG_NODES(i, j, x0, x1, y0, y1) S[sb + j * L.e_pitch + i] = L.sidm[(size_t)j * L.W + i];
    }
  }
  const int n_ops = P->n_ops;
  for (int o = 0; o < n_ops; ++o) {
    const LegOpRec op = P->op[o];
    if (op.flags & LEGF_SYNC) __syncthreads();
    const int lx = I0 << op.shift, ly = J0 << op.shift;
    const int hx = min(I1 << op.shift, op.W), hy = min(J1 << op.shift, op.H);
    const int x0 = max(lx - op.halo, 0), x1 = min(hx + op.halo, op.W);
    const int y0 = max(ly - op.halo, 0), y1 = min(hy + op.halo, op.H);
    // index of lattice node (i, j) in an array: base + (j * pitch + i) * NV
    const int db = op.d_off + ((op.d_g - ly) * op.d_pitch + op.d_g - lx) * NV, dp = op.d_pitch * NV;
    const int sb = op.s_off + ((op.s_g - ly) * op.s_pitch + op.s_g - lx) * NV, sp = op.s_pitch * NV;
    const int bb = op.b_off + ((op.b_g - ly) * op.b_pitch + op.b_g - lx) * NV, bp = op.b_pitch * NV;
    const uint8_t* __restrict__ S = reinterpret_cast<const uint8_t*>(lds) + op.e_off + (op.e_g - ly) * op.e_pitch + op.e_g - lx;
    switch (op.type) {
      case LEG_LOADB:
      case LEG_LOADX: {
        const double* __restrict__ g = (op.flags & LEGF_EXT_IN) ? ext_in : op.gin;
        LEG_NODES(i, j, x0, x1, y0, y1) {
          const int a = db + j * dp + i * NV;
          const size_t gi = ((size_t)j * op.W + i) * NV;
#pragma unroll
          for (int c = 0; c < NV; ++c) lds[a + c] = g[gi + c];
        }
      } break;
      case LEG_STORE: {
        double* __restrict__ g = (op.flags & LEGF_EXT_OUT) ? ext_out : op.gout;
        LEG_NODES(i, j, lx, hx, ly, hy) {
          const int a = sb + j * sp + i * NV;
          const size_t gi = ((size_t)j * op.W + i) * NV;
#pragma unroll
          for (int c = 0; c < NV; ++c) g[gi + c] = lds[a + c];
        }
      } break;
      case LEG_RESTRICT_G:
      case LEG_RESTRICT_L: {
        // right-hand side = R (vector of the next finer lattice): a global vector (2 W - 1 wide) or array `o_` of the
        // next finer level of this launch; masked rows 0; own nodes optionally stored to gout
        const bool from_lds = op.type == LEG_RESTRICT_L;
        const int flx = I0 << op.o_shift, fly = J0 << op.o_shift;
        const double* __restrict__ F = from_lds ? lds + op.o_off : ((op.flags & LEGF_EXT_IN) ? ext_in : op.gin);
        const size_t fp = from_lds ? (size_t)op.o_pitch : (size_t)op.o_W;
        const int fo = from_lds ? (op.o_g - fly) * op.o_pitch + op.o_g - flx : 0;
        const bool to_lds = op.d_off >= 0;
        double* __restrict__ gout = op.gout;
        LEG_NODES(i, j, x0, x1, y0, y1) {
          const int fi = 2 * i, fj = 2 * j;
          double v[NV];
          leg_restrict7<NV>(F, (size_t)(fo + fj * (int)fp + fi), fp, fi > 0, fi < op.o_W - 1, fj > 0, fj < op.o_H - 1, v);
          const int mk = S[j * op.e_pitch + i] >> 6;
          const bool own = i >= lx && i < hx && j >= ly && j < hy;
#pragma unroll
          for (int c = 0; c < NV; ++c) {
            const double r = ((mk >> c) & 1) ? 0.0 : v[c];
            if (to_lds) lds[db + j * dp + i * NV + c] = r;
            if (gout && own) gout[((size_t)j * op.W + i) * NV + c] = r;
          }
        }
      } break;
      case LEG_CHEB0: {
        // first step from a zero start:  x = d = c2 D^-1 b  (masked rows: 0)
        const double* __restrict__ tab = lds + op.t_off;
        LEG_NODES(i, j, x0, x1, y0, y1) {
          const int sbyte = S[j * op.e_pitch + i], e = sbyte & 63, mk = sbyte >> 6;
          const double di = tab[e * (op.np + 1) + op.np];
#pragma unroll
          for (int c = 0; c < NV; ++c) {
            double v = 0.0;
            if (!((mk >> c) & 1)) v = op.c2 * di * lds[bb + j * bp + i * NV + c];
            else if (op.flags & LEGF_IDENT) v = lds[bb + j * bp + i * NV + c];
            lds[db + j * dp + i * NV + c] = v;
          }
        }
      } break;
      case LEG_CHEB:
      case LEG_RESID: {
        const double* __restrict__ tab = lds + op.t_off;
        const double* __restrict__ X = lds;
        const bool cheb = op.type == LEG_CHEB;
        const bool has_old = cheb && op.c1 != 0.0 && !(op.flags & LEGF_OLD_ZERO);
        const bool ident = (op.flags & LEGF_IDENT) != 0;
        const double c1 = op.c1, c2 = op.c2;
        LEG_NODES(i, j, x0, x1, y0, y1) {
          const int sbyte = S[j * op.e_pitch + i], e = sbyte & 63, mk = sbyte >> 6;
          const double* __restrict__ tv = tab + e * (op.np + 1);
          const int as = sb + j * sp + i * NV, ad = db + j * dp + i * NV, ab = bb + j * bp + i * NV;
          double acc[NV];
          if (op.R == 1) leg_product<NV, 1>(X, as, op.s_pitch, tv, acc);
          else leg_product<NV, 2>(X, as, op.s_pitch, tv, acc);
          if (cheb) {
            const double di = tv[op.np];
#pragma unroll
            for (int c = 0; c < NV; ++c) {
              double xn = 0.0;
              if (!((mk >> c) & 1)) {
                const double xc = X[as + c];
                double dn = c2 * di * (lds[ab + c] - acc[c]);
                if (c1 != 0.0) dn += c1 * (has_old ? xc - lds[ad + c] : xc);
                xn = xc + dn;
              } else if (ident) {
                xn = lds[ab + c];
              }
              lds[ad + c] = xn;
            }
          } else {
#pragma unroll
            for (int c = 0; c < NV; ++c) lds[ad + c] = ((mk >> c) & 1) ? 0.0 : lds[ab + c] - acc[c];
          }
        }
      } break;
      case LEG_PROLONG: {
        // x = [x +] P x_c (array `o_` of the next coarser level of this launch); masked rows 0
        const int clx = I0 << op.o_shift, cly = J0 << op.o_shift;
        const int cb = op.o_off + ((op.o_g - cly) * op.o_pitch + op.o_g - clx) * NV, cp = op.o_pitch * NV;
        const bool add = (op.flags & LEGF_ADD) != 0;
        LEG_NODES(i, j, x0, x1, y0, y1) {
          const int pi = i & 1, pj = j & 1;
          const int c0 = cb + (j >> 1) * cp + (i >> 1) * NV;
          const int c1 = c0 + pi * NV + pj * cp;
          const double pw0 = (pi | pj) ? 0.5 : 1.0, pw1 = (pi | pj) ? 0.5 : 0.0;
          const int mk = S[j * op.e_pitch + i] >> 6;
          const int a = db + j * dp + i * NV;
#pragma unroll
          for (int c = 0; c < NV; ++c) {
            double v = pw0 * lds[c0 + c] + pw1 * lds[c1 + c];
            if (add) v = lds[a + c] + v;
            lds[a + c] = ((mk >> c) & 1) ? 0.0 : v;
          }
        }
      } break;
      case LEG_DENSE: {
        // coarsest level (single workgroup): x[(row, v)] = sum_c Ainv[v][row][c] b[(c, v)], one wavefront per output,
        // lanes stride the columns, shuffle reduction (k_dense_apply)
        const int n = op.n_dense, lane = tid & 63, nw = nthr >> 6;
        for (int idx = tid >> 6; idx < n * NV; idx += nw) {
          const int v = idx / n, row = idx - v * n;
          const double* __restrict__ A = op.gin + (size_t)v * n * n + (size_t)row * n;
          double acc = 0.0;
          for (int c = lane; c < n; c += 64) {
            const int cj = c / op.W, ci = c - cj * op.W;
            acc += A[c] * lds[bb + cj * bp + ci * NV + v];
          }
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
          if (lane == 0) {
            const int rj = row / op.W, ri = row - rj * op.W;
            lds[db + rj * dp + ri * NV + v] = acc;
          }
        }
      } break;
      default: break;
    }
  }
}
#undef LEG_NODES

// ------------------------------------------------------------------------------------------------ planner
void LegPlan::add(int type, int lev, int dst, int src, int flags, double c1, double c2, const double* gin,
                  double* gout, int n_dense) {
  LegOpDev op;
  std::memset(&op, 0, sizeof(op));
  op.type = type; op.lev = lev; op.dst = dst; op.src = src; op.flags = flags; op.n_dense = n_dense;
  op.c1 = c1; op.c2 = c2; op.gin = gin; op.gout = gout;
  ops.push_back(op);
}

bool LegPlan::finalize(hipStream_t s, int T, size_t lds_limit) {
  const int nl = (int)lv.size();
  NSFEM_REQUIRE(nl >= 1 && nl <= kLegMaxLevels && !ops.empty() && (int)ops.size() <= kLegMaxOps,
                "multigrid leg: too many levels or operations for one launch");
  int need[kLegMaxLevels][3], ext[kLegMaxLevels][3], smax[kLegMaxLevels], prod[kLegMaxLevels];
  for (int l = 0; l < nl; ++l) {
    for (int a = 0; a < 3; ++a) need[l][a] = ext[l][a] = -1;
    smax[l] = -1;
    prod[l] = 0;
  }
  auto Rof = [&](int l) { return lv[l]->A->dict->lat_r; };
  auto want = [&](int l, int a, int h) {
    need[l][a] = std::max(need[l][a], h);
    ext[l][a] = std::max(ext[l][a], h);
  };
  auto produce = [&](int l, int a) {          // halo the op must deliver in array a; the array is free before the op
    const int h = std::max(need[l][a], 0);
    need[l][a] = -1;
    ext[l][a] = std::max(ext[l][a], h);
    return h;
  };
  for (int o = (int)ops.size() - 1; o >= 0; --o) {
    LegOpDev& op = ops[o];
    const int l = op.lev;
    NSFEM_REQUIRE(l >= 0 && l < nl, "multigrid leg: level out of range");
    switch (op.type) {
      case LEG_STORE:
        op.halo = 0;
        want(l, op.src, 0);
        break;
      case LEG_LOADB:
      case LEG_LOADX:
        op.halo = produce(l, op.dst);
        break;
      case LEG_CHEB0:
        op.halo = produce(l, op.dst);
        want(l, 2, op.halo);
        smax[l] = std::max(smax[l], op.halo);
        prod[l] = 1;
        break;
      case LEG_CHEB: {
        const int h = produce(l, op.dst);
        op.halo = h;
        if (op.c1 != 0.0 && !(op.flags & LEGF_OLD_ZERO)) want(l, op.dst, h);       // x_(k-1) at the node itself
        want(l, op.src, h + Rof(l));
        want(l, 2, h);
        smax[l] = std::max(smax[l], h);
        prod[l] = 1;
      } break;
      case LEG_RESID: {
        const int h = produce(l, op.dst);
        op.halo = h;
        want(l, op.src, h + Rof(l));
        want(l, 2, h);
        smax[l] = std::max(smax[l], h);
        prod[l] = 1;
      } break;
      case LEG_RESTRICT_G:
      case LEG_RESTRICT_L: {
        int h = op.dst >= 0 ? produce(l, op.dst) : 0;
        op.halo = h;
        smax[l] = std::max(smax[l], h);
        if (op.type == LEG_RESTRICT_L) {
          NSFEM_REQUIRE(l >= 1, "multigrid leg: restriction without a finer level");
          want(l - 1, op.src, 2 * h + 1);
        }
      } break;
      case LEG_PROLONG: {
        NSFEM_REQUIRE(l + 1 < nl, "multigrid leg: prolongation without a coarser level");
        const int h = std::max(need[l][op.dst], 0);
        op.halo = h;
        ext[l][op.dst] = std::max(ext[l][op.dst], h);
        if (!(op.flags & LEGF_ADD)) need[l][op.dst] = -1;
        else need[l][op.dst] = h;
        want(l + 1, op.src, h / 2 + 1);
        smax[l] = std::max(smax[l], h);
      } break;
      case LEG_DENSE:
        op.halo = 0;
        (void)produce(l, op.dst);
        want(l, 2, 0);
        break;
      default:
        NSFEM_REQUIRE(false, "multigrid leg: unknown operation");
    }
  }
  for (int l = 0; l < nl; ++l)
    for (int a = 0; a < 3; ++a)
      NSFEM_REQUIRE(need[l][a] < 0, "multigrid leg: an array is read before the program fills it");
  // ---- LDS layout
  std::memset(&h, 0, sizeof(h));
  h.n_levels = nl;
  h.n_ops = (int)ops.size();
  h.T = T;
  const StencilDict& da = *lv[nl - 1]->A->dict;
  h.Wa = da.lat_w;
  h.Ha = da.lat_h;
  h.ntx = (h.Wa + T - 1) / T;
  h.nty = (h.Ha + T - 1) / T;
  struct Arr { int off = -1, pitch = 1, g = 0; };
  Arr arr[kLegMaxLevels][3];
  size_t cur = 0;               // doubles
  int biggest = 0;
  for (int l = 0; l < nl; ++l) {
    const MGLevel& M = *lv[l];
    const StencilDict& d = *M.A->dict;
    LegLevelDev& L = h.lv[l];
    L.W = d.lat_w; L.H = d.lat_h; L.shift = nl - 1 - l; L.R = d.lat_r;
    L.n_st = d.n_stencils; L.lmax = d.lmax; L.lp = (d.lmax + 3) & ~3; L.np = (2 * L.R + 1) * (2 * L.R + 1);
    L.sidm = M.sidm.p; L.tval = M.A->lat_vals.p; L.pack = d.pack.p; L.len = d.len.p; L.dinv = M.A->dict_dinv.p;
    NSFEM_REQUIRE(L.sidm && L.tval && L.pack && L.len && L.dinv, "multigrid leg: level without lattice tables");
    const int ox = std::min(T << L.shift, L.W), oy = std::min(T << L.shift, L.H);
    for (int a = 0; a < 3; ++a) {
      if (ext[l][a] < 0) continue;
      Arr& A = arr[l][a];
      A.g = ext[l][a] + L.R;
      A.pitch = ox + 2 * A.g;
      A.off = (int)cur;
      cur += (size_t)A.pitch * (oy + 2 * A.g) * nv;
      biggest = std::max(biggest, (ox + 2 * ext[l][a]) * (oy + 2 * ext[l][a]));
    }
    L.t_off = -1;
    if (prod[l]) {
      cur = (cur + 1) & ~(size_t)1;          // 16-byte aligned value rows
      L.t_off = (int)cur;
      cur += (size_t)L.n_st * (L.np + 1);
    }
  }
  {
    // work of the launch relative to the same operations without halos (regions of interior tiles)
    double with = 0.0, without = 0.0;
    for (const LegOpDev& op : ops) {
      if (op.type == LEG_DENSE) continue;
      const double o = (double)std::min(T << h.lv[op.lev].shift, h.lv[op.lev].W) *
                       std::min(T << h.lv[op.lev].shift, h.lv[op.lev].H);
      const double r = (double)std::min((T << h.lv[op.lev].shift) + 2 * op.halo, h.lv[op.lev].W) *
                       std::min((T << h.lv[op.lev].shift) + 2 * op.halo, h.lv[op.lev].H);
      const double wgt = (op.type == LEG_CHEB || op.type == LEG_RESID) ? 3.0 : 1.0;
      with += wgt * r;
      without += wgt * o;
    }
    redundancy = without > 0.0 ? with / without : 1.0;
  }
  size_t bytes = cur * 8;
  for (int l = 0; l < nl; ++l) {
    LegLevelDev& L = h.lv[l];
    L.e_off = -1; L.e_g = 0; L.e_pitch = 1;
    if (smax[l] < 0) continue;
    const int ox = std::min(T << L.shift, L.W), oy = std::min(T << L.shift, L.H);
    L.e_g = smax[l];
    L.e_pitch = ox + 2 * L.e_g;
    L.e_off = (int)bytes;
    bytes += (size_t)L.e_pitch * (oy + 2 * L.e_g);
  }
  bytes = (bytes + 15) & ~(size_t)15;
  h.lds_doubles = (int)(bytes / 8);
  lds_bytes = bytes;
  if (bytes > lds_limit) return false;
  // ---- resolved operation records; a barrier in front of every operation that touches what an operation since
  // the last barrier wrote (or writes what one of them reads).  Resources: (level, array); the set-up phase of the
  // kernel writes the tables and entry bytes (pseudo-arrays 3, 4 of every level)
  uint64_t pend_r = 0, pend_w = 0;
  auto bit = [](int l, int a) { return (uint64_t)1 << (l * 5 + a); };
  for (int l = 0; l < nl; ++l) pend_w |= bit(l, 3) | bit(l, 4);
  for (size_t o = 0; o < ops.size(); ++o) {
    const LegOpDev& op = ops[o];
    const int l = op.lev;
    const LegLevelDev& L = h.lv[l];
    LegOpRec& r = h.op[o];
    std::memset(&r, 0, sizeof(r));
    r.type = op.type; r.flags = op.flags & ~LEGF_SYNC; r.halo = op.halo; r.shift = L.shift;
    r.W = L.W; r.H = L.H; r.np = L.np; r.n_dense = op.n_dense; r.R = L.R;
    r.c1 = op.c1; r.c2 = op.c2; r.gin = op.gin; r.gout = op.gout;
    r.d_off = r.s_off = r.b_off = r.o_off = -1;
    r.d_pitch = r.s_pitch = r.b_pitch = r.o_pitch = 1;
    auto put = [&](int lev, int a, int& off, int& pitch, int& g) {
      if (a < 0 || arr[lev][a].off < 0) return;
      off = arr[lev][a].off; pitch = arr[lev][a].pitch; g = arr[lev][a].g;
    };
    uint64_t rd = 0, wr = 0;
    int olev = -1;
    switch (op.type) {
      case LEG_LOADB: case LEG_LOADX:
        put(l, op.dst, r.d_off, r.d_pitch, r.d_g); wr = bit(l, op.dst); break;
      case LEG_STORE:
        put(l, op.src, r.s_off, r.s_pitch, r.s_g); rd = bit(l, op.src); break;
      case LEG_RESTRICT_G:
        put(l, op.dst, r.d_off, r.d_pitch, r.d_g);
        if (op.dst >= 0) wr = bit(l, op.dst);
        rd = bit(l, 4);
        r.o_W = 2 * L.W - 1; r.o_H = 2 * L.H - 1;
        break;
      case LEG_RESTRICT_L:
        put(l, op.dst, r.d_off, r.d_pitch, r.d_g);
        if (op.dst >= 0) wr = bit(l, op.dst);
        olev = l - 1;
        put(olev, op.src, r.o_off, r.o_pitch, r.o_g);
        rd = bit(olev, op.src) | bit(l, 4);
        break;
      case LEG_CHEB0:
        put(l, op.dst, r.d_off, r.d_pitch, r.d_g); put(l, 2, r.b_off, r.b_pitch, r.b_g);
        wr = bit(l, op.dst); rd = bit(l, 2) | bit(l, 3) | bit(l, 4);
        break;
      case LEG_CHEB: case LEG_RESID:
        put(l, op.dst, r.d_off, r.d_pitch, r.d_g); put(l, op.src, r.s_off, r.s_pitch, r.s_g);
        put(l, 2, r.b_off, r.b_pitch, r.b_g);
        wr = bit(l, op.dst); rd = bit(l, op.src) | bit(l, op.dst) | bit(l, 2) | bit(l, 3) | bit(l, 4);
        break;
      case LEG_PROLONG:
        put(l, op.dst, r.d_off, r.d_pitch, r.d_g);
        olev = l + 1;
        put(olev, op.src, r.o_off, r.o_pitch, r.o_g);
        wr = bit(l, op.dst); rd = bit(olev, op.src) | bit(l, op.dst) | bit(l, 4);
        break;
      case LEG_DENSE:
        put(l, op.dst, r.d_off, r.d_pitch, r.d_g); put(l, 2, r.b_off, r.b_pitch, r.b_g);
        wr = bit(l, op.dst); rd = bit(l, 2);
        break;
      default: break;
    }
    if (olev >= 0) { r.o_shift = h.lv[olev].shift; r.o_W = h.lv[olev].W; r.o_H = h.lv[olev].H; }
    r.e_off = L.e_off >= 0 ? L.e_off : 0; r.e_pitch = L.e_pitch; r.e_g = L.e_g;
    r.t_off = L.t_off >= 0 ? L.t_off : 0;
    if ((wr & (pend_r | pend_w)) || (rd & pend_w)) {
      r.flags |= LEGF_SYNC;
      pend_r = pend_w = 0;
    }
    pend_r |= rd;
    pend_w |= wr;
  }
  // workgroup size by the largest region of the launch (every operation is a loop over its region)
  static const int force_thr = [] { const char* e = std::getenv("NSFEM_LEG_THREADS"); return e ? std::atoi(e) : 0; }();
  threads = biggest <= 320 ? 256 : (biggest <= 768 ? 512 : 1024);
  if (force_thr == 256 || force_thr == 512 || force_thr == 1024) threads = force_thr;
  static const bool dbg = std::getenv("NSFEM_LEG_DEBUG") != nullptr;
  if (dbg) {
    int nsync = 0;
    for (int o = 0; o < h.n_ops; ++o) nsync += (h.op[o].flags & LEGF_SYNC) ? 1 : 0;
    std::fprintf(stderr, "[leg] nv %d levels %d (finest %d x %d) ops %d barriers %d T %d tiles %d x %d lds %zu B threads %d biggest %d redundancy %.2f\n",
                 nv, nl, h.lv[0].W, h.lv[0].H, h.n_ops, nsync, T, h.ntx, h.nty, bytes, threads, biggest, redundancy);
    for (int l = 0; l < nl; ++l)
      std::fprintf(stderr, "[leg]   level %d: %d x %d halos x0 %d x1 %d b %d entries %d\n", l, h.lv[l].W, h.lv[l].H,
                   ext[l][0], ext[l][1], ext[l][2], smax[l]);
  }
#if NSFEM_KNOCKOUTS
  // measurement build only (wrong results): run just the first NSFEM_LEG_MAXOPS operations of every leg
  if (const char* e = std::getenv("NSFEM_LEG_MAXOPS")) h.n_ops = std::min(h.n_ops, std::max(0, std::atoi(e)));
#endif
  dev.upload(&h, 1, s);
  NSFEM_HIP(hipStreamSynchronize(s));
  return true;
}

void LegPlan::launch(hipStream_t s, const double* ext_in, double* ext_out) {
  static bool attr_done[64][3];              // per device and instantiation
  NSFEM_REQUIRE(nv == 1 || nv == 2, "multigrid leg: one or two components");
  int device = 0;
  NSFEM_HIP(hipGetDevice(&device));
  bool& attr_set = attr_done[device & 63][nv];
  if (!attr_set) {
    if (nv == 1)
      NSFEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mg_leg<1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    else
      NSFEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mg_leg<2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const dim3 grid(h.ntx * h.nty), block(threads);
  if (nv == 1) hipLaunchKernelGGL(k_mg_leg<1>, grid, block, lds_bytes, s, (const LegPlanDev*)dev.p, ext_in, ext_out);
  else hipLaunchKernelGGL(k_mg_leg<2>, grid, block, lds_bytes, s, (const LegPlanDev*)dev.p, ext_in, ext_out);
  NSFEM_HIP(hipGetLastError());
  ++launches;
}

// ------------------------------------------------------------------------------------------------ cycles
static bool g_legs_on = false;
static double g_leg_redundancy = 2.5;
static bool g_leg_restrict_inside = false;    // kind 1: the restriction chain inside the launch (halo 2 h + 1 per level)
static int g_leg_group = 2, g_leg_t = 0;
void refresh_leg_switches() {
  // off unless asked for: measured on the MI355X (round 4) the fused launches are correct to
  // 1e-16 but no faster than the launches they replace -- a leg pays 4 - 9 us of launch + set-up and 0.5 - 1 us per
  // operation (a barrier and a dependent LDS chain each), the separate k_cheb_lattice launches 6.5 - 9 us for 2 - 3 steps
  const char* e = std::getenv("NSFEM_MG_LEGS");
  g_legs_on = e ? std::atoi(e) != 0 : false;
  e = std::getenv("NSFEM_LEG_GROUP");
  g_leg_group = e ? std::max(1, std::min(3, std::atoi(e))) : 2;
  e = std::getenv("NSFEM_LEG_T");
  g_leg_t = e ? std::atoi(e) : 0;
  e = std::getenv("NSFEM_LEG_REDUNDANCY");
  g_leg_redundancy = e ? std::atof(e) : 2.5;
  e = std::getenv("NSFEM_LEG_RESTRICT");
  g_leg_restrict_inside = e ? std::atoi(e) != 0 : false;
}

void Multigrid::ensure_sidm(hipStream_t s, MGLevel& L) {
  if (L.sidm_for != (const void*)L.A->dict || L.sidm_mask != (const void*)L.mask || L.sidm.n != (size_t)L.n) {
    if (L.sidm.n != (size_t)L.n) L.sidm.alloc((size_t)L.n);
    launch_lattice_sidm(s, *L.A, nv, L.mask, L.sidm.p);
    L.sidm_for = (const void*)L.A->dict;
    L.sidm_mask = (const void*)L.mask;
  }
}

bool Multigrid::leg_level_ok(size_t l) const {
  const MGLevel& L = lv[l];
  return !L.additive && !(comm_active() && L.has_halo) && lattice_tables_available(*L.A, nv) &&
         L.A->dict->lat_r >= 1 && L.A->dict->lat_r <= 2 && (int64_t)L.A->dict->lat_w * L.A->dict->lat_h == L.n;
}

// levels l, l + 1 nest as lattices and the transfer between them is the lattice interpolation
static bool leg_transfer_ok(Multigrid& mg, size_t l) {
  if (!lattice_transfers_enabled() || l + 1 >= mg.lv.size()) return false;
  MGLevel& L = mg.lv[l];
  if (!L.transfer) return false;
  const StencilDict& df = *L.A->dict;
  const StencilDict& dc = *mg.lv[l + 1].A->dict;
  if (dc.lat_w != (df.lat_w + 1) / 2 || dc.lat_h != (df.lat_h + 1) / 2) return false;
  return L.transfer->is_lattice(df.lat_w, df.lat_h);
}

// tile size of a multi-workgroup leg.  A workgroup runs its operations one after the other with a barrier between
// them, so a launch is fast when MANY small workgroups share a CU (their phases overlap) -- as long as the halo does
// not multiply the work: the smallest tile whose widest region stays within `kLegRedundancy` times its own nodes.
static bool finalize_tiled(hipStream_t s, LegPlan& p, size_t lds_limit) {
  static const int cand[] = {2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32};
  if (g_leg_t > 0) return p.finalize(s, g_leg_t, lds_limit);
  int best = 0;
  for (int T : cand) {
    if (!p.finalize(s, T, lds_limit)) break;              // (larger tiles need more LDS still)
    best = T;
    if (p.redundancy <= g_leg_redundancy) return true;
  }
  return best > 0 && p.finalize(s, best, lds_limit);
}

constexpr size_t kLegLds = 160 * 1024 - 64;

void Multigrid::build_legs(hipStream_t s) {
  legs_kind = 0;
  legs_down.clear();
  legs_up.clear();
  leg_tail.reset();
  leg_coarse.reset();
  if (!g_legs_on || comm_active() || lv.size() < 2 || nv > 2) return;
  const int pre = pre_degree >= 0 ? pre_degree : degree;
  auto coeffs = [&](const MGLevel& L, int steps, std::vector<double>& c1, std::vector<double>& c2) {
    c1.resize(steps); c2.resize(steps);
    double rho = 0.0;
    for (int k = 0; k < steps; ++k) {
      double rn;
      cheb_coeffs(L, k, rho, c1[k], c2[k], rn);
      rho = rn;
    }
  };
  std::vector<double> c1, c2;
  if (truncated() && pre == 0) {
    // ---- kind 1: everything below the finest level in one launch.  b_1 = R b_0 ... b_t = R b_(t-1), the truncated
    // level solved by Chebyshev iteration from zero, then prolongation + post-smoothing up to level 1; the finest
    // level's own launch (k_cheb_lattice) takes x_1 as the coarse correction of its fused prolongation
    const size_t t = active - 1;
    if (t < 1 || !lattice_ok(lv[0])) return;
    for (size_t l = 1; l <= t; ++l)
      if (!leg_level_ok(l) || lv[l].A->dict->lat_r != 1) return;
    for (size_t l = 0; l < t; ++l)
      if (!leg_transfer_ok(*this, l)) return;
    if (trunc_steps < 1 || trunc_steps + 2 * (int)t > kLegMaxOps - 8) return;
    std::unique_ptr<LegPlan> p(new LegPlan);
    p->nv = nv;
    for (size_t l = 1; l <= t; ++l) {
      ensure_sidm(s, lv[l]);
      p->lv.push_back(&lv[l]);
    }
    const int nb = (int)t - 1;                       // plan index of the truncated level
    if (g_leg_restrict_inside) {
      p->add(LEG_RESTRICT_G, 0, 2, -1, LEGF_EXT_IN);
      for (int l = 1; l <= nb; ++l) p->add(LEG_RESTRICT_L, l, 2, 2);
    } else {
      // (the restriction chain runs before the launch -- k_restrict_lattice2 / k_restrict_lattice read the fine vector
      // once; inside the launch its halo of 2 h + 1 per level multiplies the tile)
      for (int l = 0; l <= nb; ++l) p->add(LEG_LOADB, l, 2, -1, 0, 0.0, 0.0, lv[(size_t)l + 1].b.p);
    }
    int cur = 0;
    coeffs(lv[t], trunc_steps, c1, c2);
    p->add(LEG_CHEB0, nb, 0, -1, 0, 0.0, c2[0]);
    for (int k = 1; k < trunc_steps; ++k) {
      p->add(LEG_CHEB, nb, 1 - cur, cur, k == 1 ? LEGF_OLD_ZERO : 0, c1[k], c2[k]);
      cur = 1 - cur;
    }
    for (int l = nb - 1; l >= 0; --l) {
      p->add(LEG_PROLONG, l, 0, cur);
      cur = 0;
      coeffs(lv[(size_t)l + 1], degree, c1, c2);
      for (int k = 0; k < degree; ++k) {
        p->add(LEG_CHEB, l, 1 - cur, cur, 0, c1[k], c2[k]);
        cur = 1 - cur;
      }
    }
    p->add(LEG_STORE, 0, -1, cur, 0, 0.0, 0.0, nullptr, lv[1].x.p);
    if (!finalize_tiled(s, *p, kLegLds)) return;
    leg_coarse = std::move(p);
    legs_kind = 1;
    return;
  }
  if (truncated() || pre <= 0 || !dense_coarse || smoother_only) return;
  // ---- kind 2: V(pre, degree) with a dense coarsest level
  const size_t nlv = lv.size();
  if (lv[nlv - 1].n > 144) return;                   // the dense solve of the tail is one workgroup's work
  for (size_t l = 0; l < nlv; ++l)
    if (!leg_level_ok(l) || lv[l].A->dict->lat_r != 1) return;
  for (size_t l = 0; l + 1 < nlv; ++l)
    if (!leg_transfer_ok(*this, l)) return;
  for (size_t l = 0; l < nlv; ++l) ensure_sidm(s, lv[l]);
  // pre-smoothed iterate of a level between the down- and the up-leg: the level's x (level 0: its xc)
  auto xkeep = [&](size_t l) { return l == 0 ? lv[0].xc.p : lv[l].x.p; };
  // down part of level l inside plan p at plan index pl: [load b] pre-smoothing from zero, residual, store x
  auto emit_down = [&](LegPlan& p, int pl, size_t l, bool load, bool store_x, int& cur, int& res) {
    if (load) p.add(LEG_LOADB, pl, 2, -1, l == 0 ? LEGF_EXT_IN : 0, 0.0, 0.0, l == 0 ? nullptr : lv[l].b.p);
    coeffs(lv[l], pre, c1, c2);
    p.add(LEG_CHEB0, pl, 0, -1, 0, 0.0, c2[0]);
    cur = 0;
    for (int k = 1; k < pre; ++k) {
      p.add(LEG_CHEB, pl, 1 - cur, cur, k == 1 ? LEGF_OLD_ZERO : 0, c1[k], c2[k]);
      cur = 1 - cur;
    }
    if (store_x) p.add(LEG_STORE, pl, -1, cur, 0, 0.0, 0.0, nullptr, xkeep(l));
    res = 1 - cur;
    p.add(LEG_RESID, pl, res, cur);
  };
  auto emit_up = [&](LegPlan& p, int pl, size_t l, int& cur, int cur_coarse, bool ident) {
    p.add(LEG_PROLONG, pl, cur, cur_coarse, LEGF_ADD);
    coeffs(lv[l], degree, c1, c2);
    for (int k = 0; k < degree; ++k) {
      p.add(LEG_CHEB, pl, 1 - cur, cur, (ident && k == degree - 1) ? LEGF_IDENT : 0, c1[k], c2[k]);
      cur = 1 - cur;
    }
  };
  // tail: levels t0 .. nlv - 1 in one workgroup; t0 = the first level from which it fits the LDS
  size_t t0 = nlv;
  std::unique_ptr<LegPlan> tail;
  for (size_t cand = 0; cand + 1 < nlv; ++cand) {
    std::unique_ptr<LegPlan> p(new LegPlan);
    p->nv = nv;
    for (size_t l = cand; l < nlv; ++l) p->lv.push_back(&lv[l]);
    const int np = (int)(nlv - cand);
    std::vector<int> curv(np, 0);
    for (int pl = 0; pl + 1 < np; ++pl) {
      int res;
      emit_down(*p, pl, cand + pl, pl == 0, false, curv[pl], res);
      p->add(LEG_RESTRICT_L, pl + 1, 2, res);
    }
    p->add(LEG_DENSE, np - 1, 0, -1, 0, 0.0, 0.0, coarse_inv.p, nullptr, lv[nlv - 1].n);
    curv[np - 1] = 0;
    for (int pl = np - 2; pl >= 0; --pl)
      emit_up(*p, pl, cand + pl, curv[pl], curv[pl + 1], identity_rows && cand + pl == 0);
    p->add(LEG_STORE, 0, -1, curv[0], cand == 0 ? LEGF_EXT_OUT : 0, 0.0, 0.0, nullptr, cand == 0 ? nullptr : lv[cand].x.p);
    const StencilDict& da = *lv[nlv - 1].A->dict;
    if ((int)p->ops.size() <= kLegMaxOps && p->finalize(s, std::max(da.lat_w, da.lat_h), kLegLds)) {
      p->threads = 1024;
      t0 = cand;
      tail = std::move(p);
      break;
    }
  }
  if (!tail) return;
  // legs above the tail: groups of g levels, the finest group first
  std::vector<std::unique_ptr<LegPlan>> down, up;
  for (size_t a = 0; a < t0;) {
    const size_t g = std::min<size_t>((size_t)g_leg_group, t0 - a);
    // down: levels a .. a + g - 1, anchor a + g (its right-hand side is the leg's output)
    {
      std::unique_ptr<LegPlan> p(new LegPlan);
      p->nv = nv;
      for (size_t l = a; l <= a + g; ++l) p->lv.push_back(&lv[l]);
      for (int pl = 0; pl < (int)g; ++pl) {
        int cur, res;
        emit_down(*p, pl, a + pl, pl == 0, true, cur, res);
        const bool last = pl + 1 == (int)g;
        p->add(LEG_RESTRICT_L, pl + 1, last ? -1 : 2, res, 0, 0.0, 0.0, nullptr, lv[a + pl + 1].b.p);
      }
      if (!finalize_tiled(s, *p, kLegLds)) return;
      down.push_back(std::move(p));
    }
    // up: the anchor's x comes from the coarser part; levels a + g - 1 .. a prolongate, smooth, the finest stores
    {
      std::unique_ptr<LegPlan> p(new LegPlan);
      p->nv = nv;
      for (size_t l = a; l <= a + g; ++l) p->lv.push_back(&lv[l]);
      p->add(LEG_LOADX, (int)g, 0, -1, 0, 0.0, 0.0, lv[a + g].x.p);
      int cur_c = 0;
      for (int pl = (int)g - 1; pl >= 0; --pl) {
        const size_t l = a + pl;
        p->add(LEG_LOADX, pl, 0, -1, 0, 0.0, 0.0, xkeep(l));
        p->add(LEG_LOADB, pl, 2, -1, l == 0 ? LEGF_EXT_IN : 0, 0.0, 0.0, l == 0 ? nullptr : lv[l].b.p);
        int cur = 0;
        emit_up(*p, pl, l, cur, cur_c, identity_rows && l == 0);
        cur_c = cur;
      }
      p->add(LEG_STORE, 0, -1, cur_c, a == 0 ? LEGF_EXT_OUT : 0, 0.0, 0.0, nullptr, a == 0 ? nullptr : lv[a].x.p);
      if (!finalize_tiled(s, *p, kLegLds)) return;
      up.push_back(std::move(p));
    }
    a += g;
  }
  legs_down = std::move(down);
  legs_up = std::move(up);
  leg_tail = std::move(tail);
  legs_kind = 2;
}

// the cycle through the fused launches; false: this hierarchy has none (the caller runs the separate launches)
bool Multigrid::vcycle_legs(hipStream_t s, const double* b, double* x) {
  if (legs_kind < 0) build_legs(s);
  if (legs_kind == 1) {
    if (!g_leg_restrict_inside) {
      const size_t t = active - 1;
      const double* fine = b;
      for (size_t l = 0; l < t;) {
        const StencilDict& d0 = *lv[l].A->dict;
        const StencilDict& d1 = *lv[l + 1].A->dict;
        if (l + 2 <= t) {
          const StencilDict& d2 = *lv[l + 2].A->dict;
          NSFEM_REQUIRE(launch_restrict_lattice2(s, nv, d2.lat_w, d2.lat_h, d1.lat_w, d1.lat_h, d0.lat_w, d0.lat_h, fine,
                                                 lv[l + 1].mask, lv[l + 2].mask, lv[l + 1].b.p, lv[l + 2].b.p),
                        "lattice restriction: levels do not nest");
          fine = lv[l + 2].b.p;
          l += 2;
        } else {
          NSFEM_REQUIRE(launch_restrict_lattice(s, nv, d1.lat_w, d1.lat_h, d0.lat_w, d0.lat_h, fine, lv[l + 1].mask,
                                                lv[l + 1].b.p),
                        "lattice restriction: levels do not nest");
          fine = lv[l + 1].b.p;
          l += 1;
        }
      }
    }
    leg_coarse->launch(s, b, nullptr);
    ++leg_launches;
    MGLevel& L = lv[0];
    smooth_lattice(s, L, b, nullptr, x, degree, identity_rows, nullptr, lv[1].x.p, nullptr);
    return true;
  }
  if (legs_kind == 2) {
    for (auto& p : legs_down) { p->launch(s, b, nullptr); ++leg_launches; }
    leg_tail->launch(s, b, x);
    ++leg_launches;
    for (size_t k = legs_up.size(); k-- > 0;) { legs_up[k]->launch(s, b, x); ++leg_launches; }
    return true;
  }
  return false;
}

}  // namespace nsfem
