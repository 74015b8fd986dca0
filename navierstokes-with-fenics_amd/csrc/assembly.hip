// Element integration kernels for P2/P1 Taylor-Hood triangles on gfx950.
//
// Replaces the FFC/uflacs generated tabulate_tensor kernels + dolfin::Assembler
// cell loops that the reference triggers at source/ns_ipcs_solver.py:126-141,
// 160-169,183-193 and source/ns_bdf_solver.py:68-94.
//
// Layout choices (MI355X):
//   * one thread per (cell, test function i): consecutive lanes = consecutive
//     cells, so every per-cell array (vertex coords, dof maps, slot maps) is stored
//     SoA [k][n_cells] and read fully coalesced (512 B per wave-instruction);
//   * reference basis tables live in __constant__ memory: the quadrature index is
//     wave-uniform, so they arrive through the scalar cache (s_load) and occupy no
//     VGPRs / LDS;
//   * element tensors stay in registers (<= 24 fp64 accumulators per thread);
//   * no atomics: every element tensor is written with plain stores to an element
//     buffer indexed by (cell, i, j); a second kernel sums, for every CSR slot / dof,
//     the entries listed in a precomputed inverted index in a fixed order.  Results
//     are bitwise reproducible and each CSR value is written exactly once.
//
// All integrands are polynomials of degree <= 5 on affine cells, so the 7-point
// degree-5 rule reproduces FEniCS' integrals to round-off.
#include "nsfem_internal.hpp"

namespace nsfem {

__constant__ QuadTables c_q;

void upload_quad_tables(const QuadTables& t) {
  NSFEM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_q), &t, sizeof(QuadTables)));
}

struct CellGeo {
  double ji00, ji01, ji10, ji11;   // J^{-1}
  double adet;
};

__device__ __forceinline__ CellGeo load_geo(const double* __restrict__ vx, int nc, int c) {
  // (products fuse with the sums of their own statement only: every kernel that inlines this gets the same bits)
#pragma clang fp contract(on)
  const double x0 = vx[c], y0 = vx[(size_t)nc + c];
  const double x1 = vx[(size_t)2 * nc + c], y1 = vx[(size_t)3 * nc + c];
  const double x2 = vx[(size_t)4 * nc + c], y2 = vx[(size_t)5 * nc + c];
  const double j00 = x1 - x0, j01 = x2 - x0, j10 = y1 - y0, j11 = y2 - y0;
  const double det = j00 * j11 - j01 * j10;
  const double id = 1.0 / det;
  CellGeo g;
  g.ji00 = j11 * id;
  g.ji01 = -j01 * id;
  g.ji10 = -j10 * id;
  g.ji11 = j00 * id;
  g.adet = fabs(det);
  return g;
}

// physical gradient of a reference gradient (dr0, dr1): g_a = sum_b Jinv[b][a] dr_b
__device__ __forceinline__ void phys(const CellGeo& g, double dr0, double dr1, double& gx,
                                     double& gy) {
#pragma clang fp contract(on)
  gx = g.ji00 * dr0 + g.ji10 * dr1;
  gy = g.ji01 * dr0 + g.ji11 * dr1;
}


// ------------------------------------------------------------------ scalar P2
__global__ __launch_bounds__(256) void k_p2_scalar(int nc, const double* __restrict__ vx,
                                                   const int32_t* __restrict__ slot,
                                                   double* __restrict__ mass,
                                                   double* __restrict__ stiff) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nc * 6) return;
  const int i = (int)(t / nc), c = (int)(t % nc);
  const CellGeo g = load_geo(vx, nc, c);
  double m[6], k[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) m[j] = k[j] = 0.0;
  for (int q = 0; q < 7; ++q) {
    const double w = c_q.w[q] * g.adet;
    double gix, giy;
    phys(g, c_q.dphi2[q][i][0], c_q.dphi2[q][i][1], gix, giy);
    const double pi = c_q.phi2[q][i] * w;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double gjx, gjy;
      phys(g, c_q.dphi2[q][j][0], c_q.dphi2[q][j][1], gjx, gjy);
      m[j] += pi * c_q.phi2[q][j];
      k[j] += w * (gix * gjx + giy * gjy);
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const size_t s = ((size_t)c * 6 + i) * 6 + j;     // element-buffer index (cell, i, j)
    mass[s] = m[j];
    stiff[s] = k[j];
  }
}

// ------------------------------------------------------------------ scalar P1
__global__ __launch_bounds__(256) void k_p1_scalar(int nc, const double* __restrict__ vx,
                                                   const int32_t* __restrict__ slot,
                                                   double* __restrict__ stiff,
                                                   double* __restrict__ mass) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nc * 3) return;
  const int i = (int)(t / nc), c = (int)(t % nc);
  const CellGeo g = load_geo(vx, nc, c);
  const double dl[3][2] = {{-1.0, -1.0}, {1.0, 0.0}, {0.0, 1.0}};
  double gix, giy;
  phys(g, dl[i][0], dl[i][1], gix, giy);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    double gjx, gjy;
    phys(g, dl[j][0], dl[j][1], gjx, gjy);
    double m = 0.0;
    for (int q = 0; q < 7; ++q) m += c_q.w[q] * c_q.phi1[q][i] * c_q.phi1[q][j];
    const size_t s = ((size_t)c * 3 + i) * 3 + j;
    stiff[s] = 0.5 * g.adet * (gix * gjx + giy * gjy);
    mass[s] = m * g.adet;
  }
}

// --------------------------------------------------- divergence (P1 rows, 1x2)
__global__ __launch_bounds__(256) void k_div(int nc, const double* __restrict__ vx,
                                             const int32_t* __restrict__ slot12,
                                             double* __restrict__ div) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nc * 3) return;
  const int i = (int)(t / nc), c = (int)(t % nc);
  const CellGeo g = load_geo(vx, nc, c);
  double dx[6], dy[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) dx[j] = dy[j] = 0.0;
  for (int q = 0; q < 7; ++q) {
    const double wp = c_q.w[q] * g.adet * c_q.phi1[q][i];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double gjx, gjy;
      phys(g, c_q.dphi2[q][j][0], c_q.dphi2[q][j][1], gjx, gjy);
      dx[j] += wp * gjx;
      dy[j] += wp * gjy;
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const size_t s = ((size_t)c * 3 + i) * 6 + j;
    div[2 * s] = dx[j];
    div[2 * s + 1] = dy[j];
  }
}

// --------------------------- gradient / transposed divergence (P2 rows, 2x1)
__global__ __launch_bounds__(256) void k_grad(int nc, const double* __restrict__ vx,
                                              const int32_t* __restrict__ slot21,
                                              double* __restrict__ grad,
                                              double* __restrict__ divT) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nc * 6) return;
  const int i = (int)(t / nc), c = (int)(t % nc);
  const CellGeo g = load_geo(vx, nc, c);
  const double dl[3][2] = {{-1.0, -1.0}, {1.0, 0.0}, {0.0, 1.0}};
  double gr[3][2], dt[3][2];
#pragma unroll
  for (int j = 0; j < 3; ++j) gr[j][0] = gr[j][1] = dt[j][0] = dt[j][1] = 0.0;
  for (int q = 0; q < 7; ++q) {
    const double w = c_q.w[q] * g.adet;
    const double pi = c_q.phi2[q][i] * w;
    double gix, giy;
    phys(g, c_q.dphi2[q][i][0], c_q.dphi2[q][i][1], gix, giy);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double gjx, gjy;
      phys(g, dl[j][0], dl[j][1], gjx, gjy);
      gr[j][0] += pi * gjx;                       // int phi_i d_x psi_j
      gr[j][1] += pi * gjy;
      dt[j][0] += w * gix * c_q.phi1[q][j];       // int d_x phi_i psi_j
      dt[j][1] += w * giy * c_q.phi1[q][j];
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const size_t s = ((size_t)c * 6 + i) * 3 + j;
    grad[2 * s] = gr[j][0];
    grad[2 * s + 1] = gr[j][1];
    divT[2 * s] = dt[j][0];
    divT[2 * s + 1] = dt[j][1];
  }
}

// ----------- traction-form extra block  E[(i,a),(j,b)] = int d_b phi_i d_a phi_j
__global__ __launch_bounds__(256) void k_visc_extra(int nc, const double* __restrict__ vx,
                                                    const int32_t* __restrict__ slot,
                                                    double* __restrict__ E) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nc * 6) return;
  const int i = (int)(t / nc), c = (int)(t % nc);
  const CellGeo g = load_geo(vx, nc, c);
  double acc[6][4];
#pragma unroll
  for (int j = 0; j < 6; ++j) acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0.0;
  for (int q = 0; q < 7; ++q) {
    const double w = c_q.w[q] * g.adet;
    double gi[2];
    phys(g, c_q.dphi2[q][i][0], c_q.dphi2[q][i][1], gi[0], gi[1]);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double gj[2];
      phys(g, c_q.dphi2[q][j][0], c_q.dphi2[q][j][1], gj[0], gj[1]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[j][a * 2 + b] += w * gi[b] * gj[a];
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const size_t s = ((size_t)c * 6 + i) * 6 + j;
#pragma unroll
    for (int e = 0; e < 4; ++e) E[4 * s + e] = acc[j][e];
  }
}

// ----------------------------------------------------- J = L (x) I_2 + cvE * E
__global__ __launch_bounds__(256) void k_jac_init(int nnz, const double* __restrict__ L,
                                                  const double* __restrict__ E, double cvE,
                                                  double* __restrict__ J) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz;
       k += (int64_t)gridDim.x * blockDim.x) {
    const double l = L[k];
    double2 r0 = make_double2(l, 0.0), r1 = make_double2(0.0, l);
    if (E) {
      const double2 e0 = reinterpret_cast<const double2*>(E)[2 * k];
      const double2 e1 = reinterpret_cast<const double2*>(E)[2 * k + 1];
      r0.x += cvE * e0.x; r0.y += cvE * e0.y;
      r1.x += cvE * e1.x; r1.y += cvE * e1.y;
    }
    reinterpret_cast<double2*>(J)[2 * k] = r0;
    reinterpret_cast<double2*>(J)[2 * k + 1] = r1;
  }
}

// ---- convection blocks of the Newton (or Picard) matrix for the four weak forms of
// source/ns_solver_base.py:370-390 (Gateaux derivatives) / :478-499 (Picard linearisations).
// FORM: 0 standard, 1 rotational, 2 divergence, 3 skew-symmetric.  With u, G = grad u,
// div, curl at the quadrature point, test (phi_i, a), trial (phi_j, b):
//   standard    phi_i [ (u.g_j) d_ab + phi_j G_ab ]                 (Picard: first term)
//   divergence  standard + 1/2 phi_i [ g_j,b u_a + div phi_j d_ab ] (Picard: 1st + 4th term)
//   skew        1/2 standard - 1/2 [ g_i,b phi_j u_a + (u.g_i) phi_j d_ab ]
//   rotational  phi_i e_a [ dcurl_j,b u_(1-a) + curl phi_j d_(b,1-a) ],  e = (-1, +1)
template <int FORM, bool PICARD>
__global__ __launch_bounds__(256) void k_conv_jac(int nc, const double* __restrict__ vx,
                                                  const int32_t* __restrict__ p2,
                                                  const double* __restrict__ u, double cc,
                                                  double* __restrict__ ebuf) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nc * 6) return;
  const int i = (int)(t / nc), c = (int)(t % nc);
  const CellGeo g = load_geo(vx, nc, c);
  double ux[6], uy[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int node = p2[(size_t)k * nc + c];
    const double2 v = reinterpret_cast<const double2*>(u)[node];
    ux[k] = v.x;
    uy[k] = v.y;
  }
  double acc[6][4];
#pragma unroll
  for (int j = 0; j < 6; ++j) acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0.0;
  for (int q = 0; q < 7; ++q) {
    double gx[6], gy[6];
    double uqx = 0.0, uqy = 0.0, g00 = 0.0, g01 = 0.0, g10 = 0.0, g11 = 0.0;
    double gix = 0.0, giy = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      phys(g, c_q.dphi2[q][k][0], c_q.dphi2[q][k][1], gx[k], gy[k]);
      const double ph = c_q.phi2[q][k];
      uqx += ph * ux[k];
      uqy += ph * uy[k];
      g00 += gx[k] * ux[k];   // d_x u_x
      g01 += gy[k] * ux[k];   // d_y u_x
      g10 += gx[k] * uy[k];   // d_x u_y
      g11 += gy[k] * uy[k];   // d_y u_y
      if (k == i) { gix = gx[k]; giy = gy[k]; }
    }
    const double w = c_q.w[q] * g.adet * cc;
    const double wpi = w * c_q.phi2[q][i];
    const double div = g00 + g11, curl = g10 - g01;
    const double udgi = uqx * gix + uqy * giy;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const double udg = uqx * gx[j] + uqy * gy[j];
      const double pj = c_q.phi2[q][j];
      if (FORM == 0) {
        acc[j][0] += wpi * (udg + (PICARD ? 0.0 : pj * g00));
        acc[j][3] += wpi * (udg + (PICARD ? 0.0 : pj * g11));
        if (!PICARD) {
          acc[j][1] += wpi * (pj * g01);
          acc[j][2] += wpi * (pj * g10);
        }
      } else if (FORM == 2) {
        const double hd = 0.5 * div * pj;
        acc[j][0] += wpi * (udg + hd + (PICARD ? 0.0 : pj * g00 + 0.5 * gx[j] * uqx));
        acc[j][3] += wpi * (udg + hd + (PICARD ? 0.0 : pj * g11 + 0.5 * gy[j] * uqy));
        if (!PICARD) {
          acc[j][1] += wpi * (pj * g01 + 0.5 * gy[j] * uqx);
          acc[j][2] += wpi * (pj * g10 + 0.5 * gx[j] * uqy);
        }
      } else if (FORM == 3) {
        const double sk = 0.5 * (wpi * udg - w * udgi * pj);
        acc[j][0] += sk;
        acc[j][3] += sk;
        if (!PICARD) {
          const double hw = 0.5 * w * pj;
          acc[j][0] += 0.5 * wpi * pj * g00 - hw * gix * uqx;
          acc[j][1] += 0.5 * wpi * pj * g01 - hw * giy * uqx;
          acc[j][2] += 0.5 * wpi * pj * g10 - hw * gix * uqy;
          acc[j][3] += 0.5 * wpi * pj * g11 - hw * giy * uqy;
        }
      } else {   // rotational: rows a = 0 (e = -1, u_1) and a = 1 (e = +1, u_0)
        const double cj = curl * pj;
        acc[j][1] += wpi * (-cj);          // a = 0, b = 1 = 1 - a
        acc[j][2] += wpi * (cj);           // a = 1, b = 0
        if (!PICARD) {
          // dcurl_j,0 = -d_y phi_j ; dcurl_j,1 = d_x phi_j
          acc[j][0] += wpi * (gy[j] * uqy);
          acc[j][1] += wpi * (-gx[j] * uqy);
          acc[j][2] += wpi * (-gy[j] * uqx);
          acc[j][3] += wpi * (gx[j] * uqx);
        }
      }
    }
  }
  // plain stores of the 6 blocks of row i: ebuf[cell][i][j][4]
  double2* out = reinterpret_cast<double2*>(ebuf) + ((size_t)c * 36 + (size_t)i * 6) * 2;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    out[2 * j] = make_double2(acc[j][0], acc[j][1]);
    out[2 * j + 1] = make_double2(acc[j][2], acc[j][3]);
  }
}

// J[s] = L[s] (x) I_2 + cvE * E[s] + sum of the element blocks scattered to slot s, summed
// in ascending (cell, i, j) order: deterministic, every value written exactly once
__global__ __launch_bounds__(256) void k_jac_gather(int nnz, const int32_t* __restrict__ cptr,
                                                    const int32_t* __restrict__ cidx,
                                                    const double* __restrict__ ebuf,
                                                    const double* __restrict__ L,
                                                    const double* __restrict__ E, double cvE,
                                                    double* __restrict__ J) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nnz) return;
  const double l = L[s];
  double2 r0 = make_double2(l, 0.0), r1 = make_double2(0.0, l);
  if (E) {
    const double2 e0 = reinterpret_cast<const double2*>(E)[2 * (size_t)s];
    const double2 e1 = reinterpret_cast<const double2*>(E)[2 * (size_t)s + 1];
    r0.x += cvE * e0.x; r0.y += cvE * e0.y;
    r1.x += cvE * e1.x; r1.y += cvE * e1.y;
  }
  for (int k = cptr[s]; k < cptr[s + 1]; ++k) {
    const double2* src = reinterpret_cast<const double2*>(ebuf) + 2 * (size_t)cidx[k];
    const double2 a = src[0], b = src[1];
    r0.x += a.x; r0.y += a.y;
    r1.x += b.x; r1.y += b.y;
  }
  reinterpret_cast<double2*>(J)[2 * (size_t)s] = r0;
  reinterpret_cast<double2*>(J)[2 * (size_t)s + 1] = r1;
}

// ---- CFL diagnostic (reference source/ns_problem.py:554-587): cell-local L2 projection onto
// DG2 of  degree |u| k / h  (degree = 2, h = dolfin CellDiameter = circumdiameter) with the
// degree-4 rule, then the max-norm of the coefficients.  One thread per cell, one partial max
// per block.
__global__ __launch_bounds__(256) void k_cfl(int nc, const double* __restrict__ vx,
                                             const int32_t* __restrict__ p2,
                                             const double* __restrict__ u, double scale,
                                             double* __restrict__ parts) {
  __shared__ double sh[4];
  double best = 0.0;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
    const double x0 = vx[c], y0 = vx[(size_t)nc + c];
    const double x1 = vx[(size_t)2 * nc + c], y1 = vx[(size_t)3 * nc + c];
    const double x2 = vx[(size_t)4 * nc + c], y2 = vx[(size_t)5 * nc + c];
    const double la = sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
    const double lb = sqrt((x0 - x2) * (x0 - x2) + (y0 - y2) * (y0 - y2));
    const double lc = sqrt((x0 - x1) * (x0 - x1) + (y0 - y1) * (y0 - y1));
    const double area2 = fabs((x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0));
    const double h = la * lb * lc / area2;                  // 2 R = abc / (2 A)
    double ux[6], uy[6], f[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double2 v = reinterpret_cast<const double2*>(u)[p2[(size_t)k * nc + c]];
      ux[k] = v.x;
      uy[k] = v.y;
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        a += c_q.cfl_phi[q][k] * ux[k];
        b += c_q.cfl_phi[q][k] * uy[k];
      }
      f[q] = scale * sqrt(a * a + b * b) / h;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double ci = 0.0;
#pragma unroll
      for (int q = 0; q < 6; ++q) ci += c_q.cfl_inv[i][q] * f[q];
      best = fmax(best, fabs(ci));
    }
  }
  for (int off = 32; off > 0; off >>= 1) best = fmax(best, __shfl_xor(best, off));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) parts[blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

// nodal values of e_z x x = (-y, x) at the P2 nodes (vertices and edge midpoints of the affine
// cells; every cell sharing a node writes the same value)
__global__ __launch_bounds__(256) void k_rot_field(int nc, const double* __restrict__ vx,
                                                   const int32_t* __restrict__ p2,
                                                   double* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  double x[3], y[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    x[k] = vx[(size_t)(2 * k) * nc + c];
    y[k] = vx[(size_t)(2 * k + 1) * nc + c];
  }
  const int pr[3][2] = {{1, 2}, {0, 2}, {0, 1}};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double xn = k < 3 ? x[k] : 0.5 * (x[pr[k - 3][0]] + x[pr[k - 3][1]]);
    const double yn = k < 3 ? y[k] : 0.5 * (y[pr[k - 3][0]] + y[pr[k - 3][1]]);
    reinterpret_cast<double2*>(out)[p2[(size_t)k * nc + c]] = make_double2(-yn, xn);
  }
}
void launch_rot_field(hipStream_t s, const MeshDev& m, double* out) {
  hipLaunchKernelGGL(k_rot_field, dim3((m.n_cells + 255) / 256), dim3(256), 0, s, m.n_cells, m.vx.p,
                     m.p2.p, out);
  NSFEM_HIP(hipGetLastError());
}

void launch_cfl(hipStream_t s, const MeshDev& m, const double* u, double scale, double* parts,
                int n_parts) {
  hipLaunchKernelGGL(k_cfl, dim3(n_parts), dim3(256), 0, s, m.n_cells, m.vx.p, m.p2.p, u, scale,
                     parts);
  NSFEM_HIP(hipGetLastError());
}

// ---- convection residual / linearised action, one thread per CELL (the 2D counterpart of
// k3_conv_cell): u (and the direction v) at the 6 nodes in registers, u_q and grad u_q formed once
// per quadrature point for all 6 test functions.
//   LIN = 0  r_(i,a) = int c(u)_a phi_i ;  LIN = 1  Newton matrix times v ;  LIN = 2  Picard matrix
//   times v.  FORM as in k_conv_jac (0 standard, 1 rotational, 2 divergence, 3 skew-symmetric).
// element vector of one cell: u (and the direction w) at the 6 nodes in registers; shared by the one-thread-per-cell
// kernel below and the lattice kernel k_jac_lattice (same arithmetic, bit for bit)
template <int FORM, int LIN>
__device__ __forceinline__ void conv_cell_eval(const CellGeo& g, const double* ux, const double* uy,
                                               const double* wx, const double* wy, double cc,
                                               double* rx, double* ry) {
  // contraction inside statements only (not across them, which depends on the surrounding kernel): the two kernels
  // that inline this function produce the same element vectors bit for bit
#pragma clang fp contract(on)
#pragma unroll
  for (int i = 0; i < 6; ++i) rx[i] = ry[i] = 0.0;
  for (int q = 0; q < 7; ++q) {
    double gx[6], gy[6];
    double uqx = 0.0, uqy = 0.0, g00 = 0.0, g01 = 0.0, g10 = 0.0, g11 = 0.0;
    double vqx = 0.0, vqy = 0.0, h00 = 0.0, h01 = 0.0, h10 = 0.0, h11 = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      phys(g, c_q.dphi2[q][k][0], c_q.dphi2[q][k][1], gx[k], gy[k]);
      const double ph = c_q.phi2[q][k];
      uqx += ph * ux[k];
      uqy += ph * uy[k];
      g00 += gx[k] * ux[k];
      g01 += gy[k] * ux[k];
      g10 += gx[k] * uy[k];
      g11 += gy[k] * uy[k];
      if (LIN) {
        vqx += ph * wx[k];
        vqy += ph * wy[k];
        h00 += gx[k] * wx[k];
        h01 += gy[k] * wx[k];
        h10 += gx[k] * wy[k];
        h11 += gy[k] * wy[k];
      }
    }
    const double w = c_q.w[q] * g.adet * cc;
    double fx, fy;                       // coefficient of phi_i
    if (LIN == 0) {
      const double ax = g00 * uqx + g01 * uqy, ay = g10 * uqx + g11 * uqy;       // (grad u) u
      if (FORM == 1) {
        const double curl = g10 - g01;
        fx = -curl * uqy;
        fy = curl * uqx;
      } else {
        fx = ax;
        fy = ay;
        if (FORM == 2) {
          const double hd = 0.5 * (g00 + g11);
          fx += hd * uqx;
          fy += hd * uqy;
        }
      }
    } else {
      const double gvu_x = h00 * uqx + h01 * uqy, gvu_y = h10 * uqx + h11 * uqy;   // (grad v) u
      const double guv_x = g00 * vqx + g01 * vqy, guv_y = g10 * vqx + g11 * vqy;   // (grad u) v
      if (FORM == 1) {
        const double cu = g10 - g01, cv = h10 - h01;
        fx = -cu * vqy;
        fy = cu * vqx;
        if (LIN == 1) {
          fx += -cv * uqy;
          fy += cv * uqx;
        }
      } else {
        fx = gvu_x + (LIN == 1 ? guv_x : 0.0);
        fy = gvu_y + (LIN == 1 ? guv_y : 0.0);
        if (FORM == 2) {
          const double hdu = 0.5 * (g00 + g11);
          fx += hdu * vqx;
          fy += hdu * vqy;
          if (LIN == 1) {
            const double hdv = 0.5 * (h00 + h11);
            fx += hdv * uqx;
            fy += hdv * uqy;
          }
        }
      }
    }
    if (FORM == 3) { fx *= 0.5; fy *= 0.5; }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double wpi = w * c_q.phi2[q][i];
      rx[i] += wpi * fx;
      ry[i] += wpi * fy;
      if (FORM == 3) {
        const double ugi = uqx * gx[i] + uqy * gy[i];
        if (LIN == 0) {
          rx[i] -= 0.5 * w * ugi * uqx;
          ry[i] -= 0.5 * w * ugi * uqy;
        } else {
          const double vgi = vqx * gx[i] + vqy * gy[i];
          rx[i] -= 0.5 * w * (ugi * vqx + (LIN == 1 ? vgi * uqx : 0.0));
          ry[i] -= 0.5 * w * (ugi * vqy + (LIN == 1 ? vgi * uqy : 0.0));
        }
      }
    }
  }
}

template <int FORM, int LIN>
__global__ __launch_bounds__(256) void k_conv_cell(int nc, const double* __restrict__ vx,
                                                   const int32_t* __restrict__ p2,
                                                   const double* __restrict__ u,
                                                   const double* __restrict__ v, double cc,
                                                   const int32_t* __restrict__ ndst,
                                                   double* __restrict__ rbuf) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const CellGeo g = load_geo(vx, nc, c);
  double ux[6], uy[6], wx[LIN ? 6 : 1], wy[LIN ? 6 : 1];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int node = p2[(size_t)k * nc + c];
    const double2 a = reinterpret_cast<const double2*>(u)[node];
    ux[k] = a.x;
    uy[k] = a.y;
    if (LIN) {
      const double2 b = reinterpret_cast<const double2*>(v)[node];
      wx[k] = b.x;
      wy[k] = b.y;
    }
  }
  double rx[6], ry[6];
  conv_cell_eval<FORM, LIN>(g, ux, uy, wx, wy, cc, rx, ry);
  // node-sorted element buffer: entry (c, i) lands inside the contiguous run of its node
  double2* out = reinterpret_cast<double2*>(rbuf);
#pragma unroll
  for (int i = 0; i < 6; ++i) out[ndst[(size_t)i * nc + c]] = make_double2(rx[i], ry[i]);
}

// b[(node, a)] += sum of the element vectors of the cells around the node: the contiguous run
// nptr[n] .. nptr[n + 1] of the node-sorted buffer, summed in ascending (cell) order
__global__ __launch_bounds__(256) void k_res_gather(int n_nodes, const int32_t* __restrict__ nptr,
                                                    const double* __restrict__ rbuf,
                                                    const uint8_t* __restrict__ skip,
                                                    double* __restrict__ b) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_nodes) return;
  bool sx = false, sy = false;          // flagged entries (Dirichlet / ghost rows) stay untouched
  if (skip) {
    sx = skip[2 * (size_t)n] != 0;
    sy = skip[2 * (size_t)n + 1] != 0;
    if (sx && sy) return;
  }
  const double2* __restrict__ rb = reinterpret_cast<const double2*>(rbuf);
  double2 acc = reinterpret_cast<double2*>(b)[n];
  int k = nptr[n];
  const int e = nptr[n + 1];
  for (; k + 4 <= e; k += 4) {          // four independent loads in flight per lane
    const double2 v0 = rb[k], v1 = rb[k + 1], v2 = rb[k + 2], v3 = rb[k + 3];
    acc.x += v0.x; acc.y += v0.y;
    acc.x += v1.x; acc.y += v1.y;
    acc.x += v2.x; acc.y += v2.y;
    acc.x += v3.x; acc.y += v3.y;
  }
  for (; k < e; ++k) {
    const double2 v = rb[k];
    acc.x += v.x;
    acc.y += v.y;
  }
  if (sx) acc.x = reinterpret_cast<double2*>(b)[n].x;
  if (sy) acc.y = reinterpret_cast<double2*>(b)[n].y;
  reinterpret_cast<double2*>(b)[n] = acc;
}

// vals[slot][bs] = sum over the slot's sources of ebuf[source][bs]  (fixed order)
__global__ __launch_bounds__(256) void k_gather_vals(int nnz, int bs,
                                                     const int32_t* __restrict__ cptr,
                                                     const int32_t* __restrict__ cidx,
                                                     const double* __restrict__ ebuf,
                                                     double* __restrict__ vals) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nnz * bs) return;
  const int s = (int)(t / bs), e = (int)(t % bs);
  double acc = 0.0;
  for (int k = cptr[s]; k < cptr[s + 1]; ++k) acc += ebuf[(size_t)cidx[k] * bs + e];
  vals[t] = acc;
}

// ------------------------------------------------------------- launch wrappers
static inline int grid_for(int64_t n) { return (int)((n + kBlock - 1) / kBlock); }

void gather_vals(hipStream_t s, const Pattern& p, int bs, const double* ebuf, double* vals) {
  NSFEM_REQUIRE(p.cptr.p && p.cidx.p, "pattern has no inverted index");
  hipLaunchKernelGGL(k_gather_vals, dim3(grid_for((int64_t)p.nnz * bs)), dim3(kBlock), 0, s,
                     p.nnz, bs, p.cptr.p, p.cidx.p, ebuf, vals);
  NSFEM_HIP(hipGetLastError());
}

void launch_assemble_p2_scalar(hipStream_t s, const MeshDev& m, const Pattern& p22, double* mass,
                               double* stiff) {
  if (m.dim == 3) return assemble_p2_scalar_3d(s, m, p22, mass, stiff);
  DevBuf<double> tmp;
  tmp.alloc((size_t)m.n_cells * 72);
  double* t0 = tmp.p;
  double* t1 = tmp.p + (size_t)m.n_cells * 36;
  hipLaunchKernelGGL(k_p2_scalar, dim3(grid_for((int64_t)m.n_cells * 6)), dim3(kBlock), 0, s,
                     m.n_cells, m.vx.p, p22.slot.p, t0, t1);
  gather_vals(s, p22, 1, t0, mass);
  gather_vals(s, p22, 1, t1, stiff);
  NSFEM_HIP(hipStreamSynchronize(s));
}
void launch_assemble_p1_scalar(hipStream_t s, const MeshDev& m, const Pattern& p11, double* stiff,
                               double* mass) {
  if (m.dim == 3) return assemble_p1_scalar_3d(s, m, p11, stiff, mass);
  DevBuf<double> tmp;
  tmp.alloc((size_t)m.n_cells * 18);
  double* t0 = tmp.p;
  double* t1 = tmp.p + (size_t)m.n_cells * 9;
  hipLaunchKernelGGL(k_p1_scalar, dim3(grid_for((int64_t)m.n_cells * 3)), dim3(kBlock), 0, s,
                     m.n_cells, m.vx.p, p11.slot.p, t0, t1);
  gather_vals(s, p11, 1, t0, stiff);
  gather_vals(s, p11, 1, t1, mass);
  NSFEM_HIP(hipStreamSynchronize(s));
}
void launch_assemble_div_grad(hipStream_t s, const MeshDev& m, const Pattern& p12,
                              const Pattern& p21, double* div, double* grad, double* divT) {
  if (m.dim == 3) return assemble_div_grad_3d(s, m, p12, p21, div, grad, divT);
  DevBuf<double> tmp;
  tmp.alloc((size_t)m.n_cells * 36 * 3);
  double* t0 = tmp.p;
  double* t1 = tmp.p + (size_t)m.n_cells * 36;
  double* t2 = tmp.p + (size_t)m.n_cells * 72;
  hipLaunchKernelGGL(k_div, dim3(grid_for((int64_t)m.n_cells * 3)), dim3(kBlock), 0, s, m.n_cells,
                     m.vx.p, p12.slot.p, t0);
  hipLaunchKernelGGL(k_grad, dim3(grid_for((int64_t)m.n_cells * 6)), dim3(kBlock), 0, s, m.n_cells,
                     m.vx.p, p21.slot.p, t1, t2);
  gather_vals(s, p12, 2, t0, div);
  gather_vals(s, p21, 2, t1, grad);
  gather_vals(s, p21, 2, t2, divT);
  NSFEM_HIP(hipStreamSynchronize(s));
}
void launch_assemble_viscous_extra(hipStream_t s, const MeshDev& m, const Pattern& p22,
                                   double* extra) {
  if (m.dim == 3) return assemble_viscous_extra_3d(s, m, p22, extra);
  DevBuf<double> tmp;
  tmp.alloc((size_t)m.n_cells * 144);
  hipLaunchKernelGGL(k_visc_extra, dim3(grid_for((int64_t)m.n_cells * 6)), dim3(kBlock), 0, s,
                     m.n_cells, m.vx.p, p22.slot.p, tmp.p);
  gather_vals(s, p22, 4, tmp.p, extra);
  NSFEM_HIP(hipStreamSynchronize(s));
}
void launch_jacobian_init(hipStream_t s, int nnz, const double* L, const double* E, double cvE,
                          double* J) {
  int grid = grid_for(nnz);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_jac_init, dim3(grid), dim3(kBlock), 0, s, nnz, L, E, cvE, J);
  NSFEM_HIP(hipGetLastError());
}
void launch_convection_jacobian(hipStream_t s, const MeshDev& m, const Pattern& p22,
                                const double* u, double cc, const double* L, const double* E,
                                double cvE, double* J, int form, bool picard) {
  if (m.dim == 3) return convection_jacobian_3d(s, m, p22, u, cc, L, E, cvE, J, form, picard);
  const dim3 grid(grid_for((int64_t)m.n_cells * 6)), block(kBlock);
#define NSFEM_CJ(F, P) \
  hipLaunchKernelGGL((k_conv_jac<F, P>), grid, block, 0, s, m.n_cells, m.vx.p, m.p2.p, u, cc, m.ebuf.p)
  switch (form * 2 + (picard ? 1 : 0)) {
    case 0: NSFEM_CJ(0, false); break;
    case 1: NSFEM_CJ(0, true); break;
    case 2: NSFEM_CJ(1, false); break;
    case 3: NSFEM_CJ(1, true); break;
    case 4: NSFEM_CJ(2, false); break;
    case 5: NSFEM_CJ(2, true); break;
    case 6: NSFEM_CJ(3, false); break;
    case 7: NSFEM_CJ(3, true); break;
    default: throw Error(NSFEM_ERR_ARG, "unknown convective form");
  }
#undef NSFEM_CJ
  hipLaunchKernelGGL(k_jac_gather, dim3(grid_for(p22.nnz)), dim3(kBlock), 0, s, p22.nnz,
                     p22.cptr.p, p22.cidx.p, m.ebuf.p, L, E, cvE, J);
  NSFEM_HIP(hipGetLastError());
}
template <int LIN>
static void launch_conv_cell(hipStream_t s, const MeshDev& m, const double* u, const double* v,
                             double cc, int form) {
  const dim3 grid(grid_for(m.n_cells)), block(kBlock);
#define NSFEM_CC(F) \
  hipLaunchKernelGGL((k_conv_cell<F, LIN>), grid, block, 0, s, m.n_cells, m.vx.p, m.p2.p, u, v, cc, m.ndst.p, m.rbuf.p)
  switch (form) {
    case 0: NSFEM_CC(0); break;
    case 1: NSFEM_CC(1); break;
    case 2: NSFEM_CC(2); break;
    case 3: NSFEM_CC(3); break;
    default: throw Error(NSFEM_ERR_ARG, "unknown convective form");
  }
#undef NSFEM_CC
  NSFEM_HIP(hipGetLastError());
}

void launch_convection_residual(hipStream_t s, const MeshDev& m, const double* u, double cc,
                                double* b, int form) {
  if (m.dim == 3) return convection_residual_3d(s, m, u, cc, b, form);
  launch_conv_cell<0>(s, m, u, nullptr, cc, form);
  hipLaunchKernelGGL(k_res_gather, dim3(grid_for(m.n_p2)), dim3(kBlock), 0, s, m.n_p2, m.nptr.p,
                     m.rbuf.p, (const uint8_t*)nullptr, b);
  NSFEM_HIP(hipGetLastError());
}

// y += c_c [d conv(u)/du] v (Newton) or its Picard linearisation: matrix-free action of the
// convection blocks of the velocity Jacobian
void launch_convection_action(hipStream_t s, const MeshDev& m, const double* u, const double* v,
                              double cc, double* y, int form, bool picard, const uint8_t* skipmask) {
  if (m.dim == 3) return convection_action_3d(s, m, u, v, cc, y, form, picard, skipmask);
  if (picard) launch_conv_cell<2>(s, m, u, v, cc, form);
  else launch_conv_cell<1>(s, m, u, v, cc, form);
  hipLaunchKernelGGL(k_res_gather, dim3(grid_for(m.n_p2)), dim3(kBlock), 0, s, m.n_p2, m.nptr.p,
                     m.rbuf.p, skipmask, y);
  NSFEM_HIP(hipGetLastError());
}

void launch_convection_cells(hipStream_t s, const MeshDev& m, const double* u, const double* v, double cc,
                             int form, bool picard) {
  NSFEM_REQUIRE(m.dim == 2, "launch_convection_cells: triangles only");
  if (picard) launch_conv_cell<2>(s, m, u, v, cc, form);
  else launch_conv_cell<1>(s, m, u, v, cc, form);
}


// ---------------------------------------------------------------- lattice Jacobian action
// y = L x + c_c [d conv(u)/du] x  (identity on the rows flagged in the mask) in ONE launch on 2D lattice meshes
// (rectangle_mesh numbering: cell 2 (sy nx + sx) + t, P2 node j W + i): the pair of launches it replaces --
// k_conv_cell (element vectors -> node-sorted buffer in HBM) and the dictionary product that sums the buffer --
// moves 96 B per cell out and in again, and the element kernel's scattered 16-byte stores and dependent index
// loads keep it at a third of its VALU bound.  Here a workgroup owns the nodes of 31 x 7 squares:
//   phase 0  u and x on the 65 x 17 nodes the 32 x 8 squares around them touch are staged in LDS (row-contiguous
//            global reads), together with the operator's stencil dictionary;
//   phase 1  one thread per owned node: acc = (L x)(node) from the LDS copy of x -- the dictionary entry's
//            products in entry order, exactly the sum of k_spmv_dict;
//   phase 2  one thread per cell (512 cells, the one-square ring around the owned nodes is recomputed by the
//            neighbouring workgroups): node values from LDS, conv_cell_eval in registers -- no index loads, the
//            node positions of a cell are a template of its type;
//   phase 3  six rounds: in round r every cell adds its element vector entry to the node it is the r-th
//            adjacent cell of (ascending cell number: at most one writer per node and round, and the node sums
//            come out in the order of the node-sorted buffer, bit for bit);
//   phase 4  y written once (coalesced 16-byte stores).
// Algorithmic bytes per launch: u, x read, y written (48 B per node), 48 B per cell of vertex coordinates, one
// byte per node of dictionary ids and two of masks.
// tile shapes: SX x SY squares per workgroup (powers of two; one cell per thread, 2 SX SY threads), of which the
// workgroup owns the nodes of (SX - 1) x (SY - 1)

struct JacLatArgs {
  int nx, ny, W, H, nc;
  int ntx, ntiles;
  int tbase, tsplit, tskip;   // tile = tbase + (t < tsplit ? t : t + tskip), t < ntiles: the whole lattice, or the tile
                              // rows away from / next to the ghost lines of a partitioned strip (two launches around
                              // the halo exchange)
  int lmax, lp, n_st;   // lp: table row stride (lmax rounded up to 4, zero padded)
  int dbg;              // knock-out build only (NSFEM_KNOCKOUTS): 1 no element kernel, 2 no L product, 4 no node sums
  double cc;
  // per cell type t and local node k, packed 6 x 5 bits (k-th field): the node's lattice offset from the square's corner
  // node as dj * 3 + di, and the cell's rank among the cells around that node (ascending cell number).  Plain
  // integers on purpose: an array indexed by the lane's cell type turns into VECTOR loads from the kernel-argument
  // segment (38 per wave, ~14 us of the launch).
  int noff0, noff1, rank0, rank1;
};

template <int FORM, int LIN, int SX, int SY>
__global__ __launch_bounds__(2 * SX * SY) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_jac_lattice(JacLatArgs a, const double* __restrict__ vx, const double* __restrict__ u,
                   const double* __restrict__ x, const uint8_t* __restrict__ sid8,
                   const uint8_t* __restrict__ mask, const int32_t* __restrict__ slen,
                   const int32_t* __restrict__ spack, const double* __restrict__ sval,
                   const double* __restrict__ gadd, double* __restrict__ y, const double* __restrict__ ugeo) {
  // LIN = 0: the momentum residual  y = L u + g + c_c conv(u)  (x is not read, no mask: the Dirichlet rows are
  // set by the caller afterwards); g joins the node's sum after the L product and before the element vectors,
  // the order of the launches it replaces (product, axpby, k_conv_cell, k_res_gather)
  constexpr bool RES = LIN == 0;
  extern __shared__ __attribute__((aligned(16))) double sh_jl[];
  constexpr int NT = 2 * SX * SY;                                   // threads
  constexpr int kJlNW = 2 * SX + 1, kJlNH = 2 * SY + 1;             // staged node lines
  constexpr int kJlOX = 2 * (SX - 1), kJlOY = 2 * (SY - 1);         // owned node lines
  constexpr int LOGSX = SX == 32 ? 5 : SX == 16 ? 4 : 3;
  static_assert((1 << LOGSX) == SX, "SX: 8, 16 or 32");
  constexpr int NN = kJlNW * kJlNH;
  double2* __restrict__ su = reinterpret_cast<double2*>(sh_jl);
  double2* __restrict__ sx = RES ? su : su + NN;
  double2* __restrict__ sa = sx + NN;
  double* __restrict__ tv = reinterpret_cast<double*>(sa + NN);       // [n_st * lp]
  int* __restrict__ to = reinterpret_cast<int*>(tv + a.n_st * a.lp);
  int* __restrict__ tl = to + a.n_st * a.lp;
  // XCD x (workgroups b = x mod 8) walks a contiguous range of tiles: neighbouring tiles share their halo in L2
  const int per = (a.ntiles + 7) >> 3;
  const int tlin = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (tlin >= a.ntiles || NSFEM_KO(a.dbg & 256)) return;
  const int tile = a.tbase + (tlin < a.tsplit ? tlin : tlin + a.tskip);
  const int ty = tile / a.ntx, tx = tile - ty * a.ntx;
  const int i0 = tx * kJlOX - 2, j0 = ty * kJlOY - 2;               // lattice position of the LDS tile's corner
  const int tid = threadIdx.x;
  const double2* __restrict__ u2 = reinterpret_cast<const double2*>(u);
  const double2* __restrict__ x2 = reinterpret_cast<const double2*>(x);
  // ---- phase 0: every global load of the workgroup is requested before the first one is waited for
  constexpr int NLD = (NN + NT - 1) / NT;
  double2 uv[NLD], xv[NLD];
#pragma unroll
  for (int r = 0; r < NLD; ++r) {
    const int t = tid + r * NT;
    const int lj = t / kJlNW, li = t - lj * kJlNW;
    const int gi = i0 + li, gj = j0 + lj;
    uv[r] = make_double2(0.0, 0.0);
    xv[r] = uv[r];
    if (t < NN && gi >= 0 && gi < a.W && gj >= 0 && gj < a.H && !NSFEM_KO(a.dbg & 8)) {
      const size_t g = (size_t)gj * a.W + gi;
      uv[r] = u2[g];
      if (!RES) xv[r] = x2[g];
    }
  }
  // (this thread's cell, its two owned nodes)
  const int ct = tid & 1, sxl = (tid >> 1) & (SX - 1), syl = tid >> (1 + LOGSX);
  const int sqx = (i0 >> 1) + sxl, sqy = (j0 >> 1) + syl;          // (i0, j0 even; >> is an arithmetic shift)
  const bool cell = sqx >= 0 && sqx < a.nx && sqy >= 0 && sqy < a.ny;
  CellGeo geo;
  if (ugeo) {
    // uniform lattice: the two cell types' geometry through the scalar cache (wave-uniform addresses), picked per lane
    const double g0[5] = {ugeo[0], ugeo[1], ugeo[2], ugeo[3], ugeo[4]};
    const double g1[5] = {ugeo[5], ugeo[6], ugeo[7], ugeo[8], ugeo[9]};
    geo.ji00 = cell ? (ct ? g1[0] : g0[0]) : 1.0;
    geo.ji01 = cell ? (ct ? g1[1] : g0[1]) : 1.0;
    geo.ji10 = cell ? (ct ? g1[2] : g0[2]) : 1.0;
    geo.ji11 = cell ? (ct ? g1[3] : g0[3]) : 1.0;
    geo.adet = cell ? (ct ? g1[4] : g0[4]) : 1.0;
  } else if (cell && !NSFEM_KO(a.dbg & 16)) geo = load_geo(vx, a.nc, 2 * (sqy * a.nx + sqx) + ct);
  else geo.ji00 = geo.ji01 = geo.ji10 = geo.ji11 = geo.adet = 1.0;
  constexpr int OWN = kJlOX * kJlOY, NOWN = (OWN + NT - 1) / NT;
  int obase[NOWN], oent[NOWN], omask[NOWN];
  size_t onode[NOWN];
  double2 og[RES ? NOWN : 1];
#pragma unroll
  for (int r = 0; r < NOWN; ++r) {
    const int o = tid + r * NT;
    const int oj = o / kJlOX, oi = o - oj * kJlOX;
    const int gi = i0 + 2 + oi, gj = j0 + 2 + oj;
    obase[r] = -1;
    oent[r] = 0;
    omask[r] = 0;
    onode[r] = 0;
    if (o < OWN && gi < a.W && gj < a.H) {
      obase[r] = (oj + 2) * kJlNW + oi + 2;
      onode[r] = (size_t)gj * a.W + gi;
      oent[r] = sid8[onode[r]];
      if (RES) og[r] = reinterpret_cast<const double2*>(gadd)[onode[r]];
      else omask[r] = reinterpret_cast<const uint16_t*>(mask)[onode[r]];
    }
  }
  for (int t = tid; t < a.n_st * a.lp; t += NT) {
    const int e = t / a.lp, k = t - e * a.lp;
    const bool in = k < a.lmax;
    tv[t] = in ? sval[e * a.lmax + k] : 0.0;
    const int pk = in ? spack[e * a.lmax + k] : (8 * 32 + 8);
    to[t] = ((pk >> 5) - 8) * kJlNW + ((pk & 31) - 8);
  }
  if (tid < a.n_st) tl[tid] = slen[tid];
  if NSFEM_KO(a.dbg & 512) return;
#pragma unroll
  for (int r = 0; r < NLD; ++r) {
    const int t = tid + r * NT;
    if (t < NN) {
      su[t] = uv[r];
      if (!RES) sx[t] = xv[r];
    }
  }
  __syncthreads();
  if NSFEM_KO(a.dbg & 128) return;
  // ---- phase 1: (L x) on the owned nodes; four entries of the row in flight (rows are zero padded to 4).
  // (Measured and rejected: advancing the rows of the thread's two nodes together with the next trip's table
  // entries prefetched -- 51.1 instead of 48.3 us per launch at n = 512.)
#pragma unroll
  for (int r = 0; r < NOWN; ++r) {
    if (obase[r] < 0) continue;
    const int base = obase[r];
    const int L = NSFEM_KO(a.dbg & 2) ? 0 : tl[oent[r]];
    const int* __restrict__ op = to + oent[r] * a.lp;
    const double* __restrict__ vp = tv + oent[r] * a.lp;
    double ax = 0.0, ay = 0.0;
    for (int k = 0; k < L; k += 4) {
      const int4 o4 = *reinterpret_cast<const int4*>(op + k);
      const double2 va = *reinterpret_cast<const double2*>(vp + k), vb = *reinterpret_cast<const double2*>(vp + k + 2);
      const double2 x0 = sx[base + o4.x], x1 = sx[base + o4.y], x2_ = sx[base + o4.z], x3 = sx[base + o4.w];
      ax = fma(va.x, x0.x, ax); ay = fma(va.x, x0.y, ay);
      ax = fma(va.y, x1.x, ax); ay = fma(va.y, x1.y, ay);
      ax = fma(vb.x, x2_.x, ax); ay = fma(vb.x, x2_.y, ay);
      ax = fma(vb.y, x3.x, ax); ay = fma(vb.y, x3.y, ay);
    }
    if (RES) { ax += og[r].x; ay += og[r].y; }
    sa[base] = make_double2(ax, ay);
  }
  // ---- phase 2: the element vector of this thread's cell
  int nl[6];
  double rx[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, ry[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const int corner = 2 * syl * kJlNW + 2 * sxl;
  const int pno = a.noff0 ^ ((a.noff0 ^ a.noff1) & -ct), prk = a.rank0 ^ ((a.rank0 ^ a.rank1) & -ct);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int f = (pno >> (5 * k)) & 31;          // dj * 3 + di
    const int dj = (f * 11) >> 5;                 // f / 3 for f < 9
    nl[k] = corner + dj * kJlNW + (f - 3 * dj);
  }
  if (cell && !NSFEM_KO(a.dbg & 64)) {
    double ux[6], uy[6], wx[RES ? 1 : 6], wy[RES ? 1 : 6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double2 p = su[nl[k]];
      ux[k] = p.x; uy[k] = p.y;
      if (!RES) {
        const double2 q = sx[nl[k]];
        wx[k] = q.x; wy[k] = q.y;
      }
    }
    if NSFEM_KO(a.dbg & 1) {
#pragma unroll
      for (int k = 0; k < 6; ++k) { rx[k] = ux[k]; ry[k] = uy[k]; }
    } else {
      conv_cell_eval<FORM, LIN>(geo, ux, uy, wx, wy, a.cc, rx, ry);
    }
  }
  __syncthreads();
  // ---- phase 3: node sums in ascending cell order
#pragma unroll
  for (int r = 0; r < 6; ++r) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int rk = (prk >> (5 * k)) & 31;
      if (cell && rk == r && !NSFEM_KO(a.dbg & 4)) {
        double2 v = sa[nl[k]];
        v.x += rx[k];
        v.y += ry[k];
        sa[nl[k]] = v;
      }
    }
    __syncthreads();
  }
  // ---- phase 4
  double2* __restrict__ y2 = reinterpret_cast<double2*>(y);
#pragma unroll
  for (int r = 0; r < NOWN; ++r) {
    if (obase[r] < 0) continue;
    double2 v = sa[obase[r]];
    if (!RES) {
      const double2 xo = sx[obase[r]];
      // (flag 1: identity row; flag 2: ghost row of a partitioned strip -> 0, as the SpMV kernels write it)
      if (omask[r] & 0x00ff) v.x = (omask[r] & 0x0002) ? 0.0 : xo.x;
      if (omask[r] & 0xff00) v.y = (omask[r] & 0x0200) ? 0.0 : xo.y;
    }
    if (!NSFEM_KO(a.dbg & 32) || v.x == 1.2345) y2[onode[r]] = v;
  }
}

bool build_cell_lattice(const int32_t* p2map, int nc, int W, int H, CellLattice& cl) {
  cl.ok = false;
  cl.geo_uniform = false;
  cl.nx = cl.ny = cl.W = cl.H = 0;
  cl.tried = true;
  if (W < 7 || H < 7 || !(W & 1) || !(H & 1)) return false;
  const int nx = (W - 1) / 2, ny = (H - 1) / 2;
  if ((int64_t)2 * nx * ny != nc) return false;
  // the template of the two cell types: node positions relative to the square's corner node
  for (int t = 0; t < 2; ++t)
    for (int k = 0; k < 6; ++k) {
      const int node = p2map[(size_t)t * 6 + k];
      const int j = node / W, i = node - j * W;
      if (i > 2 || j > 2) return false;
      cl.di[t][k] = i;
      cl.dj[t][k] = j;
    }
  for (int sy = 0; sy < ny; ++sy)
    for (int sx = 0; sx < nx; ++sx)
      for (int t = 0; t < 2; ++t) {
        const size_t c = 2 * ((size_t)sy * nx + sx) + t;
        for (int k = 0; k < 6; ++k)
          if (p2map[c * 6 + k] != (2 * sy + cl.dj[t][k]) * W + 2 * sx + cl.di[t][k]) return false;
      }
  // rank of (t, k) among the cells around its node, taken at the interior square (1, 1)
  for (int t = 0; t < 2; ++t)
    for (int k = 0; k < 6; ++k) {
      const size_t c = 2 * ((size_t)1 * nx + 1) + t;
      const int node = p2map[c * 6 + k];
      int before = 0;
      for (int sy = 0; sy < 3; ++sy)
        for (int sx = 0; sx < 3; ++sx)
          for (int t2 = 0; t2 < 2; ++t2) {
            const size_t c2 = 2 * ((size_t)sy * nx + sx) + t2;
            if (c2 >= c) continue;
            for (int k2 = 0; k2 < 6; ++k2) before += p2map[c2 * 6 + k2] == node;
          }
      if (before > 5) return false;
      cl.rank[t][k] = before;
    }
  cl.nx = nx; cl.ny = ny; cl.W = W; cl.H = H;
  cl.ok = true;
  return true;
}

// is the geometry of every cell that of the first cell of its type, bit for bit?  (load_geo itself: what the kernels
// would compute.)  flag[0] != 0: not uniform; ref[2][5]: the two types' geometry
__global__ __launch_bounds__(256) void k_geo_uniform(int nc, const double* __restrict__ vx, double* __restrict__ ref,
                                                     int* __restrict__ flag) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const CellGeo g = load_geo(vx, nc, c), r = load_geo(vx, nc, c & 1);
  const bool same = g.ji00 == r.ji00 && g.ji01 == r.ji01 && g.ji10 == r.ji10 && g.ji11 == r.ji11 && g.adet == r.adet;
  if (!same) flag[0] = 1;
  if (c < 2) {
    double* o = ref + 5 * c;
    o[0] = g.ji00; o[1] = g.ji01; o[2] = g.ji10; o[3] = g.ji11; o[4] = g.adet;
  }
}
void check_uniform_geometry(hipStream_t s, MeshDev& m) {
  CellLattice& cl = m.cl;
  cl.geo_uniform = false;
  if (!cl.ok || m.dim != 2 || m.n_cells < 2) return;
  cl.ugeo.alloc(10);
  DevBuf<int> flag;
  flag.alloc(1);
  flag.zero(s);
  hipLaunchKernelGGL(k_geo_uniform, dim3((m.n_cells + 255) / 256), dim3(256), 0, s, m.n_cells, (const double*)m.vx.p,
                     cl.ugeo.p, flag.p);
  NSFEM_HIP(hipGetLastError());
  int h = 1;
  NSFEM_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
  NSFEM_HIP(hipStreamSynchronize(s));
  cl.geo_uniform = h == 0;
}

static bool g_jac_lattice_on = true;
static int g_jac_lattice_dbg = 0;
// tile shape in use: 0 = 32 x 8 squares (512 threads), 1 = 16 x 8 (256).  Measured at n = 512 (us per launch of the
// Newton action): 32 x 8: 48.3, 16 x 8: 50.4, 16 x 16: 50.5, 8 x 8: 55.5, 32 x 16 (1024 threads, one workgroup per CU):
// 62.7 -- the launch is bound by the latency chain of
// a wave (loads, barriers, LDS phases) at 4 waves per SIMD, not by the shape of the tile
static int g_jac_lattice_tile = 0;
static bool g_partitioned_lattice = true;
static bool g_jac_uniform_geo = true;                 // NSFEM_JL_UNIFORM_GEO=0: always load the cell coordinates
bool partitioned_lattice_kernels() { return g_partitioned_lattice; }
void refresh_assembly_switches() {
  const char* e = std::getenv("NSFEM_JAC_LATTICE");
  g_jac_lattice_on = e ? std::atoi(e) != 0 : true;
  e = std::getenv("NSFEM_JL_UNIFORM_GEO");
  g_jac_uniform_geo = e ? std::atoi(e) != 0 : true;
  e = std::getenv("NSFEM_PARTITIONED_LATTICE");       // 0: partitioned strips keep the one-step / multi-launch kernels
  g_partitioned_lattice = e ? std::atoi(e) != 0 : true;
#if NSFEM_KNOCKOUTS
  e = std::getenv("NSFEM_JL_DBG");
  g_jac_lattice_dbg = e ? std::atoi(e) : 0;
#endif
  e = std::getenv("NSFEM_JL_TILE");
  g_jac_lattice_tile = e ? std::max(0, std::min(1, std::atoi(e))) : 0;
}
static void jac_lattice_shape(int& sx, int& sy) {
  static const int shapes[2][2] = {{32, 8}, {16, 8}};
  sx = shapes[g_jac_lattice_tile][0];
  sy = shapes[g_jac_lattice_tile][1];
}
static size_t jac_lattice_lds(const StencilDict& d) {
  const size_t lp = (size_t)((d.lmax + 3) & ~3);
  int sx, sy;
  jac_lattice_shape(sx, sy);
  return (size_t)3 * (2 * sx + 1) * (2 * sy + 1) * sizeof(double2) + (size_t)d.n_stencils * lp * 12 +
         (size_t)d.n_stencils * 4 + 16;
}

bool jacobian_lattice_available(const MeshDev& m, const BlockMat& L) {
  const CellLattice& cl = m.cl;
  if (!g_jac_lattice_on || !cl.ok || m.dim != 2 || !L.dict_ready || !L.dict) return false;
  const StencilDict& d = *L.dict;
  return d.lat_w == cl.W && d.lat_h == cl.H && d.lat_r <= 2 && !d.rect && L.br == 1 && L.bc == 1 &&
         d.n_stencils <= 64 && jac_lattice_lds(d) <= (size_t)96 * 1024;
}
int64_t jacobian_lattice_bytes(const MeshDev& m) {
  // (uniform lattices: the cell geometry comes from 10 scalar loads, no coordinate stream)
  return (int64_t)m.n_p2 * (3 * 16 + 1 + 2) + ((g_jac_uniform_geo && m.cl.geo_uniform) ? 0 : (int64_t)m.n_cells * 48);
}

// lin: 0 residual (x, mask unused; gadd = g), 1 Newton action, 2 Picard action
// phase 0: every tile; 1: the tile rows that read no lattice line below `safe_lo` or from `safe_hi` on (the interior of
// a partitioned strip, launched under the halo exchange); 2: the other tile rows.  jacobian_lattice_split tells
// whether phases 1 / 2 exist for the given ghost lines
static bool lattice_tile_rows(const CellLattice& cl, int gh_lo, int gh_hi, int& nty, int& r0, int& r1) {
  int sx, sy;
  jac_lattice_shape(sx, sy);
  const int oy = 2 * (sy - 1);
  nty = (cl.H + oy - 1) / oy;
  // tile row ty reads the lattice lines ty * oy - 2 ... (ty + 1) * oy
  r0 = 0;
  while (r0 < nty && r0 * oy - 2 < gh_lo) ++r0;
  r1 = nty;
  while (r1 > r0 && r1 * oy >= cl.H - gh_hi) --r1;
  return r1 > r0 && (r0 > 0 || r1 < nty);
}
bool jacobian_lattice_split(const MeshDev& m, int gh_lo, int gh_hi) {
  int nty, r0, r1;
  return m.cl.ok && lattice_tile_rows(m.cl, gh_lo, gh_hi, nty, r0, r1);
}
static bool launch_lattice_cells(hipStream_t s, const MeshDev& m, const BlockMat& L, const double* u,
                                 const double* x, double cc, int form, int lin, const uint8_t* mask,
                                 const double* gadd, double* y, int phase = 0, int gh_lo = 0, int gh_hi = 0) {
  const CellLattice& cl = m.cl;
  if (!jacobian_lattice_available(m, L)) return false;
  const StencilDict& d = *L.dict;
  JacLatArgs a;
  a.nx = cl.nx; a.ny = cl.ny; a.W = cl.W; a.H = cl.H; a.nc = m.n_cells;
  int sx, sy;
  jac_lattice_shape(sx, sy);
  const int ox = 2 * (sx - 1), oy = 2 * (sy - 1);
  a.ntx = (cl.W + ox - 1) / ox;
  a.ntiles = a.ntx * ((cl.H + oy - 1) / oy);
  a.tbase = 0; a.tsplit = a.ntiles; a.tskip = 0;
  if (phase != 0) {
    int nty, r0, r1;
    if (!lattice_tile_rows(cl, gh_lo, gh_hi, nty, r0, r1)) return false;
    if (phase == 1) {
      a.tbase = r0 * a.ntx;
      a.ntiles = (r1 - r0) * a.ntx;
      a.tsplit = a.ntiles;
    } else {
      a.ntiles = (r0 + nty - r1) * a.ntx;
      a.tsplit = r0 * a.ntx;
      a.tskip = (r1 - r0) * a.ntx;
    }
    if (a.ntiles == 0) return true;
  }
  a.lmax = d.lmax; a.lp = (d.lmax + 3) & ~3; a.n_st = d.n_stencils;
  a.dbg = g_jac_lattice_dbg;
  a.cc = cc;
  int pk[2][2] = {{0, 0}, {0, 0}};
  for (int t = 0; t < 2; ++t)
    for (int k = 0; k < 6; ++k) {
      pk[0][t] |= (cl.dj[t][k] * 3 + cl.di[t][k]) << (5 * k);
      pk[1][t] |= cl.rank[t][k] << (5 * k);
    }
  a.noff0 = pk[0][0]; a.noff1 = pk[0][1]; a.rank0 = pk[1][0]; a.rank1 = pk[1][1];
  const size_t lds = jac_lattice_lds(d);
  const int grid = ((a.ntiles + 7) / 8) * 8;
#define NSFEM_JL_T(F, LIN, SX, SY)                                                                          \
  do {                                                                                                      \
    static bool attr = false;                                                                               \
    if (!attr) {                                                                                            \
      NSFEM_HIP(hipFuncSetAttribute((const void*)k_jac_lattice<F, LIN, SX, SY>,                             \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));                \
      attr = true;                                                                                          \
      if (std::getenv("NSFEM_JL_OCC")) {                                                                    \
        int nb = 0;                                                                                         \
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_jac_lattice<F, LIN, SX, SY>, \
                                                           2 * SX * SY, lds);                               \
        std::fprintf(stderr, "k_jac_lattice<%d x %d>: %d workgroups per CU at %zu B of LDS\n", SX, SY, nb, lds); \
      }                                                                                                     \
    }                                                                                                       \
    hipLaunchKernelGGL((k_jac_lattice<F, LIN, SX, SY>), dim3(grid), dim3(2 * SX * SY), lds, s, a, m.vx.p, u, x, \
                       d.sid8.p, mask, d.len.p, d.pack.p, L.dict_vals.p, gadd, y,                           \
                       (const double*)(g_jac_uniform_geo && cl.geo_uniform ? cl.ugeo.p : nullptr));         \
  } while (0)
#define NSFEM_JL(F, LIN)                                                                                    \
  do {                                                                                                      \
    switch (g_jac_lattice_tile) {                                                                           \
      case 1: NSFEM_JL_T(F, LIN, 16, 8); break;                                                             \
      default: NSFEM_JL_T(F, LIN, 32, 8); break;                                                            \
    }                                                                                                       \
  } while (0)
  if (lds > 96 * 1024) return false;
#define NSFEM_JL_F(LIN)                                                                                     \
  switch (form) {                                                                                           \
    case 0: NSFEM_JL(0, LIN); break;                                                                        \
    case 1: NSFEM_JL(1, LIN); break;                                                                        \
    case 2: NSFEM_JL(2, LIN); break;                                                                        \
    case 3: NSFEM_JL(3, LIN); break;                                                                        \
    default: throw Error(NSFEM_ERR_ARG, "unknown convective form");                                         \
  }
  if (lin == 0) { NSFEM_JL_F(0) } else if (lin == 2) { NSFEM_JL_F(2) } else { NSFEM_JL_F(1) }
#undef NSFEM_JL_F
#undef NSFEM_JL_T
#undef NSFEM_JL
  NSFEM_HIP(hipGetLastError());
  return true;
}

bool launch_jacobian_lattice(hipStream_t s, const MeshDev& m, const BlockMat& L, const double* u,
                             const double* x, double cc, int form, bool picard, const uint8_t* mask,
                             double* y, int phase, int gh_lo, int gh_hi) {
  if (!mask) return false;
  return launch_lattice_cells(s, m, L, u, x, cc, form, picard ? 2 : 1, mask, nullptr, y, phase, gh_lo, gh_hi);
}
// y = L u + g + c_c conv(u): the momentum residual before its Dirichlet rows are set.  Only on dictionaries that
// equal the assembled matrix bit for bit (the residual decides the convergence of the Newton iteration)
bool launch_residual_lattice(hipStream_t s, const MeshDev& m, const BlockMat& L, const double* u, const double* g,
                             double cc, int form, double* y) {
  if (!L.dict || !L.dict->exact || !g) return false;
  return launch_lattice_cells(s, m, L, u, u, cc, form, 0, nullptr, g, y);
}

}  // namespace nsfem
