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
//   * scatter into CSR values through the precomputed slot map with hardware
//     fp64 atomics (global_atomic_add_f64, no CAS loop).
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
  gx = g.ji00 * dr0 + g.ji10 * dr1;
  gy = g.ji01 * dr0 + g.ji11 * dr1;
}

__device__ __forceinline__ void atomic_add(double* p, double v) { unsafeAtomicAdd(p, v); }

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
    const int s = slot[(size_t)(i * 6 + j) * nc + c];
    atomic_add(mass + s, m[j]);
    atomic_add(stiff + s, k[j]);
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
    const int s = slot[(size_t)(i * 3 + j) * nc + c];
    atomic_add(stiff + s, 0.5 * g.adet * (gix * gjx + giy * gjy));
    atomic_add(mass + s, m * g.adet);
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
    const int s = slot12[(size_t)(i * 6 + j) * nc + c];
    atomic_add(div + (size_t)2 * s, dx[j]);
    atomic_add(div + (size_t)2 * s + 1, dy[j]);
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
    const int s = slot21[(size_t)(i * 3 + j) * nc + c];
    atomic_add(grad + (size_t)2 * s, gr[j][0]);
    atomic_add(grad + (size_t)2 * s + 1, gr[j][1]);
    atomic_add(divT + (size_t)2 * s, dt[j][0]);
    atomic_add(divT + (size_t)2 * s + 1, dt[j][1]);
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
    const int s = slot[(size_t)(i * 6 + j) * nc + c];
#pragma unroll
    for (int e = 0; e < 4; ++e) atomic_add(E + (size_t)4 * s + e, acc[j][e]);
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

// ---- Newton convection block:  cc * [ phi_i (u.grad phi_j) delta_ab + phi_i phi_j d_b u_a ]
// (Gateaux derivative of dot(dot(grad(u), u), v), source/ns_solver_base.py:378)
__global__ __launch_bounds__(256) void k_conv_jac(int nc, const double* __restrict__ vx,
                                                  const int32_t* __restrict__ p2,
                                                  const int32_t* __restrict__ slot,
                                                  const double* __restrict__ u, double cc,
                                                  double* __restrict__ J) {
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
    }
    const double wpi = c_q.w[q] * g.adet * c_q.phi2[q][i] * cc;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const double udg = uqx * gx[j] + uqy * gy[j];
      const double pj = c_q.phi2[q][j];
      acc[j][0] += wpi * (udg + pj * g00);
      acc[j][1] += wpi * (pj * g01);
      acc[j][2] += wpi * (pj * g10);
      acc[j][3] += wpi * (udg + pj * g11);
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int s = slot[(size_t)(i * 6 + j) * nc + c];
#pragma unroll
    for (int e = 0; e < 4; ++e) atomic_add(J + (size_t)4 * s + e, acc[j][e]);
  }
}

// ---- convection residual:  b_(i,a) += cc * int ((grad u) u)_a phi_i
__global__ __launch_bounds__(256) void k_conv_res(int nc, const double* __restrict__ vx,
                                                  const int32_t* __restrict__ p2,
                                                  const double* __restrict__ u, double cc,
                                                  double* __restrict__ b) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nc * 6) return;
  const int i = (int)(t / nc), c = (int)(t % nc);
  const CellGeo g = load_geo(vx, nc, c);
  double ux[6], uy[6];
  int node_i = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int node = p2[(size_t)k * nc + c];
    if (k == i) node_i = node;
    const double2 v = reinterpret_cast<const double2*>(u)[node];
    ux[k] = v.x;
    uy[k] = v.y;
  }
  double rx = 0.0, ry = 0.0;
  for (int q = 0; q < 7; ++q) {
    double uqx = 0.0, uqy = 0.0, g00 = 0.0, g01 = 0.0, g10 = 0.0, g11 = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      double gx, gy;
      phys(g, c_q.dphi2[q][k][0], c_q.dphi2[q][k][1], gx, gy);
      const double ph = c_q.phi2[q][k];
      uqx += ph * ux[k];
      uqy += ph * uy[k];
      g00 += gx * ux[k];
      g01 += gy * ux[k];
      g10 += gx * uy[k];
      g11 += gy * uy[k];
    }
    const double wpi = c_q.w[q] * g.adet * c_q.phi2[q][i] * cc;
    rx += wpi * (g00 * uqx + g01 * uqy);
    ry += wpi * (g10 * uqx + g11 * uqy);
  }
  atomic_add(b + (size_t)2 * node_i, rx);
  atomic_add(b + (size_t)2 * node_i + 1, ry);
}

// ------------------------------------------------------------- launch wrappers
static inline int grid_for(int64_t n) { return (int)((n + kBlock - 1) / kBlock); }

void launch_assemble_p2_scalar(hipStream_t s, const MeshDev& m, const Pattern& p22, double* mass,
                               double* stiff) {
  hipLaunchKernelGGL(k_p2_scalar, dim3(grid_for((int64_t)m.n_cells * 6)), dim3(kBlock), 0, s,
                     m.n_cells, m.vx.p, p22.slot.p, mass, stiff);
  NSFEM_HIP(hipGetLastError());
}
void launch_assemble_p1_scalar(hipStream_t s, const MeshDev& m, const Pattern& p11, double* stiff,
                               double* mass) {
  hipLaunchKernelGGL(k_p1_scalar, dim3(grid_for((int64_t)m.n_cells * 3)), dim3(kBlock), 0, s,
                     m.n_cells, m.vx.p, p11.slot.p, stiff, mass);
  NSFEM_HIP(hipGetLastError());
}
void launch_assemble_div_grad(hipStream_t s, const MeshDev& m, const Pattern& p12,
                              const Pattern& p21, double* div, double* grad, double* divT) {
  hipLaunchKernelGGL(k_div, dim3(grid_for((int64_t)m.n_cells * 3)), dim3(kBlock), 0, s, m.n_cells,
                     m.vx.p, p12.slot.p, div);
  hipLaunchKernelGGL(k_grad, dim3(grid_for((int64_t)m.n_cells * 6)), dim3(kBlock), 0, s, m.n_cells,
                     m.vx.p, p21.slot.p, grad, divT);
  NSFEM_HIP(hipGetLastError());
}
void launch_assemble_viscous_extra(hipStream_t s, const MeshDev& m, const Pattern& p22,
                                   double* extra) {
  hipLaunchKernelGGL(k_visc_extra, dim3(grid_for((int64_t)m.n_cells * 6)), dim3(kBlock), 0, s,
                     m.n_cells, m.vx.p, p22.slot.p, extra);
  NSFEM_HIP(hipGetLastError());
}
void launch_jacobian_init(hipStream_t s, int nnz, const double* L, const double* E, double cvE,
                          double* J) {
  int grid = grid_for(nnz);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_jac_init, dim3(grid), dim3(kBlock), 0, s, nnz, L, E, cvE, J);
  NSFEM_HIP(hipGetLastError());
}
void launch_convection_jacobian(hipStream_t s, const MeshDev& m, const Pattern& p22,
                                const double* u, double cc, double* J) {
  hipLaunchKernelGGL(k_conv_jac, dim3(grid_for((int64_t)m.n_cells * 6)), dim3(kBlock), 0, s,
                     m.n_cells, m.vx.p, m.p2.p, p22.slot.p, u, cc, J);
  NSFEM_HIP(hipGetLastError());
}
void launch_convection_residual(hipStream_t s, const MeshDev& m, const double* u, double cc,
                                double* b) {
  hipLaunchKernelGGL(k_conv_res, dim3(grid_for((int64_t)m.n_cells * 6)), dim3(kBlock), 0, s,
                     m.n_cells, m.vx.p, m.p2.p, u, cc, b);
  NSFEM_HIP(hipGetLastError());
}

}  // namespace nsfem
