// Element kernels for tetrahedral Taylor-Hood meshes (P2: 4 vertex + 6 edge nodes in UFC order,
// P1: 4 vertices) -- the 3D counterpart of assembly.hip with the same conventions:
//   * one thread per (cell, local row i); SoA per-cell geometry / dof maps (coalesced);
//   * reference tables in __constant__ memory (uniform quadrature index => scalar loads);
//   * element tensors are STORED to an element buffer [cell][i][j][block] and summed per CSR
//     slot by the gather kernels of assembly.hip in a fixed order (no atomics, reproducible).
// Forms: source/ns_solver_base.py:370-399,478-499 (the reference writes them dimension-
// independently; its 3D branches are never exercised).  The 15-point Keast rule
// is exact for degree 5, the highest degree any of these integrands reaches on affine cells.
#include "nsfem_internal.hpp"

namespace nsfem {

__constant__ QuadTables3 c_q3;

void fill_quad_tables_3d(QuadTables3& t) {
  const double s15 = std::sqrt(15.0);
  const double a1 = (7.0 - s15) / 34.0, a2 = (7.0 + s15) / 34.0, b = (10.0 - 2.0 * s15) / 40.0;
  const double w0 = 16.0 / 135.0, w1 = (2665.0 + 14.0 * s15) / 37800.0,
               w2 = (2665.0 - 14.0 * s15) / 37800.0, w3 = 10.0 / 189.0;
  double pts[15][3];
  int n = 0;
  pts[n][0] = pts[n][1] = pts[n][2] = 0.25; t.w[n++] = w0 / 6.0;
  const double as[2] = {a1, a2}, ws[2] = {w1, w2};
  for (int k = 0; k < 2; ++k) {
    const double a = as[k], c = 1.0 - 3.0 * a;
    const double p4[4][3] = {{a, a, a}, {c, a, a}, {a, c, a}, {a, a, c}};
    for (int m = 0; m < 4; ++m) {
      for (int d = 0; d < 3; ++d) pts[n][d] = p4[m][d];
      t.w[n++] = ws[k] / 6.0;
    }
  }
  const double c = 0.5 - b;
  const double p6[6][3] = {{b, b, c}, {b, c, b}, {c, b, b}, {b, c, c}, {c, b, c}, {c, c, b}};
  for (int m = 0; m < 6; ++m) {
    for (int d = 0; d < 3; ++d) pts[n][d] = p6[m][d];
    t.w[n++] = w3 / 6.0;
  }
  const double dl[4][3] = {{-1.0, -1.0, -1.0}, {1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}};
  const int pr[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
  for (int q = 0; q < 15; ++q) {
    const double l[4] = {1.0 - pts[q][0] - pts[q][1] - pts[q][2], pts[q][0], pts[q][1], pts[q][2]};
    for (int i = 0; i < 4; ++i) {
      t.phi1[q][i] = l[i];
      t.phi2[q][i] = l[i] * (2.0 * l[i] - 1.0);
      for (int d = 0; d < 3; ++d) t.dphi2[q][i][d] = (4.0 * l[i] - 1.0) * dl[i][d];
    }
    for (int e = 0; e < 6; ++e) {
      const int a = pr[e][0], bb = pr[e][1];
      t.phi2[q][4 + e] = 4.0 * l[a] * l[bb];
      for (int d = 0; d < 3; ++d) t.dphi2[q][4 + e][d] = 4.0 * (l[a] * dl[bb][d] + l[bb] * dl[a][d]);
    }
  }
}

// CFL diagnostic tables: P2 basis at the 14 points of the degree-4 Keast rule (FFC's "default"
// scheme for quadrature degree 4 on tetrahedra, which the reference's DG2 projection requests:
// source/ns_problem.py:570) and the projection matrix  M^{-1} Phi^T W  (10 x 14).
struct CflTables3 {
  double phi[14][10];
  double inv[10][14];
};
__constant__ CflTables3 c_cfl3;

static void fill_cfl_tables_3d(CflTables3& t) {
  double l[14][4], w[14];
  int n = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = i + 1; j < 4; ++j) {
      for (int k = 0; k < 4; ++k) l[n][k] = (k == i || k == j) ? 0.5 : 0.0;
      w[n++] = 0.0031746031746032;
    }
  const double pa[2] = {0.1005267652252045, 0.3143728734931922};
  const double qa[2] = {0.6984197043243866, 0.0568813795204234};
  const double wa[2] = {0.0147649707904968, 0.0221397911142651};
  for (int r = 0; r < 2; ++r)
    for (int k = 0; k < 4; ++k) {
      for (int m = 0; m < 4; ++m) l[n][m] = (m == k) ? qa[r] : pa[r];
      w[n++] = wa[r];
    }
  const int pr[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
  for (int q = 0; q < 14; ++q) {
    for (int i = 0; i < 4; ++i) t.phi[q][i] = l[q][i] * (2.0 * l[q][i] - 1.0);
    for (int e = 0; e < 6; ++e) t.phi[q][4 + e] = 4.0 * l[q][pr[e][0]] * l[q][pr[e][1]];
  }
  double m[10][20];
  for (int i = 0; i < 10; ++i)
    for (int j = 0; j < 10; ++j) {
      double a = 0.0;
      for (int q = 0; q < 14; ++q) a += w[q] * t.phi[q][i] * t.phi[q][j];
      m[i][j] = a;
      m[i][10 + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < 10; ++c) {                      // Gauss-Jordan with partial pivoting
    int piv = c;
    for (int r = c + 1; r < 10; ++r)
      if (std::fabs(m[r][c]) > std::fabs(m[piv][c])) piv = r;
    for (int k = 0; k < 20; ++k) std::swap(m[c][k], m[piv][k]);
    const double d = 1.0 / m[c][c];
    for (int k = 0; k < 20; ++k) m[c][k] *= d;
    for (int r = 0; r < 10; ++r)
      if (r != c) {
        const double f = m[r][c];
        for (int k = 0; k < 20; ++k) m[r][k] -= f * m[c][k];
      }
  }
  for (int i = 0; i < 10; ++i)
    for (int q = 0; q < 14; ++q) {
      double a = 0.0;
      for (int j = 0; j < 10; ++j) a += m[i][10 + j] * t.phi[q][j];
      t.inv[i][q] = a * w[q];
    }
}

void upload_quad_tables_3d() {
  QuadTables3 t;
  fill_quad_tables_3d(t);
  NSFEM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_q3), &t, sizeof(QuadTables3)));
  CflTables3 cf;
  fill_cfl_tables_3d(cf);
  NSFEM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_cfl3), &cf, sizeof(CflTables3)));
}

struct CellGeo3 {
  double ji[3][3];   // J^{-1}[b][a] = d xi_b / d x_a
  double adet;
};

__device__ __forceinline__ CellGeo3 load_geo3(const double* __restrict__ vx, int nc, int c) {
  double x[4][3];
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int d = 0; d < 3; ++d) x[v][d] = vx[(size_t)(3 * v + d) * nc + c];
  double J[3][3];   // J[a][b] = x_{b+1}[a] - x_0[a]
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) J[a][b] = x[b + 1][a] - x[0][a];
  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
  const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
  const double id = 1.0 / det;
  CellGeo3 g;
  g.ji[0][0] = c00 * id;
  g.ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
  g.ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
  g.ji[1][0] = c01 * id;
  g.ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
  g.ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
  g.ji[2][0] = c02 * id;
  g.ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
  g.ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
  g.adet = fabs(det);
  return g;
}

// physical gradient of a reference gradient dr: out_a = sum_b Jinv[b][a] dr_b
__device__ __forceinline__ void phys3(const CellGeo3& g, const double* dr, double* out) {
#pragma unroll
  for (int a = 0; a < 3; ++a) out[a] = g.ji[0][a] * dr[0] + g.ji[1][a] * dr[1] + g.ji[2][a] * dr[2];
}

// ------------------------------------------------------------------ scalar P2
__global__ __launch_bounds__(256) void k3_p2_scalar(int nc, const double* __restrict__ vx,
                                                    double* __restrict__ mass,
                                                    double* __restrict__ stiff) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nc * 10) return;
  const int i = (int)(t / nc), c = (int)(t % nc);
  const CellGeo3 g = load_geo3(vx, nc, c);
  double m[10], k[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) m[j] = k[j] = 0.0;
  for (int q = 0; q < 15; ++q) {
    const double w = c_q3.w[q] * g.adet;
    double gi[3];
    phys3(g, c_q3.dphi2[q][i], gi);
    const double pi = c_q3.phi2[q][i] * w;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      double gj[3];
      phys3(g, c_q3.dphi2[q][j], gj);
      m[j] += pi * c_q3.phi2[q][j];
      k[j] += w * (gi[0] * gj[0] + gi[1] * gj[1] + gi[2] * gj[2]);
    }
  }
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    const size_t s = ((size_t)c * 10 + i) * 10 + j;
    mass[s] = m[j];
    stiff[s] = k[j];
  }
}

// ------------------------------------------------------------------ scalar P1
__global__ __launch_bounds__(256) void k3_p1_scalar(int nc, const double* __restrict__ vx,
                                                    double* __restrict__ stiff,
                                                    double* __restrict__ mass) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nc * 4) return;
  const int i = (int)(t / nc), c = (int)(t % nc);
  const CellGeo3 g = load_geo3(vx, nc, c);
  const double dl[4][3] = {{-1.0, -1.0, -1.0}, {1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}};
  double gi[3];
  phys3(g, dl[i], gi);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double gj[3];
    phys3(g, dl[j], gj);
    const size_t s = ((size_t)c * 4 + i) * 4 + j;
    stiff[s] = g.adet / 6.0 * (gi[0] * gj[0] + gi[1] * gj[1] + gi[2] * gj[2]);
    mass[s] = g.adet / 120.0 * (i == j ? 2.0 : 1.0);       // V/20 (1 + delta_ij), V = adet/6
  }
}

// --------------------------------------------------- divergence (P1 rows, 1x3)
__global__ __launch_bounds__(256) void k3_div(int nc, const double* __restrict__ vx,
                                              double* __restrict__ div) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nc * 4) return;
  const int i = (int)(t / nc), c = (int)(t % nc);
  const CellGeo3 g = load_geo3(vx, nc, c);
  double d[10][3];
#pragma unroll
  for (int j = 0; j < 10; ++j) d[j][0] = d[j][1] = d[j][2] = 0.0;
  for (int q = 0; q < 15; ++q) {
    const double wp = c_q3.w[q] * g.adet * c_q3.phi1[q][i];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      double gj[3];
      phys3(g, c_q3.dphi2[q][j], gj);
      d[j][0] += wp * gj[0];
      d[j][1] += wp * gj[1];
      d[j][2] += wp * gj[2];
    }
  }
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    const size_t s = ((size_t)c * 4 + i) * 10 + j;
    div[3 * s] = d[j][0];
    div[3 * s + 1] = d[j][1];
    div[3 * s + 2] = d[j][2];
  }
}

// --------------------------- gradient / transposed divergence (P2 rows, 3x1)
__global__ __launch_bounds__(256) void k3_grad(int nc, const double* __restrict__ vx,
                                               double* __restrict__ grad,
                                               double* __restrict__ divT) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nc * 10) return;
  const int i = (int)(t / nc), c = (int)(t % nc);
  const CellGeo3 g = load_geo3(vx, nc, c);
  const double dl[4][3] = {{-1.0, -1.0, -1.0}, {1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}};
  double gr[4][3], dt[4][3], gp[4][3];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    phys3(g, dl[j], gp[j]);
    gr[j][0] = gr[j][1] = gr[j][2] = dt[j][0] = dt[j][1] = dt[j][2] = 0.0;
  }
  for (int q = 0; q < 15; ++q) {
    const double w = c_q3.w[q] * g.adet;
    const double pi = c_q3.phi2[q][i] * w;
    double gi[3];
    phys3(g, c_q3.dphi2[q][i], gi);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        gr[j][a] += pi * gp[j][a];                          // int phi_i d_a psi_j
        dt[j][a] += w * gi[a] * c_q3.phi1[q][j];            // int d_a phi_i psi_j
      }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const size_t s = ((size_t)c * 10 + i) * 4 + j;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      grad[3 * s + a] = gr[j][a];
      divT[3 * s + a] = dt[j][a];
    }
  }
}

// ---- convection blocks (Newton / Picard), forms 0 standard, 2 divergence, 3 skew-symmetric;
// with u, G_ab = d_b u_a at the quadrature point, test (phi_i, a), trial (phi_j, b):
//   standard    phi_i [ (u.g_j) d_ab + phi_j G_ab ]                 (Picard: first term)
//   divergence  standard + 1/2 phi_i [ g_j,b u_a + div phi_j d_ab ] (Picard: 1st + 4th term)
//   skew        1/2 standard - 1/2 [ g_i,b phi_j u_a + (u.g_i) phi_j d_ab ]
template <int FORM, bool PICARD>
__global__ __launch_bounds__(256) void k3_conv_jac(int nc, const double* __restrict__ vx,
                                                   const int32_t* __restrict__ p2,
                                                   const double* __restrict__ u, double cc,
                                                   double* __restrict__ ebuf) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nc * 10) return;
  const int i = (int)(t / nc), c = (int)(t % nc);
  const CellGeo3 g = load_geo3(vx, nc, c);
  int node[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) node[k] = p2[(size_t)k * nc + c];
  double acc[10][9];
#pragma unroll
  for (int j = 0; j < 10; ++j)
#pragma unroll
    for (int e = 0; e < 9; ++e) acc[j][e] = 0.0;
  for (int q = 0; q < 15; ++q) {
    double gk[10][3];
    double uq[3] = {0.0, 0.0, 0.0};
    double G[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      phys3(g, c_q3.dphi2[q][k], gk[k]);
      const double ph = c_q3.phi2[q][k];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const double ua = u[(size_t)node[k] * 3 + a];      // L1/L2-resident re-read per q
        uq[a] += ph * ua;
#pragma unroll
        for (int b = 0; b < 3; ++b) G[a][b] += gk[k][b] * ua;
      }
    }
    const double w = c_q3.w[q] * g.adet * cc;
    const double wpi = w * c_q3.phi2[q][i];
    const double div = G[0][0] + G[1][1] + G[2][2];
    const double udgi = uq[0] * gk[i][0] + uq[1] * gk[i][1] + uq[2] * gk[i][2];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const double udg = uq[0] * gk[j][0] + uq[1] * gk[j][1] + uq[2] * gk[j][2];
      const double pj = c_q3.phi2[q][j];
      if (FORM == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          acc[j][a * 3 + a] += wpi * udg;
          if (!PICARD)
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[j][a * 3 + b] += wpi * pj * G[a][b];
        }
      } else if (FORM == 1) {
        // rotational: d/dv_(j,b) of [curl(u) x v + curl(v) x u]_a
        //   = phi_j eps_(a c b) curl(u)_c  +  (u . g_j) d_ab - u_b g_j,a      (Picard: first term)
        const double wu[3] = {G[2][1] - G[1][2], G[0][2] - G[2][0], G[1][0] - G[0][1]};
        const double cx[3][3] = {{0.0, -wu[2], wu[1]}, {wu[2], 0.0, -wu[0]}, {-wu[1], wu[0], 0.0}};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            double t = pj * cx[a][b];
            if (!PICARD) t += (a == b ? udg : 0.0) - uq[b] * gk[j][a];
            acc[j][a * 3 + b] += wpi * t;
          }
      } else if (FORM == 2) {
        const double hd = 0.5 * div * pj;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          acc[j][a * 3 + a] += wpi * (udg + hd);
          if (!PICARD)
#pragma unroll
            for (int b = 0; b < 3; ++b)
              acc[j][a * 3 + b] += wpi * (pj * G[a][b] + 0.5 * gk[j][b] * uq[a]);
        }
      } else {
        const double sk = 0.5 * (wpi * udg - w * udgi * pj);
        const double hw = 0.5 * w * pj;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          acc[j][a * 3 + a] += sk;
          if (!PICARD)
#pragma unroll
            for (int b = 0; b < 3; ++b)
              acc[j][a * 3 + b] += 0.5 * wpi * pj * G[a][b] - hw * gk[i][b] * uq[a];
        }
      }
    }
  }
  double* out = ebuf + ((size_t)c * 100 + (size_t)i * 10) * 9;
#pragma unroll
  for (int j = 0; j < 10; ++j)
#pragma unroll
    for (int e = 0; e < 9; ++e) out[j * 9 + e] = acc[j][e];
}

// ---- convection residual and its linearised action, one thread per CELL: u (and the direction
// v) at the 10 nodes live in registers, u_q / grad u_q are formed once per quadrature point and
// feed all 10 test functions (the per-(cell, i) mapping recomputed them 10 times).
//   LIN = 0   r_(i,a) = int c(u)_a phi_i                          (residual)
//   LIN = 1   r_(i,a) = int [d c(u)/du . v]_a phi_i               (Newton matrix times v)
//   LIN = 2   Picard linearisation times v (source/ns_solver_base.py:478-499)
// forms: 0 standard (grad u) u; 1 rotational curl(u) x u; 2 divergence + 1/2 div(u) u; 3 skew-symmetric
//   1/2 [ (grad u) u . phi - ((grad phi) u) . u ]
template <int FORM, int LIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k3_conv_cell(int nc, const double* __restrict__ vx,
                                                    const int32_t* __restrict__ p2,
                                                    const double* __restrict__ u,
                                                    const double* __restrict__ v, double cc,
                                                    const int32_t* __restrict__ ndst,
                                                    double* __restrict__ rbuf) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const CellGeo3 g = load_geo3(vx, nc, c);
  double un[10][3], vn[LIN ? 10 : 1][3];
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const size_t node = (size_t)p2[(size_t)k * nc + c];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      un[k][a] = u[node * 3 + a];
      if (LIN) vn[k][a] = v[node * 3 + a];
    }
  }
  double r[10][3];
#pragma unroll
  for (int i = 0; i < 10; ++i) r[i][0] = r[i][1] = r[i][2] = 0.0;
  for (int q = 0; q < 15; ++q) {
    double gk[10][3];
    double uq[3] = {0.0, 0.0, 0.0}, vq[3] = {0.0, 0.0, 0.0};
    double Gu[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
    double Gv[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      phys3(g, c_q3.dphi2[q][k], gk[k]);
      const double ph = c_q3.phi2[q][k];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        uq[a] += ph * un[k][a];
        if (LIN) vq[a] += ph * vn[k][a];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          Gu[a][b] += gk[k][b] * un[k][a];
          if (LIN) Gv[a][b] += gk[k][b] * vn[k][a];
        }
      }
    }
    const double w = c_q3.w[q] * g.adet * cc;
    // f_a: coefficient of phi_i ; h_a, hs: the skew form's  - (s . grad phi_i) h_a  terms
    double f[3];
    const double divu = Gu[0][0] + Gu[1][1] + Gu[2][2];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double adv_u = Gu[a][0] * uq[0] + Gu[a][1] * uq[1] + Gu[a][2] * uq[2];     // (grad u) u
      if (FORM == 1) {
        f[a] = 0.0;              // rotational form: set after the loop (needs all components)
      } else if (LIN == 0) {
        f[a] = adv_u + (FORM == 2 ? 0.5 * divu * uq[a] : 0.0);
      } else {
        const double gv_u = Gv[a][0] * uq[0] + Gv[a][1] * uq[1] + Gv[a][2] * uq[2];    // (grad v) u
        const double gu_v = Gu[a][0] * vq[0] + Gu[a][1] * vq[1] + Gu[a][2] * vq[2];    // (grad u) v
        f[a] = gv_u + (LIN == 1 ? gu_v : 0.0);
        if (FORM == 2) {
          f[a] += 0.5 * divu * vq[a];
          if (LIN == 1) f[a] += 0.5 * (Gv[0][0] + Gv[1][1] + Gv[2][2]) * uq[a];
        }
      }
      if (FORM == 3) f[a] *= 0.5;
    }
    if (FORM == 1) {
      // curl(u) x u ; Newton: curl(v) x u + curl(u) x v ; Picard: curl(u) x v
      // (source/ns_solver_base.py:384, :492)
      const double wu[3] = {Gu[2][1] - Gu[1][2], Gu[0][2] - Gu[2][0], Gu[1][0] - Gu[0][1]};
      const double* z = LIN ? vq : uq;                       // curl(u) x z
      f[0] = wu[1] * z[2] - wu[2] * z[1];
      f[1] = wu[2] * z[0] - wu[0] * z[2];
      f[2] = wu[0] * z[1] - wu[1] * z[0];
      if (LIN == 1) {
        const double wv[3] = {Gv[2][1] - Gv[1][2], Gv[0][2] - Gv[2][0], Gv[1][0] - Gv[0][1]};
        f[0] += wv[1] * uq[2] - wv[2] * uq[1];
        f[1] += wv[2] * uq[0] - wv[0] * uq[2];
        f[2] += wv[0] * uq[1] - wv[1] * uq[0];
      }
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const double wpi = w * c_q3.phi2[q][i];
#pragma unroll
      for (int a = 0; a < 3; ++a) r[i][a] += wpi * f[a];
      if (FORM == 3) {
        const double ugi = uq[0] * gk[i][0] + uq[1] * gk[i][1] + uq[2] * gk[i][2];
        if (LIN == 0) {
#pragma unroll
          for (int a = 0; a < 3; ++a) r[i][a] -= 0.5 * w * ugi * uq[a];
        } else {
          const double vgi = vq[0] * gk[i][0] + vq[1] * gk[i][1] + vq[2] * gk[i][2];
#pragma unroll
          for (int a = 0; a < 3; ++a)
            r[i][a] -= 0.5 * w * (ugi * vq[a] + (LIN == 1 ? vgi * uq[a] : 0.0));
        }
      }
    }
  }
  // node-sorted element buffer: entry (c, i) lands inside the contiguous run of its node
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    double* out = rbuf + (size_t)ndst[(size_t)i * nc + c] * 3;
    out[0] = r[i][0];
    out[1] = r[i][1];
    out[2] = r[i][2];
  }
}

// J[s] = L[s] I_3 + sum of the element blocks scattered to slot s (ascending source order)
__global__ __launch_bounds__(256) void k3_jac_gather(int nnz, const int32_t* __restrict__ cptr,
                                                     const int32_t* __restrict__ cidx,
                                                     const double* __restrict__ ebuf,
                                                     const double* __restrict__ L,
                                                     const double* __restrict__ E, double cvE,
                                                     double* __restrict__ J) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nnz * 9) return;
  const int s = (int)(t / 9), e = (int)(t % 9);
  double acc = (e == 0 || e == 4 || e == 8) ? L[s] : 0.0;
  if (E) acc += cvE * E[t];                      // traction-form block (3x3, same slot layout)
  if (ebuf)
    for (int k = cptr[s]; k < cptr[s + 1]; ++k) acc += ebuf[(size_t)cidx[k] * 9 + e];
  J[t] = acc;
}

__global__ __launch_bounds__(256) void k3_res_gather(int n_nodes, const int32_t* __restrict__ nptr,
                                                     const double* __restrict__ rbuf,
                                                     const uint8_t* __restrict__ skip,
                                                     double* __restrict__ b) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)n_nodes * 3) return;
  if (skip && skip[t] != 0) return;     // flagged entries (Dirichlet / ghost rows) stay untouched
  const int n = (int)(t / 3), a = (int)(t % 3);
  double acc = b[t];
  int k = nptr[n];
  const int e = nptr[n + 1];
  // the node's contributions are the contiguous run nptr[n] .. nptr[n+1] of the node-sorted
  // buffer; four independent loads in flight per lane, summed in ascending (cell) order
  for (; k + 4 <= e; k += 4) {
    const double v0 = rbuf[(size_t)k * 3 + a], v1 = rbuf[(size_t)(k + 1) * 3 + a];
    const double v2 = rbuf[(size_t)(k + 2) * 3 + a], v3 = rbuf[(size_t)(k + 3) * 3 + a];
    acc += v0;
    acc += v1;
    acc += v2;
    acc += v3;
  }
  for (; k < e; ++k) acc += rbuf[(size_t)k * 3 + a];
  b[t] = acc;
}

// ------------------------------------------------------------- launch wrappers
static inline int grid3(int64_t n) { return (int)((n + kBlock - 1) / kBlock); }
void gather_vals(hipStream_t s, const Pattern& p, int bs, const double* ebuf, double* vals);

void assemble_p2_scalar_3d(hipStream_t s, const MeshDev& m, const Pattern& p22, double* mass,
                           double* stiff) {
  DevBuf<double> tmp;
  tmp.alloc((size_t)m.n_cells * 200);
  double* t0 = tmp.p;
  double* t1 = tmp.p + (size_t)m.n_cells * 100;
  hipLaunchKernelGGL(k3_p2_scalar, dim3(grid3((int64_t)m.n_cells * 10)), dim3(kBlock), 0, s,
                     m.n_cells, m.vx.p, t0, t1);
  gather_vals(s, p22, 1, t0, mass);
  gather_vals(s, p22, 1, t1, stiff);
  NSFEM_HIP(hipStreamSynchronize(s));
}
void assemble_p1_scalar_3d(hipStream_t s, const MeshDev& m, const Pattern& p11, double* stiff,
                           double* mass) {
  DevBuf<double> tmp;
  tmp.alloc((size_t)m.n_cells * 32);
  double* t0 = tmp.p;
  double* t1 = tmp.p + (size_t)m.n_cells * 16;
  hipLaunchKernelGGL(k3_p1_scalar, dim3(grid3((int64_t)m.n_cells * 4)), dim3(kBlock), 0, s,
                     m.n_cells, m.vx.p, t0, t1);
  gather_vals(s, p11, 1, t0, stiff);
  gather_vals(s, p11, 1, t1, mass);
  NSFEM_HIP(hipStreamSynchronize(s));
}
void assemble_div_grad_3d(hipStream_t s, const MeshDev& m, const Pattern& p12, const Pattern& p21,
                          double* div, double* grad, double* divT) {
  DevBuf<double> tmp;
  tmp.alloc((size_t)m.n_cells * 120 * 3);
  double* t0 = tmp.p;
  double* t1 = tmp.p + (size_t)m.n_cells * 120;
  double* t2 = tmp.p + (size_t)m.n_cells * 240;
  hipLaunchKernelGGL(k3_div, dim3(grid3((int64_t)m.n_cells * 4)), dim3(kBlock), 0, s, m.n_cells,
                     m.vx.p, t0);
  hipLaunchKernelGGL(k3_grad, dim3(grid3((int64_t)m.n_cells * 10)), dim3(kBlock), 0, s, m.n_cells,
                     m.vx.p, t1, t2);
  gather_vals(s, p12, 3, t0, div);
  gather_vals(s, p21, 3, t1, grad);
  gather_vals(s, p21, 3, t2, divT);
  NSFEM_HIP(hipStreamSynchronize(s));
}
void jacobian_init_3d(hipStream_t s, int nnz, const double* L, const double* E, double cvE,
                      double* J) {
  hipLaunchKernelGGL(k3_jac_gather, dim3(grid3((int64_t)nnz * 9)), dim3(kBlock), 0, s, nnz, nullptr,
                     nullptr, nullptr, L, E, cvE, J);
  NSFEM_HIP(hipGetLastError());
}

// traction-form extra block  E[(i,a),(j,b)] = int d_b phi_i d_a phi_j  (the grad(u)^T part of
// inner(grad u + grad u^T, sym grad v), source/ns_solver_base.py:669-671), one thread per (cell, i)
__global__ __launch_bounds__(256) void k3_visc_extra(int nc, const double* __restrict__ vx,
                                                     double* __restrict__ E) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)nc * 10) return;
  const int i = (int)(t / nc), c = (int)(t % nc);
  const CellGeo3 g = load_geo3(vx, nc, c);
  double acc[10][9];
#pragma unroll
  for (int j = 0; j < 10; ++j)
#pragma unroll
    for (int e = 0; e < 9; ++e) acc[j][e] = 0.0;
  for (int q = 0; q < 15; ++q) {
    const double w = c_q3.w[q] * g.adet;
    double gi[3];
    phys3(g, c_q3.dphi2[q][i], gi);
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      double gj[3];
      phys3(g, c_q3.dphi2[q][j], gj);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[j][a * 3 + b] += w * gi[b] * gj[a];
    }
  }
  double* out = E + ((size_t)c * 100 + (size_t)i * 10) * 9;
#pragma unroll
  for (int j = 0; j < 10; ++j)
#pragma unroll
    for (int e = 0; e < 9; ++e) out[j * 9 + e] = acc[j][e];
}

void assemble_viscous_extra_3d(hipStream_t s, const MeshDev& m, const Pattern& p22, double* extra) {
  DevBuf<double> tmp;
  tmp.alloc((size_t)m.n_cells * 900);
  hipLaunchKernelGGL(k3_visc_extra, dim3(grid3((int64_t)m.n_cells * 10)), dim3(kBlock), 0, s,
                     m.n_cells, m.vx.p, tmp.p);
  NSFEM_HIP(hipGetLastError());
  gather_vals(s, p22, 9, tmp.p, extra);
  NSFEM_HIP(hipStreamSynchronize(s));
}

void convection_jacobian_3d(hipStream_t s, const MeshDev& m, const Pattern& p22, const double* u,
                            double cc, const double* L, const double* E, double cvE, double* J,
                            int form, bool picard) {
  const dim3 grid(grid3((int64_t)m.n_cells * 10)), block(kBlock);
#define NSFEM_CJ3(F, P) \
  hipLaunchKernelGGL((k3_conv_jac<F, P>), grid, block, 0, s, m.n_cells, m.vx.p, m.p2.p, u, cc, m.ebuf.p)
  switch (form * 2 + (picard ? 1 : 0)) {
    case 0: NSFEM_CJ3(0, false); break;
    case 1: NSFEM_CJ3(0, true); break;
    case 2: NSFEM_CJ3(1, false); break;
    case 3: NSFEM_CJ3(1, true); break;
    case 4: NSFEM_CJ3(2, false); break;
    case 5: NSFEM_CJ3(2, true); break;
    case 6: NSFEM_CJ3(3, false); break;
    case 7: NSFEM_CJ3(3, true); break;
    default: throw Error(NSFEM_ERR_ARG, "unknown convective form");
  }
#undef NSFEM_CJ3
  hipLaunchKernelGGL(k3_jac_gather, dim3(grid3((int64_t)p22.nnz * 9)), dim3(kBlock), 0, s, p22.nnz,
                     p22.cptr.p, p22.cidx.p, m.ebuf.p, L, E, cvE, J);
  NSFEM_HIP(hipGetLastError());
}
template <int LIN>
static void launch_conv_cell(hipStream_t s, const MeshDev& m, const double* u, const double* v,
                             double cc, int form) {
  const dim3 grid(grid3(m.n_cells)), block(kBlock);
#define NSFEM_CC3(F) \
  hipLaunchKernelGGL((k3_conv_cell<F, LIN>), grid, block, 0, s, m.n_cells, m.vx.p, m.p2.p, u, v, cc, m.ndst.p, m.rbuf.p)
  switch (form) {
    case 0: NSFEM_CC3(0); break;
    case 1: NSFEM_CC3(1); break;
    case 2: NSFEM_CC3(2); break;
    case 3: NSFEM_CC3(3); break;
    default: throw Error(NSFEM_ERR_ARG, "unknown convective form");
  }
#undef NSFEM_CC3
  NSFEM_HIP(hipGetLastError());
}

void convection_residual_3d(hipStream_t s, const MeshDev& m, const double* u, double cc, double* b,
                            int form) {
  launch_conv_cell<0>(s, m, u, nullptr, cc, form);
  hipLaunchKernelGGL(k3_res_gather, dim3(grid3((int64_t)m.n_p2 * 3)), dim3(kBlock), 0, s, m.n_p2,
                     m.nptr.p, m.rbuf.p, (const uint8_t*)nullptr, b);
  NSFEM_HIP(hipGetLastError());
}

// y += c_c [d conv(u)/du] v  (Newton) or its Picard linearisation: the matrix-free action of the
// convection blocks of the velocity Jacobian
void convection_action_3d(hipStream_t s, const MeshDev& m, const double* u, const double* v,
                          double cc, double* y, int form, bool picard, const uint8_t* skipmask) {
  if (picard) launch_conv_cell<2>(s, m, u, v, cc, form);
  else launch_conv_cell<1>(s, m, u, v, cc, form);
  hipLaunchKernelGGL(k3_res_gather, dim3(grid3((int64_t)m.n_p2 * 3)), dim3(kBlock), 0, s, m.n_p2,
                     m.nptr.p, m.rbuf.p, skipmask, y);
  NSFEM_HIP(hipGetLastError());
}

// ---- CFL diagnostic on tetrahedra (reference source/ns_problem.py:554-587; the 2D kernel is
// k_cfl): cell-local L2 projection onto DG2 of  2 |u| k / h  (h = dolfin CellDiameter = diameter of
// the circumsphere) with the 14-point degree-4 rule, then the max-norm of the 10 coefficients.
__global__ __launch_bounds__(256) void k3_cfl(int nc, const double* __restrict__ vx,
                                              const int32_t* __restrict__ p2,
                                              const double* __restrict__ u, double scale,
                                              double* __restrict__ parts) {
  __shared__ double sh[4];
  double best = 0.0;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
    double x[4][3];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int d = 0; d < 3; ++d) x[v][d] = vx[(size_t)(3 * v + d) * nc + c];
    // circumradius R = sqrt(P (P - aA)(P - bB)(P - cC)) / (6 V), P = (aA + bB + cC) / 2, with
    // (a, A), (b, B), (c, C) the lengths of the three pairs of opposite edges
    auto len = [&](int i, int j) {
      const double dx = x[i][0] - x[j][0], dy = x[i][1] - x[j][1], dz = x[i][2] - x[j][2];
      return sqrt(dx * dx + dy * dy + dz * dz);
    };
    const double pa = len(0, 1) * len(2, 3), pb = len(0, 2) * len(1, 3), pc = len(0, 3) * len(1, 2);
    const double P = 0.5 * (pa + pb + pc);
    const double e1[3] = {x[1][0] - x[0][0], x[1][1] - x[0][1], x[1][2] - x[0][2]};
    const double e2[3] = {x[2][0] - x[0][0], x[2][1] - x[0][1], x[2][2] - x[0][2]};
    const double e3[3] = {x[3][0] - x[0][0], x[3][1] - x[0][1], x[3][2] - x[0][2]};
    const double det = e1[0] * (e2[1] * e3[2] - e2[2] * e3[1]) - e1[1] * (e2[0] * e3[2] - e2[2] * e3[0]) +
                       e1[2] * (e2[0] * e3[1] - e2[1] * e3[0]);
    const double h = 2.0 * sqrt(fmax(P * (P - pa) * (P - pb) * (P - pc), 0.0)) / fabs(det);   // 6 V = |det|
    double uu[10][3];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const double* src = u + 3 * (size_t)p2[(size_t)k * nc + c];
      uu[k][0] = src[0]; uu[k][1] = src[1]; uu[k][2] = src[2];
    }
    double ci[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) ci[i] = 0.0;
    for (int q = 0; q < 14; ++q) {
      double a = 0.0, b = 0.0, d = 0.0;
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        const double ph = c_cfl3.phi[q][k];
        a += ph * uu[k][0]; b += ph * uu[k][1]; d += ph * uu[k][2];
      }
      const double f = scale * sqrt(a * a + b * b + d * d) / h;
#pragma unroll
      for (int i = 0; i < 10; ++i) ci[i] += c_cfl3.inv[i][q] * f;
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) best = fmax(best, fabs(ci[i]));
  }
  for (int off = 32; off > 0; off >>= 1) best = fmax(best, __shfl_xor(best, off));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) parts[blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

void launch_cfl_3d(hipStream_t s, const MeshDev& m, const double* u, double scale, double* parts,
                   int n_parts) {
  hipLaunchKernelGGL(k3_cfl, dim3(n_parts), dim3(256), 0, s, m.n_cells, m.vx.p, m.p2.p, u, scale,
                     parts);
  NSFEM_HIP(hipGetLastError());
}

// nodal coordinates of the P2 nodes (vertices and edge midpoints of the affine tetrahedra; every
// cell sharing a node writes the same value) -- the field x of the Euler acceleration alpha x x
__global__ __launch_bounds__(256) void k3_coord_field(int nc, const double* __restrict__ vx,
                                                      const int32_t* __restrict__ p2,
                                                      double* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  double x[4][3];
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int d = 0; d < 3; ++d) x[v][d] = vx[(size_t)(3 * v + d) * nc + c];
  const int pr[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    double* o = out + 3 * (size_t)p2[(size_t)k * nc + c];
#pragma unroll
    for (int d = 0; d < 3; ++d)
      o[d] = k < 4 ? x[k][d] : 0.5 * (x[pr[k - 4][0]][d] + x[pr[k - 4][1]][d]);
  }
}
void launch_coord_field_3d(hipStream_t s, const MeshDev& m, double* out) {
  hipLaunchKernelGGL(k3_coord_field, dim3(grid3(m.n_cells)), dim3(kBlock), 0, s, m.n_cells, m.vx.p,
                     m.p2.p, out);
  NSFEM_HIP(hipGetLastError());
}

}  // namespace nsfem
