// Boundary functionals of the discrete solution: surface force (drag / lift), mass flux and
// measure of a set of boundary facets.
//
// Replaces the dolfin.assemble(... * ds(subdomain_id)) calls of the reference's post-processing
// hooks -- demo/dfg_benchmark.py:44-66 (traction = -p n + 1/Re sym(grad u) n integrated over the
// cylinder), demo/gravity_driven_flow.py:66-70 (total mass flux dot(n, u) ds) -- with one thread
// per boundary facet: the adjacent cell's geometry, the 6 / 10 nodal velocities and the 3 / 4
// nodal pressures are gathered once, the integrand is evaluated at the points of a facet rule
// that is exact for it (grad u and p are linear on an affine facet, u.n is quadratic):
// 2D: 2-point Gauss rule on the edge, 3D: the 3 edge midpoints of the face.  Per-facet results
// are stored and summed by the caller in facet order (deterministic, no atomics).
#include "nsfem_internal.hpp"

namespace nsfem {

// P2 basis values and physical gradients at barycentric point lam[] of a simplex with
// barycentric-coordinate gradients gl[v][d]; local order = UFC (vertices, then edges opposite-
// ordered as in fem_mesh.Mesh: triangle e(12), e(02), e(01); tetrahedron e(23), e(13), e(12),
// e(03), e(02), e(01))
template <int DIM>
__device__ __forceinline__ void p2_at(const double* lam, const double (*gl)[3], double* phi,
                                       double (*dphi)[3]) {
  constexpr int NV = DIM + 1;
  constexpr int NE = DIM == 2 ? 3 : 6;
  const int ea2[3] = {1, 0, 0}, eb2[3] = {2, 2, 1};
  const int ea3[6] = {2, 1, 1, 0, 0, 0}, eb3[6] = {3, 3, 2, 3, 2, 1};
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    phi[v] = lam[v] * (2.0 * lam[v] - 1.0);
#pragma unroll
    for (int d = 0; d < DIM; ++d) dphi[v][d] = (4.0 * lam[v] - 1.0) * gl[v][d];
  }
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int a = DIM == 2 ? ea2[e] : ea3[e], b = DIM == 2 ? eb2[e] : eb3[e];
    phi[NV + e] = 4.0 * lam[a] * lam[b];
#pragma unroll
    for (int d = 0; d < DIM; ++d) dphi[NV + e][d] = 4.0 * (lam[a] * gl[b][d] + lam[b] * gl[a][d]);
  }
}

// out[f][0..DIM-1] = int_f ( -p n + nu (grad u + sym grad u^T) n ) dS
// out[f][DIM] = int_f u.n dS ; out[f][DIM+1] = |f|
template <int DIM>
__global__ __launch_bounds__(256) void k_boundary_force(int nf, int nc,
                                                        const int32_t* __restrict__ fcell,
                                                        const int32_t* __restrict__ flocal,
                                                        const double* __restrict__ vx,
                                                        const int32_t* __restrict__ p2,
                                                        const int32_t* __restrict__ p1,
                                                        const double* __restrict__ u,
                                                        const double* __restrict__ p, double nu,
                                                        double sym, double* __restrict__ out) {
  constexpr int NV = DIM + 1, N2 = DIM == 2 ? 6 : 10, NQ = DIM == 2 ? 2 : 3;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const int c = fcell[f], opp = flocal[f];
  double x[NV][3];
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int d = 0; d < DIM; ++d) x[v][d] = vx[(size_t)(DIM * v + d) * nc + c];
  // gradients of the barycentric coordinates: rows of J^-1 (J = [x1-x0, .., xd-x0]) for
  // lambda_1..d, lambda_0 = 1 - sum
  double gl[NV][3];
  double vol;   // |T| * d!
  if (DIM == 2) {
    const double a00 = x[1][0] - x[0][0], a01 = x[2][0] - x[0][0];
    const double a10 = x[1][1] - x[0][1], a11 = x[2][1] - x[0][1];
    const double det = a00 * a11 - a01 * a10, id = 1.0 / det;
    gl[1][0] = a11 * id;  gl[1][1] = -a01 * id;
    gl[2][0] = -a10 * id; gl[2][1] = a00 * id;
    gl[0][0] = -gl[1][0] - gl[2][0];
    gl[0][1] = -gl[1][1] - gl[2][1];
    vol = fabs(det);
  } else {
    double a[3][3];   // columns = edge vectors
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      a[d][0] = x[1][d] - x[0][d];
      a[d][1] = x[2][d] - x[0][d];
      a[d][2] = x[3][d] - x[0][d];
    }
    const double c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1];
    const double c01 = a[1][2] * a[2][0] - a[1][0] * a[2][2];
    const double c02 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
    const double det = a[0][0] * c00 + a[0][1] * c01 + a[0][2] * c02, id = 1.0 / det;
    // inverse = adjugate / det ; row k of the inverse = grad lambda_(k+1)
    gl[1][0] = c00 * id;
    gl[1][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) * id;
    gl[1][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) * id;
    gl[2][0] = c01 * id;
    gl[2][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) * id;
    gl[2][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) * id;
    gl[3][0] = c02 * id;
    gl[3][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) * id;
    gl[3][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) * id;
#pragma unroll
    for (int d = 0; d < 3; ++d) gl[0][d] = -gl[1][d] - gl[2][d] - gl[3][d];
    vol = fabs(det);
  }
  // outward unit normal of the facet opposite vertex `opp` and its measure:
  // grad lambda_opp points inward, |f| = |grad lambda_opp| * d * |T| = |grad lambda_opp| vol / (d-1)!
  double gn = 0.0, nrm[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int v = 0; v < NV; ++v)
    if (v == opp) {
#pragma unroll
      for (int d = 0; d < DIM; ++d) nrm[d] = -gl[v][d];
    }
#pragma unroll
  for (int d = 0; d < DIM; ++d) gn += nrm[d] * nrm[d];
  gn = sqrt(gn);
#pragma unroll
  for (int d = 0; d < DIM; ++d) nrm[d] /= gn;
  const double area = gn * vol / (DIM == 2 ? 1.0 : 2.0);
  // nodal data
  double un[N2][3], pn[NV];
#pragma unroll
  for (int k = 0; k < N2; ++k) {
    const int node = p2[(size_t)k * nc + c];
#pragma unroll
    for (int d = 0; d < DIM; ++d) un[k][d] = u[(size_t)node * DIM + d];
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) pn[v] = p[p1[(size_t)v * nc + c]];
  // facet rule in barycentric coordinates of the facet's own vertices (all cell vertices but opp)
  const double g2 = 0.21132486540518713;              // (1 - 1/sqrt(3)) / 2
  double force[3] = {0.0, 0.0, 0.0}, flux = 0.0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    double fl[3];                                       // barycentric point inside the facet
    if (DIM == 2) {
      fl[0] = q == 0 ? g2 : 1.0 - g2;
      fl[1] = 1.0 - fl[0];
      fl[2] = 0.0;
    } else {
      fl[0] = q == 0 ? 0.0 : 0.5;
      fl[1] = q == 1 ? 0.0 : 0.5;
      fl[2] = q == 2 ? 0.0 : 0.5;
    }
    double lam[NV];
    int t = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (v == opp) lam[v] = 0.0;
      else { lam[v] = t == 0 ? fl[0] : (t == 1 ? fl[1] : fl[2]); ++t; }
    }
    double phi[N2], dphi[N2][3];
    p2_at<DIM>(lam, gl, phi, dphi);
    double uq[3] = {0.0, 0.0, 0.0}, G[3][3] = {{0.0}}, pq = 0.0;
#pragma unroll
    for (int k = 0; k < N2; ++k)
#pragma unroll
      for (int a = 0; a < DIM; ++a) {
        uq[a] += phi[k] * un[k][a];
#pragma unroll
        for (int b = 0; b < DIM; ++b) G[a][b] += un[k][a] * dphi[k][b];      // d_b u_a
      }
#pragma unroll
    for (int v = 0; v < NV; ++v) pq += lam[v] * pn[v];
    const double w = area / NQ;
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
      double s = -pq * nrm[a];
#pragma unroll
      for (int b = 0; b < DIM; ++b) s += nu * (G[a][b] + sym * G[b][a]) * nrm[b];
      force[a] += w * s;
      flux += w * uq[a] * nrm[a];
    }
  }
  double* o = out + (size_t)f * (DIM + 2);
#pragma unroll
  for (int a = 0; a < DIM; ++a) o[a] = force[a];
  o[DIM] = flux;
  o[DIM + 1] = area;
}

void launch_boundary_force(hipStream_t s, const MeshDev& m, int nf, const int32_t* fcell,
                           const int32_t* flocal, const double* u, const double* p, double nu,
                           double sym, double* out) {
  if (nf <= 0) return;
  const int grid = (nf + 255) / 256;
  if (m.dim == 2)
    hipLaunchKernelGGL((k_boundary_force<2>), dim3(grid), dim3(256), 0, s, nf, m.n_cells, fcell, flocal,
                       m.vx.p, m.p2.p, m.p1.p, u, p, nu, sym, out);
  else
    hipLaunchKernelGGL((k_boundary_force<3>), dim3(grid), dim3(256), 0, s, nf, m.n_cells, fcell, flocal,
                       m.vx.p, m.p2.p, m.p1.p, u, p, nu, sym, out);
  NSFEM_HIP(hipGetLastError());
}

}  // namespace nsfem
