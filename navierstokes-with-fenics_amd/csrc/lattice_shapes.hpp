// Canonical interior stencils of right-diagonal lattices, shared by the lattice kernels (k_cheb_lattice in linalg.hip,
// k_jac_lattice in assembly.hip): rows of such shape read their neighbours at compile-time LDS offsets.
#pragma once

namespace nsfem {

// The shapes in dictionary order (ascending column = ascending
// (dj, di)): the P2 operator's four parity classes (pi, pj) = (0,0) vertex, (1,0), (0,1), (1,1) edge midpoints, and
// the 7-point P1 stencil (the same shape for every class).
template <int SHAPE, int CLS> struct LatShape;
#define NSFEM_LAT_SHAPE(SH, CL, NN, ...)                                   \
  template <> struct LatShape<SH, CL> {                                    \
    static constexpr int N = NN;                                           \
    static constexpr int d[NN][2] = {__VA_ARGS__};   /* (dj, di) */        \
  };
NSFEM_LAT_SHAPE(1, 0, 19, {-2, -2}, {-2, -1}, {-2, 0}, {-1, -2}, {-1, -1}, {-1, 0}, {-1, 1}, {0, -2}, {0, -1}, {0, 0}, {0, 1},
                {0, 2}, {1, -1}, {1, 0}, {1, 1}, {1, 2}, {2, 0}, {2, 1}, {2, 2})
NSFEM_LAT_SHAPE(1, 1, 9, {-2, -1}, {-1, -1}, {-1, 0}, {0, -1}, {0, 0}, {0, 1}, {1, 0}, {1, 1}, {2, 1})
NSFEM_LAT_SHAPE(1, 2, 9, {-1, -2}, {-1, -1}, {-1, 0}, {0, -1}, {0, 0}, {0, 1}, {1, 0}, {1, 1}, {1, 2})
NSFEM_LAT_SHAPE(1, 3, 9, {-1, -1}, {-1, 0}, {-1, 1}, {0, -1}, {0, 0}, {0, 1}, {1, -1}, {1, 0}, {1, 1})
NSFEM_LAT_SHAPE(2, 0, 7, {-1, -1}, {-1, 0}, {0, -1}, {0, 0}, {0, 1}, {1, 0}, {1, 1})
NSFEM_LAT_SHAPE(2, 1, 7, {-1, -1}, {-1, 0}, {0, -1}, {0, 0}, {0, 1}, {1, 0}, {1, 1})
NSFEM_LAT_SHAPE(2, 2, 7, {-1, -1}, {-1, 0}, {0, -1}, {0, 0}, {0, 1}, {1, 0}, {1, 1})
NSFEM_LAT_SHAPE(2, 3, 7, {-1, -1}, {-1, 0}, {0, -1}, {0, 0}, {0, 1}, {1, 0}, {1, 1})
#undef NSFEM_LAT_SHAPE
// LDS offset (in nodes) of the neighbour (dj, di) of a node of class cls in the class-split tile with planes of
// 32 x ehh words -- the formula of lattice_offsets, at compile time
__host__ __device__ constexpr int lat_fl2(int v) { return v >= 0 ? v / 2 : -((1 - v) / 2); }
__host__ __device__ constexpr int lat_fixed_off(int cls, int dj, int di, int ehh) {
  const int pi = cls & 1, pj = (cls >> 1) & 1;
  const int c2 = ((pi + di) & 1) | (((pj + dj) & 1) << 1);
  return (c2 - cls) * 32 * ehh + lat_fl2(pj + dj) * 32 + lat_fl2(pi + di);
}


}  // namespace nsfem
