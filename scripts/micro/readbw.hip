// Practical HBM ceilings on MI355X for the access mixes of the SpMV kernels: read-only reduction,
// copy, and "3 reads + 1 write" over GB-sized arrays (>> 256 MiB Infinity Cache), 256-thread blocks,
// grid-stride, 16 B per lane per access.  Build: hipcc -O3 --offload-arch=gfx950 readbw.hip -o readbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_read(size_t n2, const double2* __restrict__ a, double* out) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    double2 v = a[i];
    acc += v.x + v.y;
  }
  if (acc == 12345.678) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_copy(size_t n2, const double2* __restrict__ a, double2* __restrict__ b) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_r3w1(size_t n2, const double2* __restrict__ a, const double2* __restrict__ b,
                                               const double2* __restrict__ c, double2* __restrict__ d) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    double2 x = a[i], y = b[i], z = c[i];
    d[i] = make_double2(x.x + y.x * z.x, x.y + y.y * z.y);
  }
}
int main() {
  const size_t n2 = (size_t)1 << 26;   // 64 Mi double2 = 1 GiB per array
  double2 *a, *b, *c, *d; double* out;
  CK(hipMalloc(&a, n2 * 16)); CK(hipMalloc(&b, n2 * 16)); CK(hipMalloc(&c, n2 * 16)); CK(hipMalloc(&d, n2 * 16));
  CK(hipMalloc(&out, 8));
  CK(hipMemset(a, 0, n2 * 16)); CK(hipMemset(b, 0, n2 * 16)); CK(hipMemset(c, 0, n2 * 16));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {2048, 8192, 32768}) {
    for (int which = 0; which < 3; ++which) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, n2, a, out);
        if (which == 1) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, n2, a, b);
        if (which == 2) hipLaunchKernelGGL(k_r3w1, dim3(grid), dim3(256), 0, 0, n2, a, b, c, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      const double bytes = (double)n2 * 16 * (which == 0 ? 1 : which == 1 ? 2 : 4);
      printf("grid %6d %-8s %.1f us  %.2f TB/s\n", grid, which == 0 ? "read" : which == 1 ? "copy" : "r3w1", best * 1e3,
             bytes / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
