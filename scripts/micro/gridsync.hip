// Scratch micro-benchmark: cost of a cooperative-groups grid barrier on MI355X as a function of
// the number of workgroups (is a persistent multi-stage kernel cheaper than separate launches?)
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
__global__ void k_sync(int n, double* x) {
  cg::grid_group g = cg::this_grid();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double v = x[i];
  for (int k = 0; k < n; ++k) {
    v = v * 1.0000001 + 1.0;
    x[i] = v;
    g.sync();
    v += x[(i + 64) % (gridDim.x * blockDim.x)] * 1e-9;
  }
  x[i] = v;
}
__global__ void k_empty(double* x) { x[blockIdx.x * blockDim.x + threadIdx.x] += 1.0; }
int main() {
  double* x;
  hipMalloc(&x, sizeof(double) * 4096 * 256);
  hipMemset(x, 0, sizeof(double) * 4096 * 256);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int blocks : {32, 64, 128, 256, 512, 1024}) {
    int n = 200;
    void* args[] = {&n, &x};
    hipError_t e = hipLaunchCooperativeKernel((void*)k_sync, dim3(blocks), dim3(256), args, 0, 0);
    if (e != hipSuccess) { printf("blocks %d: %s\n", blocks, hipGetErrorString(e)); continue; }
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchCooperativeKernel((void*)k_sync, dim3(blocks), dim3(256), args, 0, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("blocks %4d: %.2f us per grid sync\n", blocks, 1e3 * ms / n);
  }
  hipEventRecord(a);
  for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, 0, x);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("back-to-back dependent launches: %.2f us each\n", 1e3 * ms / 200);
  return 0;
}
