// Scratch micro-benchmark: what does a dependent kernel boundary cost in a stream on the MI355X, by launch mode?
//   * plain launches of a small kernel (grid g), eager
//   * the same chain captured in a HIP graph
//   * a chain that alternates a streaming kernel (writes W MB) with a small one
// Run with HIP_FORCE_DEV_KERNARG=0/1 to see the kernel-argument placement.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
__global__ void k_small(double* x, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += 1.0;
}
__global__ void k_stream(const double* __restrict__ a, double* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i] * 1.0000001;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t N = 1 << 21;                  // 16.8 MB vectors
  double *x, *a, *b;
  hipMalloc(&x, sizeof(double) * N); hipMalloc(&a, sizeof(double) * N); hipMalloc(&b, sizeof(double) * N);
  hipMemset(x, 0, sizeof(double) * N); hipMemset(a, 0, sizeof(double) * N);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 2000;
  for (int grid : {1, 16, 256, 1024}) {
    for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, s, x, grid * 256);
    hipStreamSynchronize(s);
    const double t0 = now();
    hipEventRecord(e0, s);
    for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, s, x, grid * 256);
    hipEventRecord(e1, s);
    const double t1 = now();
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("eager  grid %4d: %.2f us per launch on the GPU, %.2f us host per launch\n", grid, 1e3 * ms / reps, 1e6 * (t1 - t0) / reps);
  }
  // graph of 200 small launches
  for (int grid : {16, 256}) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
    for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, s, x, grid * 256);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int k = 0; k < 10; ++k) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("graph  grid %4d: %.2f us per launch\n", grid, 1e3 * ms / 2000);
  }
  // streaming kernel + small kernel alternating
  {
    for (int k = 0; k < 20; ++k) { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s, a, b, N); hipLaunchKernelGGL(k_small, dim3(16), dim3(256), 0, s, x, 4096); }
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int k = 0; k < 500; ++k) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s, a, b, N);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms0; hipEventElapsedTime(&ms0, e0, e1);
    hipEventRecord(e0, s);
    for (int k = 0; k < 500; ++k) { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s, a, b, N); hipLaunchKernelGGL(k_small, dim3(16), dim3(256), 0, s, x, 4096); }
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms1; hipEventElapsedTime(&ms1, e0, e1);
    printf("stream kernel alone %.2f us; + small kernel after it: +%.2f us per pair\n", 1e3 * ms0 / 500, 1e3 * (ms1 - ms0) / 500);
  }
  return 0;
}
