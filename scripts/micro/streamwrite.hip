// Round trip device -> host after a kernel: (a) a one-block publish kernel that stores a flag to coherent pinned
// memory, (b) hipStreamWriteValue64 on the same stream (a command-processor write, no kernel dispatch).
// build: hipcc --offload-arch=gfx950 -O2 scripts/micro/streamwrite.hip -o /tmp/streamwrite
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_work(double* x, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = x[i] * 1.0000001 + 1.0; }
__global__ void k_flag(volatile uint64_t* f, uint64_t v) { if (threadIdx.x == 0) { __threadfence_system(); *f = v; } }
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  double* x; const int n = 1 << 21; CK(hipMalloc(&x, n * sizeof(double))); CK(hipMemset(x, 0, n * sizeof(double)));
  uint64_t* f; CK(hipHostMalloc(&f, 64, hipHostMallocCoherent)); *f = 0;
  auto wait = [&](uint64_t v) { while (__atomic_load_n(f, __ATOMIC_ACQUIRE) != v) {} };
  for (int mode = 0; mode < 2; ++mode) {
    const int reps = 2000;
    uint64_t seq = (uint64_t)mode << 32;
    for (int w = 0; w < 50; ++w) {
      hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, x, n);
      ++seq;
      if (mode == 0) hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, (volatile uint64_t*)f, seq);
      else CK(hipStreamWriteValue64(s, f, seq, 0));
      wait(seq);
    }
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, x, n);
      ++seq;
      if (mode == 0) hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, (volatile uint64_t*)f, seq);
      else CK(hipStreamWriteValue64(s, f, seq, 0));
      wait(seq);
    }
    auto t1 = std::chrono::steady_clock::now();
    std::printf("%s: %.2f us per (kernel + flag + host wait)\n", mode == 0 ? "publish kernel      " : "hipStreamWriteValue64",
                std::chrono::duration<double, std::micro>(t1 - t0).count() / reps);
  }
  return 0;
}
