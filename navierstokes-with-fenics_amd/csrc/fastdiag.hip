// Fast diagonalisation of the pressure Poisson operator on tensor-product lattices (poisson_fd.py builds the
// factors on the host).  The reference solves the projection step with a sparse LU every step
// (source/ns_ipcs_solver.py:160-171); on the right-diagonal triangulation of a rectangle the P1 stiffness matrix is the
// tensor sum  A = K_y (x) W_x + W_y (x) K_x  of 1D stiffness and lumped mass matrices, so with the generalised
// eigenvectors V_x, V_y of the two directions
//
//     z = A^+ r = V_y ( (V_y^T R V_x) .* inv ) V_x^T ,      R = r as an H x W array,  inv_ji = 1 / (lam_y,j + lam_x,i)
//
// -- four dense products with (n + 1)-sized matrices, the one GEMM-shaped operation of the hot path.  They run on the
// matrix cores: v_mfma_f64_16x16x4_f64, one 16 x 16 accumulator tile per wavefront, operands staged through LDS
// (rows padded to 48 doubles: the four k-lines a wave reads land on disjoint halves of the 64 banks), the next
// k-block's global loads in flight under the current block's MFMAs.  (On the MI355X the fp64 matrix rate equals the
// vector rate; what the matrix instruction buys here is 16 x fewer issued instructions and 8 x fewer LDS reads per
// FMA than a register-tiled VALU kernel: these 513-sized products are latency bound, not flop bound.)
#include "nsfem_internal.hpp"

namespace nsfem {

typedef double fd_acc4 __attribute__((ext_vector_type(4)));

constexpr int kFdBM = 32, kFdBN = 32, kFdBK = 96, kFdLd = kFdBK + 2, kFdNQ = kFdBM * kFdBK / 256;

// C[M x N] = op(A)[M x K] * op(B)[K x N] (.* scale[M x N]); row-major storage.  TA: A(m, k) = A[k * lda + m];
// TB: B(k, n) = B[n * ldb + k].  256 threads = 4 waves, wave w owns the 16 x 16 tile (w >> 1, w & 1) of a 32 x 32
// block of C.  The products are latency bound (K = 513: six k-blocks of 96, every one a global round trip): a thread
// keeps the 2 x 12 loads of the NEXT TWO blocks in flight (register double buffer) under the 24 MFMAs of a block.  LDS: As[m][k], Bs[n][k]
// with lines of 98 doubles -- the fragment reads (16 lines x 4 consecutive k per wave) and the stores along k are
// conflict free.
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_fd_gemm(int M, int N, int K, const double* __restrict__ A, int lda,
                                                 const double* __restrict__ B, int ldb, double* __restrict__ C, int ldc,
                                                 const double* __restrict__ scale) {
  __shared__ double As[kFdBM * kFdLd];      // As[m][k]
  __shared__ double Bs[kFdBN * kFdLd];      // Bs[n][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * kFdBM, n0 = blockIdx.x * kFdBN;
  const int wm = (wave >> 1) * 16, wn = (wave & 1) * 16;
  // element e = tid + 256 q of a 32 x 96 block, walked along the operand's contiguous direction:
  //   contiguous along k (A row-major, B transposed): k = e % 96, line = e / 96
  //   contiguous along m / n (A transposed, B row-major): line = e % 32, k = e / 32
  double ra0[kFdNQ], rb0[kFdNQ], ra1[kFdNQ], rb1[kFdNQ];      // two k-blocks in flight (register double buffer)
  auto load_block = [&](int k0, double (&ra)[kFdNQ], double (&rb)[kFdNQ]) {
#pragma unroll
    for (int q = 0; q < kFdNQ; ++q) {
      const int e = tid + 256 * q;
      {
        const int m = TA ? (e & 31) : (e / kFdBK), k = TA ? (e >> 5) : (e % kFdBK);
        const int gm = m0 + m, gk = k0 + k;
        ra[q] = (gm < M && gk < K) ? (TA ? A[(size_t)gk * lda + gm] : A[(size_t)gm * lda + gk]) : 0.0;
      }
      {
        const int n = TB ? (e / kFdBK) : (e & 31), k = TB ? (e % kFdBK) : (e >> 5);
        const int gn = n0 + n, gk = k0 + k;
        rb[q] = (gn < N && gk < K) ? (TB ? B[(size_t)gn * ldb + gk] : B[(size_t)gk * ldb + gn]) : 0.0;
      }
    }
  };
  auto store_block = [&](const double (&ra)[kFdNQ], const double (&rb)[kFdNQ]) {
#pragma unroll
    for (int q = 0; q < kFdNQ; ++q) {
      const int e = tid + 256 * q;
      const int ma = TA ? (e & 31) : (e / kFdBK), ka = TA ? (e >> 5) : (e % kFdBK);
      As[ma * kFdLd + ka] = ra[q];
      const int nb = TB ? (e / kFdBK) : (e & 31), kb = TB ? (e % kFdBK) : (e >> 5);
      Bs[nb * kFdLd + kb] = rb[q];
    }
  };
  // (two accumulator chains and the block's 2 x 24 fragments read before its first MFMA: one chain waits for every
  // product and, just in time, for its LDS read)
  fd_acc4 acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
  auto block_products = [&](const double* __restrict__ ap, const double* __restrict__ bp) {
    double af[kFdBK / 4], bf[kFdBK / 4];
#pragma unroll
    for (int i = 0; i < kFdBK / 4; ++i) {
      af[i] = ap[4 * i];
      bf[i] = bp[4 * i];
    }
#pragma unroll
    for (int i = 0; i < kFdBK / 4; i += 2) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[i], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i + 1], bf[i + 1], acc2, 0, 0, 0);
    }
  };
  const double* __restrict__ ap = As + (wm + (lane & 15)) * kFdLd + (lane >> 4);       // A[m = lane & 15][k = lane >> 4]
  const double* __restrict__ bp = Bs + (wn + (lane & 15)) * kFdLd + (lane >> 4);       // B[k = lane >> 4][n = lane & 15]
  load_block(0, ra0, rb0);
  if (kFdBK < K) load_block(kFdBK, ra1, rb1);
  for (int k0 = 0; k0 < K; k0 += 2 * kFdBK) {
    __syncthreads();                         // (the previous block's fragments have been read)
    store_block(ra0, rb0);
    __syncthreads();
    if (k0 + 2 * kFdBK < K) load_block(k0 + 2 * kFdBK, ra0, rb0);
    block_products(ap, bp);
    if (k0 + kFdBK >= K) break;
    __syncthreads();
    store_block(ra1, rb1);
    __syncthreads();
    if (k0 + 3 * kFdBK < K) load_block(k0 + 3 * kFdBK, ra1, rb1);
    block_products(ap, bp);
  }
  acc += acc2;
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  const int col = n0 + wn + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + wm + (lane >> 4) + 4 * r;
    if (row < M && col < N) {
      double v = acc[r];
      if (scale) v *= scale[(size_t)row * N + col];
      C[(size_t)row * ldc + col] = v;
    }
  }
}

template <bool TA, bool TB>
static void fd_gemm(hipStream_t s, int M, int N, int K, const double* A, int lda, const double* B, int ldb, double* C,
                    int ldc, const double* scale) {
  const dim3 grid((N + kFdBN - 1) / kFdBN, (M + kFdBM - 1) / kFdBM), block(256);
  hipLaunchKernelGGL((k_fd_gemm<TA, TB>), grid, block, 0, s, M, N, K, A, lda, B, ldb, C, ldc, scale);
  NSFEM_HIP(hipGetLastError());
}

void FastDiag::set(hipStream_t s, int W_, int H_, const double* vx, const double* vy, const double* inv_) {
  NSFEM_REQUIRE(W_ >= 2 && H_ >= 2 && vx && vy && inv_, "fast diagonalisation: bad factors");
  W = W_;
  H = H_;
  j0 = h_loc = 0;
  Vx.upload(vx, (size_t)W * W, s);
  Vy.upload(vy, (size_t)H * H, s);
  inv.upload(inv_, (size_t)H * W, s);
  t1.alloc((size_t)H * W);
  t2.alloc((size_t)H * W);
  NSFEM_HIP(hipStreamSynchronize(s));
}

// Strips of a partitioned mesh (rank r holds the lattice lines j0 ... j0 + h_loc - 1, ghost lines included; every
// node is OWNED by one rank and r vanishes on the ghost rows): the contraction over y is a sum over the ranks,
//
//     T2 = sum_ranks V_y[lines of the rank, :]^T (R_loc V_x)          one all-reduce of H x W doubles
//     Z_loc = (V_y[lines of the rank, :] (T2 .* inv)) V_x^T           every local line, ghost lines included
//
// -- the whole projection solve of an N-rank run costs ONE collective (the multigrid-CG solve it replaces: ~10 halo
// exchanges + 2 small all-reduces per iteration), the four products shrink to 1 / N of their size, and the result
// comes with valid ghost rows (no exchange afterwards).
void FastDiag::set_rows(hipStream_t s, int W_, int H_, int j0_, int h_loc_, const double* vx, const double* vy,
                        const double* inv_) {
  NSFEM_REQUIRE(W_ >= 2 && H_ >= 2 && vx && vy && inv_ && j0_ >= 0 && h_loc_ >= 1 && j0_ + h_loc_ <= H_,
                "fast diagonalisation: bad factors / lines");
  W = W_;
  H = H_;
  j0 = j0_;
  h_loc = h_loc_;
  Vx.upload(vx, (size_t)W * W, s);
  Vy.upload(vy + (size_t)j0 * H, (size_t)h_loc * H, s);          // rows j0 ... of the row-major H x H matrix
  inv.upload(inv_, (size_t)H * W, s);
  t1.alloc((size_t)h_loc * W);
  t2.alloc((size_t)H * W);
  NSFEM_HIP(hipStreamSynchronize(s));
}

__global__ __launch_bounds__(256) void k_fd_scale(int64_t n, const double* __restrict__ a, double* __restrict__ x) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= a[i];
}

void FastDiag::apply_strip(hipStream_t s, Comm* comm, const double* r, double* z) {
  NSFEM_REQUIRE(ready() && strip() && comm, "fast diagonalisation: strip factors / communicator not set");
  fd_gemm<false, false>(s, h_loc, W, W, r, W, Vx.p, W, t1.p, W, nullptr);          // T1 = R_loc V_x
  fd_gemm<true, false>(s, H, W, h_loc, Vy.p, H, t1.p, W, t2.p, W, nullptr);         // T2 = V_y[loc, :]^T T1  (partial)
  comm->allreduce_sum(s, t2.p, (int64_t)H * W);
  const int64_t n = (int64_t)H * W;
  hipLaunchKernelGGL(k_fd_scale, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0, s, n,
                     (const double*)inv.p, t2.p);
  NSFEM_HIP(hipGetLastError());
  fd_gemm<false, false>(s, h_loc, W, H, Vy.p, H, t2.p, W, t1.p, W, nullptr);        // U = V_y[loc, :] (T2 .* inv)
  fd_gemm<false, true>(s, h_loc, W, W, t1.p, W, Vx.p, W, z, W, nullptr);            // Z_loc = U V_x^T
  ++applications;
}

// z = V_y ((V_y^T (R V_x)) .* inv) V_x^T
void FastDiag::apply(hipStream_t s, const double* r, double* z) {
  NSFEM_REQUIRE(ready() && !strip(), "fast diagonalisation: factors not set (or set for a strip)");
  fd_gemm<false, false>(s, H, W, W, r, W, Vx.p, W, t1.p, W, nullptr);            // T1 = R V_x
  fd_gemm<true, false>(s, H, W, H, Vy.p, H, t1.p, W, t2.p, W, inv.p);            // T2 = (V_y^T T1) .* inv
  fd_gemm<false, true>(s, H, W, W, t2.p, W, Vx.p, W, t1.p, W, nullptr);          // T1 = T2 V_x^T
  fd_gemm<false, false>(s, H, W, H, Vy.p, H, t1.p, W, z, W, nullptr);            // Z  = V_y T1
  ++applications;
}

}  // namespace nsfem
