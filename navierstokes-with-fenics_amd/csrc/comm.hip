// Communicators of the partitioned (multi-GPU) path.
//
// The reference has no communication layer (single process).  One
// process per GPU owns a strip of the mesh; the data path needs exactly two primitives:
//   * all-reduce (sum / max) of a few KB of per-block partial dot products, and
//   * a neighbour exchange of contiguous halo ranges (<= 2 neighbours, strips).
// RcclComm issues them on the context's HIP stream through RCCL (ncclAllReduce and grouped
// ncclSend/ncclRecv: point-to-point over the dedicated xGMI link of the neighbour, not a ring
// collective).  LocalComm implements the same interface for several contexts that live in ONE
// process on ONE device (one host thread per rank): it exists so that the partitioned algorithm
// can be tested on a single-GPU box.
#include "nsfem_internal.hpp"
#include <rccl/rccl.h>
#include <condition_variable>
#include <mutex>

namespace nsfem {

// ------------------------------------------------------------------ LocalComm
constexpr int kMaxLocalRanks = 16;

struct LocalGroup {
  int size = 1;
  std::mutex mu;
  std::condition_variable cv;
  int count = 0;
  uint64_t gen = 0;
  double* ptr[kMaxLocalRanks] = {nullptr};
  HaloRange halo[kMaxLocalRanks];
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t g = gen;
    if (++count == size) {
      count = 0;
      ++gen;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return gen != g; });
    }
  }
};

struct PtrPack {
  const double* p[kMaxLocalRanks];
};

__global__ __launch_bounds__(256) void k_reduce_ranks(int64_t n, int size, PtrPack in, int op,
                                                      double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    double acc = in.p[0][i];
    for (int r = 1; r < size; ++r) acc = op ? fmax(acc, in.p[r][i]) : acc + in.p[r][i];
    out[i] = acc;
  }
}

__global__ __launch_bounds__(256) void k_add_range(int64_t n, const double* __restrict__ src,
                                                   double* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    dst[i] += src[i];
}
static void add_range(hipStream_t s, int64_t n, const double* src, double* dst) {
  if (n <= 0) return;
  const int grid = (int)std::min<int64_t>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(k_add_range, dim3(grid), dim3(256), 0, s, n, src, dst);
  NSFEM_HIP(hipGetLastError());
}

struct LocalComm : Comm {
  LocalGroup* g = nullptr;
  DevBuf<double> scratch;
  void reduce(hipStream_t s, double* dev, int64_t count, int op) {
    count_allreduce(count);
    NSFEM_HIP(hipStreamSynchronize(s));
    g->ptr[rank] = dev;
    g->barrier();
    if (scratch.n < (size_t)count) scratch.alloc((size_t)count);
    PtrPack pk;
    for (int r = 0; r < kMaxLocalRanks; ++r) pk.p[r] = r < size ? g->ptr[r] : nullptr;
    int grid = (int)std::min<int64_t>((count + 255) / 256, 1024);
    hipLaunchKernelGGL(k_reduce_ranks, dim3(grid), dim3(256), 0, s, count, size, pk, op, scratch.p);
    NSFEM_HIP(hipGetLastError());
    NSFEM_HIP(hipStreamSynchronize(s));
    g->barrier();      // everybody has read everybody's input
    NSFEM_HIP(hipMemcpyAsync(dev, scratch.p, sizeof(double) * count, hipMemcpyDeviceToDevice, s));
  }
  void allreduce_sum(hipStream_t s, double* dev, int64_t count) override { reduce(s, dev, count, 0); }
  void allreduce_max(hipStream_t s, double* dev, int64_t count) override { reduce(s, dev, count, 1); }
  void exchange(hipStream_t s, const HaloRange& h, double* vec, int width) override {
    count_exchange(h, width);
    NSFEM_HIP(hipStreamSynchronize(s));
    g->ptr[rank] = vec;
    g->halo[rank] = h;
    g->barrier();
    const int above = up(), below = down();
    if (h.recv_above_cnt > 0 && above >= 0) {
      const HaloRange& o = g->halo[above];
      NSFEM_REQUIRE(o.send_down_cnt == h.recv_above_cnt, "halo size mismatch (above)");
      NSFEM_HIP(hipMemcpyAsync(vec + h.recv_above_off * width,
                               g->ptr[above] + o.send_down_off * width,
                               sizeof(double) * h.recv_above_cnt * width, hipMemcpyDeviceToDevice, s));
    }
    if (h.recv_below_cnt > 0 && below >= 0) {
      const HaloRange& o = g->halo[below];
      NSFEM_REQUIRE(o.send_up_cnt == h.recv_below_cnt, "halo size mismatch (below)");
      NSFEM_HIP(hipMemcpyAsync(vec + h.recv_below_off * width,
                               g->ptr[below] + o.send_up_off * width,
                               sizeof(double) * h.recv_below_cnt * width, hipMemcpyDeviceToDevice, s));
    }
    NSFEM_HIP(hipStreamSynchronize(s));
    g->barrier();      // neighbours may now overwrite their send ranges
  }
  void exchange_add(hipStream_t s, const HaloRange& h, double* vec, int width) override {
    count_exchange_add(h, width);
    NSFEM_HIP(hipStreamSynchronize(s));
    g->ptr[rank] = vec;
    g->halo[rank] = h;
    g->barrier();
    const int above = up(), below = down();
    // pull: my send-up range is the rank above's ghost range below, and vice versa
    if (h.send_up_cnt > 0 && above >= 0) {
      const HaloRange& o = g->halo[above];
      NSFEM_REQUIRE(o.recv_below_cnt == h.send_up_cnt, "halo size mismatch (above, add)");
      add_range(s, h.send_up_cnt * width, g->ptr[above] + o.recv_below_off * width,
                vec + h.send_up_off * width);
    }
    if (h.send_down_cnt > 0 && below >= 0) {
      const HaloRange& o = g->halo[below];
      NSFEM_REQUIRE(o.recv_above_cnt == h.send_down_cnt, "halo size mismatch (below, add)");
      add_range(s, h.send_down_cnt * width, g->ptr[below] + o.recv_above_off * width,
                vec + h.send_down_off * width);
    }
    NSFEM_HIP(hipStreamSynchronize(s));
    g->barrier();      // neighbours may now overwrite their ghost ranges
  }
};

// ------------------------------------------------------------------- RcclComm
#define NSFEM_NCCL(expr)                                                                  \
  do {                                                                                    \
    ncclResult_t r_ = (expr);                                                             \
    if (r_ != ncclSuccess)                                                                \
      throw ::nsfem::Error(NSFEM_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
  } while (0)

struct RcclComm : Comm {
  ncclComm_t comm = nullptr;
  DevBuf<double> stage;          // receive buffer of exchange_add
  ~RcclComm() override {
    if (comm) (void)ncclCommDestroy(comm);
  }
  void allreduce_sum(hipStream_t s, double* dev, int64_t count) override {
    count_allreduce(count);
    NSFEM_NCCL(ncclAllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, comm, s));
  }
  void allreduce_max(hipStream_t s, double* dev, int64_t count) override {
    count_allreduce(count);
    NSFEM_NCCL(ncclAllReduce(dev, dev, (size_t)count, ncclDouble, ncclMax, comm, s));
  }
  void exchange(hipStream_t s, const HaloRange& h, double* vec, int width) override {
    count_exchange(h, width);
    const int above = up(), below = down();
    NSFEM_REQUIRE(!(periodic && size == 1), "a periodic partition needs at least two RCCL ranks");
    // sends first (up, then down), receives in the order the peers send (from below = its
    // send-up, then from above = its send-down): with two ranks of a periodic partition both
    // neighbours are the SAME peer and RCCL matches the messages of a pair in issue order
    NSFEM_NCCL(ncclGroupStart());
    if (above >= 0 && h.send_up_cnt > 0)
      NSFEM_NCCL(ncclSend(vec + h.send_up_off * width, (size_t)(h.send_up_cnt * width), ncclDouble,
                          above, comm, s));
    if (below >= 0 && h.send_down_cnt > 0)
      NSFEM_NCCL(ncclSend(vec + h.send_down_off * width, (size_t)(h.send_down_cnt * width),
                          ncclDouble, below, comm, s));
    if (below >= 0 && h.recv_below_cnt > 0)
      NSFEM_NCCL(ncclRecv(vec + h.recv_below_off * width, (size_t)(h.recv_below_cnt * width),
                          ncclDouble, below, comm, s));
    if (above >= 0 && h.recv_above_cnt > 0)
      NSFEM_NCCL(ncclRecv(vec + h.recv_above_off * width, (size_t)(h.recv_above_cnt * width),
                          ncclDouble, above, comm, s));
    NSFEM_NCCL(ncclGroupEnd());
  }
  void exchange_add(hipStream_t s, const HaloRange& h, double* vec, int width) override {
    count_exchange_add(h, width);
    const int above = up(), below = down();
    NSFEM_REQUIRE(!(periodic && size == 1), "a periodic partition needs at least two RCCL ranks");
    const size_t n_up = above >= 0 ? (size_t)(h.send_up_cnt * width) : 0;      // what the rank above holds for me
    const size_t n_dn = below >= 0 ? (size_t)(h.send_down_cnt * width) : 0;
    if (stage.n < n_up + n_dn) {
      NSFEM_HIP(hipStreamSynchronize(s));
      stage.alloc(n_up + n_dn);
    }
    // same ordering rule as in `exchange`: sends up-then-down, receives below-then-above
    NSFEM_NCCL(ncclGroupStart());
    if (above >= 0 && h.recv_above_cnt > 0)
      NSFEM_NCCL(ncclSend(vec + h.recv_above_off * width, (size_t)(h.recv_above_cnt * width),
                          ncclDouble, above, comm, s));
    if (below >= 0 && h.recv_below_cnt > 0)
      NSFEM_NCCL(ncclSend(vec + h.recv_below_off * width, (size_t)(h.recv_below_cnt * width),
                          ncclDouble, below, comm, s));
    if (n_dn > 0) NSFEM_NCCL(ncclRecv(stage.p + n_up, n_dn, ncclDouble, below, comm, s));
    if (n_up > 0) NSFEM_NCCL(ncclRecv(stage.p, n_up, ncclDouble, above, comm, s));
    NSFEM_NCCL(ncclGroupEnd());
    add_range(s, (int64_t)n_up, stage.p, vec + h.send_up_off * width);
    add_range(s, (int64_t)n_dn, stage.p + n_up, vec + h.send_down_off * width);
  }
};

Comm* make_local_comm(void* group, int rank) {
  LocalGroup* g = static_cast<LocalGroup*>(group);
  NSFEM_REQUIRE(g && rank >= 0 && rank < g->size, "bad local communicator rank");
  LocalComm* c = new LocalComm();
  c->g = g;
  c->rank = rank;
  c->size = g->size;
  return c;
}

Comm* make_rccl_comm(const char* id128, int rank, int size) {
  NSFEM_REQUIRE(id128 && rank >= 0 && rank < size, "bad RCCL communicator arguments");
  static_assert(sizeof(ncclUniqueId) <= 128, "ncclUniqueId larger than the ABI buffer");
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  RcclComm* c = new RcclComm();
  c->rank = rank;
  c->size = size;
  ncclResult_t r = ncclCommInitRank(&c->comm, size, id, rank);
  if (r != ncclSuccess) {
    delete c;
    throw Error(NSFEM_ERR_COMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  }
  return c;
}

}  // namespace nsfem

using namespace nsfem;

extern "C" int nsfem_comm_local_create(int size, void** group) {
  if (!group || size < 1 || size > kMaxLocalRanks) return NSFEM_ERR_ARG;
  LocalGroup* g = new LocalGroup();
  g->size = size;
  *group = g;
  return NSFEM_OK;
}

extern "C" void nsfem_comm_local_destroy(void* group) { delete static_cast<LocalGroup*>(group); }

extern "C" int nsfem_comm_unique_id(char* id128) {
  if (!id128) return NSFEM_ERR_ARG;
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return NSFEM_ERR_COMM;
  std::memset(id128, 0, 128);
  std::memcpy(id128, &id, sizeof(id));
  return NSFEM_OK;
}
