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
// can be tested on a single-GPU box.  ShmComm does the same for one PROCESS per rank (host-staged
// through POSIX shared memory): the process-per-rank launch path on ranks that share a device.
#include "nsfem_internal.hpp"
#include <rccl/rccl.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cerrno>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

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
  int kind[kMaxLocalRanks] = {0};     // which collective every rank entered (checked after the barrier)
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

// list halos: dst[didx[i]] (op)= src[sidx[i]] for `width` interleaved values per node; a null index
// array means the identity (packed staging buffers)
__global__ __launch_bounds__(256) void k_move_idx(int64_t n, int width, const double* __restrict__ src,
                                                  const int32_t* __restrict__ sidx,
                                                  double* __restrict__ dst,
                                                  const int32_t* __restrict__ didx, int add) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * width;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / width;
    const int c = (int)(t % width);
    const int64_t si = (sidx ? (int64_t)sidx[i] : i) * width + c, di = (didx ? (int64_t)didx[i] : i) * width + c;
    if (add) dst[di] += src[si];
    else dst[di] = src[si];
  }
}
static void move_idx(hipStream_t s, int64_t n, int width, const double* src, const int32_t* sidx,
                     double* dst, const int32_t* didx, bool add) {
  if (n <= 0) return;
  const int grid = (int)std::min<int64_t>((n * width + 255) / 256, 1024);
  hipLaunchKernelGGL(k_move_idx, dim3(grid), dim3(256), 0, s, n, width, src, sidx, dst, didx, add ? 1 : 0);
  NSFEM_HIP(hipGetLastError());
}

struct LocalComm : Comm {
  LocalGroup* g = nullptr;
  DevBuf<double> scratch;
  // the ranks are threads that meet at a barrier: a rank-local branch around a collective would
  // pair an all-reduce with a halo exchange silently -- compare what everybody entered
  void check_same_collective(int mine) {
    for (int r = 0; r < size; ++r)
      if (g->kind[r] != mine)
        throw Error(NSFEM_ERR_COMM, "collective mismatch: rank " + std::to_string(rank) + " entered kind " +
                                        std::to_string(mine) + ", rank " + std::to_string(r) + " kind " +
                                        std::to_string(g->kind[r]) + " (1/2 all-reduce sum/max, 3 exchange, 4 reverse add)");
  }
  void reduce(hipStream_t s, double* dev, int64_t count, int op) {
    count_allreduce(count);
    NSFEM_HIP(hipStreamSynchronize(s));
    g->ptr[rank] = dev;
    g->kind[rank] = 1 + op;
    g->barrier();
    check_same_collective(1 + op);
    if (scratch.n < (size_t)count) scratch.alloc((size_t)count);
    PtrPack pk;
    for (int r = 0; r < kMaxLocalRanks; ++r) pk.p[r] = r < size ? g->ptr[r] : nullptr;
    int grid = (int)std::min<int64_t>((count + 255) / 256, 1024);
    hipLaunchKernelGGL(k_reduce_ranks, dim3(grid), dim3(256), 0, s, count, size, pk, op, scratch.p);
    NSFEM_HIP(hipGetLastError());
    NSFEM_HIP(hipStreamSynchronize(s));
    g->barrier();      // everybody has read everybody's input
    NSFEM_HIP(hipMemcpyAsync(dev, scratch.p, sizeof(double) * count, hipMemcpyDeviceToDevice, s));
    static const bool trace = std::getenv("NSFEM_COMM_TRACE") != nullptr;
    if (trace) {
      std::vector<double> h((size_t)count);
      NSFEM_HIP(hipMemcpyAsync(h.data(), scratch.p, sizeof(double) * count, hipMemcpyDeviceToHost, s));
      NSFEM_HIP(hipStreamSynchronize(s));
      double sum = 0.0;
      for (double v : h) sum += v;
      fprintf(stderr, "[rank %d] allreduce #%lld count %lld op %d sum %.17g\n", rank, (long long)n_allreduce,
              (long long)count, op, sum);
    }
  }
  void allreduce_sum(hipStream_t s, double* dev, int64_t count) override { reduce(s, dev, count, 0); }
  void allreduce_max(hipStream_t s, double* dev, int64_t count) override { reduce(s, dev, count, 1); }
  void exchange(hipStream_t s, const HaloRange& h, double* vec, int width) override {
    count_exchange(h, width);
    static const bool trace = std::getenv("NSFEM_COMM_TRACE") != nullptr;
    if (trace)
      fprintf(stderr, "[rank %d] exchange #%lld width %d lists %lld/%lld\n", rank, (long long)n_exchange, width,
              h.lists ? (long long)h.lists->n_send() : -1LL, h.lists ? (long long)h.lists->n_recv() : -1LL);
    NSFEM_HIP(hipStreamSynchronize(s));
    g->ptr[rank] = vec;
    g->halo[rank] = h;
    g->kind[rank] = 3;
    g->barrier();
    check_same_collective(3);
    if (h.lists) {                 // pull every neighbour's send list into my ghost entries
      const HaloLists& me = *h.lists;
      for (size_t k = 0; k < me.nbr.size(); ++k) {
        const int q = me.nbr[k];
        const HaloLists* o = g->halo[q].lists;
        NSFEM_REQUIRE(o, "neighbour without list halo");
        const int m = o->find(rank);
        const int64_t cnt = me.recv_ptr[k + 1] - me.recv_ptr[k];
        if (!(m >= 0 && o->send_ptr[m + 1] - o->send_ptr[m] == cnt))
          throw Error(NSFEM_ERR_ARG, "list halo size mismatch: rank " + std::to_string(rank) + " expects " +
                                         std::to_string(cnt) + " from rank " + std::to_string(q) + ", which sends " +
                                         std::to_string(m >= 0 ? o->send_ptr[m + 1] - o->send_ptr[m] : -1) +
                                         " (width " + std::to_string(width) + ", my lists " +
                                         std::to_string(me.n_send()) + "/" + std::to_string(me.n_recv()) + ", its " +
                                         std::to_string(o->n_send()) + "/" + std::to_string(o->n_recv()) + ")");
        move_idx(s, cnt, width, g->ptr[q], o->send_idx.p + o->send_ptr[m], vec,
                 me.recv_idx.p + me.recv_ptr[k], false);
      }
      NSFEM_HIP(hipStreamSynchronize(s));
      g->barrier();
      return;
    }
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
    g->kind[rank] = 4;
    g->barrier();
    check_same_collective(4);
    if (h.lists) {                 // add every neighbour's ghost copies of my nodes (one launch per neighbour:
      const HaloLists& me = *h.lists;          //  a node may be a ghost on several ranks)
      for (size_t k = 0; k < me.nbr.size(); ++k) {
        const int q = me.nbr[k];
        const HaloLists* o = g->halo[q].lists;
        NSFEM_REQUIRE(o, "neighbour without list halo");
        const int m = o->find(rank);
        const int64_t cnt = me.send_ptr[k + 1] - me.send_ptr[k];
        NSFEM_REQUIRE(m >= 0 && o->recv_ptr[m + 1] - o->recv_ptr[m] == cnt, "list halo size mismatch (add)");
        move_idx(s, cnt, width, g->ptr[q], o->recv_idx.p + o->recv_ptr[m], vec,
                 me.send_idx.p + me.send_ptr[k], true);
      }
      NSFEM_HIP(hipStreamSynchronize(s));
      g->barrier();
      return;
    }
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
  DevBuf<double> pack_out, pack_in;   // list halos: packed send / receive buffers
  void ensure_stage(hipStream_t s, DevBuf<double>& b, size_t n) {
    if (b.n >= n) return;
    NSFEM_HIP(hipStreamSynchronize(s));
    if (cs) NSFEM_HIP(hipStreamSynchronize(cs));
    b.alloc(n + n / 4);
  }
  // forward = true: owners -> ghosts; false: ghosts -> owners (added there)
  void exchange_lists(hipStream_t s, const HaloLists& L, double* vec, int width, bool forward) {
    const std::vector<int64_t>& out_ptr = forward ? L.send_ptr : L.recv_ptr;
    const std::vector<int64_t>& in_ptr = forward ? L.recv_ptr : L.send_ptr;
    const int32_t* out_idx = forward ? L.send_idx.p : L.recv_idx.p;
    const int32_t* in_idx = forward ? L.recv_idx.p : L.send_idx.p;
    const int64_t n_out = out_ptr.back(), n_in = in_ptr.back();
    ensure_stage(s, pack_out, (size_t)(n_out * width));
    ensure_stage(s, pack_in, (size_t)(n_in * width));
    move_idx(s, n_out, width, vec, out_idx, pack_out.p, nullptr, false);
    NSFEM_NCCL(ncclGroupStart());
    for (size_t k = 0; k < L.nbr.size(); ++k) {
      const int64_t so = out_ptr[k] * width, sc = (out_ptr[k + 1] - out_ptr[k]) * width;
      const int64_t ro = in_ptr[k] * width, rc = (in_ptr[k + 1] - in_ptr[k]) * width;
      if (sc > 0) NSFEM_NCCL(ncclSend(pack_out.p + so, (size_t)sc, ncclDouble, L.nbr[k], comm, s));
      if (rc > 0) NSFEM_NCCL(ncclRecv(pack_in.p + ro, (size_t)rc, ncclDouble, L.nbr[k], comm, s));
    }
    NSFEM_NCCL(ncclGroupEnd());
    if (forward) {
      move_idx(s, n_in, width, pack_in.p, nullptr, vec, in_idx, false);
    } else {
      for (size_t k = 0; k < L.nbr.size(); ++k)      // per neighbour: a node may be sent to several ranks
        move_idx(s, in_ptr[k + 1] - in_ptr[k], width, pack_in.p + in_ptr[k] * width, nullptr, vec,
                 in_idx + in_ptr[k], true);
    }
  }
  void exchange(hipStream_t s, const HaloRange& h, double* vec, int width) override {
    count_exchange(h, width);
    if (h.lists) { exchange_lists(s, *h.lists, vec, width, true); return; }
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
    if (h.lists) { exchange_lists(s, *h.lists, vec, width, false); return; }
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

// -------------------------------------------------------------------- ShmComm
// One PROCESS per rank, host-staged through a POSIX shared-memory segment: the transport for ranks
// that share one device (RCCL refuses two ranks on one GPU) -- it lets the process-per-rank launch
// path of bench.py (spawner, rendezvous, partitions, every collective) run end to end on a one-GPU
// box.  Not a performance path: every collective is device -> host -> device with two barriers.
// Reductions add the ranks' slots in rank order on every rank: deterministic and identical everywhere.
struct ShmHeader {
  std::atomic<int> arrived;
  std::atomic<unsigned> generation;
  std::atomic<int> attached;
  int size;
  int64_t slot_bytes;
  std::atomic<int> aborted;      // a rank that throws inside a collective sets it: its peers leave their barriers
};
struct ShmSlotHead {        // start of every rank's slot
  int64_t kind;             // collective entered (mismatch check, as LocalComm)
  int64_t n_part;           // number of parts in the slot
  int64_t peer[64], off[64], cnt[64];   // part k is meant for rank peer[k]: doubles [off, off + cnt)
};

struct ShmComm : Comm {
  std::string name;
  void* base = nullptr;
  size_t total = 0;
  ShmHeader* hd = nullptr;
  int64_t slot_bytes = 0;
  DevBuf<double> dstage;
  std::vector<double> hstage;
  ShmSlotHead* head(int r) const {
    return reinterpret_cast<ShmSlotHead*>(static_cast<char*>(base) + 4096 + (size_t)r * (size_t)slot_bytes);
  }
  double* data(int r) const { return reinterpret_cast<double*>(reinterpret_cast<char*>(head(r)) + sizeof(ShmSlotHead)); }
  int64_t capacity() const { return (slot_bytes - (int64_t)sizeof(ShmSlotHead)) / 8; }
  void barrier() {
    const unsigned g = hd->generation.load(std::memory_order_acquire);
    if (hd->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == size) {
      hd->arrived.store(0, std::memory_order_relaxed);
      hd->generation.fetch_add(1, std::memory_order_acq_rel);
    } else {
      int spins = 0;
      const auto t0 = std::chrono::steady_clock::now();
      while (hd->generation.load(std::memory_order_acquire) == g) {
        if (hd->aborted.load(std::memory_order_acquire))
          throw Error(NSFEM_ERR_COMM, "shared-memory communicator: another rank failed inside a collective");
        if (++spins > 2000) {
          std::this_thread::sleep_for(std::chrono::microseconds(50));
          if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(300))
            throw Error(NSFEM_ERR_COMM, "shared-memory communicator: a rank did not reach the barrier within 300 s");
        }
      }
    }
  }
  void check_kind(int64_t mine) {
    for (int r = 0; r < size; ++r)
      if (head(r)->kind != mine) {
        hd->aborted.store(1, std::memory_order_release);       // (the peers would wait for this rank at the next barrier)
        throw Error(NSFEM_ERR_COMM, "collective mismatch: rank " + std::to_string(rank) + " entered kind " +
                                        std::to_string(mine) + ", rank " + std::to_string(r) + " kind " +
                                        std::to_string(head(r)->kind));
      }
  }
  ~ShmComm() override {
    if (base) {
      const bool last = hd->attached.fetch_sub(1) == 1;
      munmap(base, total);
      if (last || rank == 0) shm_unlink(name.c_str());
    }
  }
  void reduce(hipStream_t s, double* dev, int64_t count, int op) {
    count_allreduce(count);
    if (count > capacity()) {
      hd->aborted.store(1, std::memory_order_release);
      throw Error(NSFEM_ERR_COMM, "all-reduce larger than the shared-memory slot");
    }
    ShmSlotHead* me = head(rank);
    me->kind = 1 + op;
    NSFEM_HIP(hipMemcpyAsync(data(rank), dev, sizeof(double) * count, hipMemcpyDeviceToHost, s));
    NSFEM_HIP(hipStreamSynchronize(s));
    barrier();
    check_kind(1 + op);
    hstage.resize((size_t)count);
    const double* d0 = data(0);
    for (int64_t i = 0; i < count; ++i) hstage[i] = d0[i];
    for (int r = 1; r < size; ++r) {
      const double* d = data(r);
      if (op) for (int64_t i = 0; i < count; ++i) hstage[i] = std::fmax(hstage[i], d[i]);
      else for (int64_t i = 0; i < count; ++i) hstage[i] += d[i];
    }
    barrier();      // everybody has read everybody's slot
    NSFEM_HIP(hipMemcpyAsync(dev, hstage.data(), sizeof(double) * count, hipMemcpyHostToDevice, s));
    NSFEM_HIP(hipStreamSynchronize(s));
  }
  void allreduce_sum(hipStream_t s, double* dev, int64_t count) override { reduce(s, dev, count, 0); }
  void allreduce_max(hipStream_t s, double* dev, int64_t count) override { reduce(s, dev, count, 1); }

  // one (peer, device range) per outgoing and per incoming part; `add`: incoming parts are added
  struct Part { int peer; double* dev; int64_t cnt; const int32_t* idx; int64_t n_idx; };
  void ensure_dstage(hipStream_t s, size_t n) {
    if (dstage.n >= n) return;
    NSFEM_HIP(hipStreamSynchronize(s));
    dstage.alloc(n + n / 4 + 64);
  }
  // out[k]: what this rank hands to out[k].peer; in[k]: what it takes from in[k].peer -- the m-th part
  // a rank addresses to a peer pairs with the m-th part that peer expects from it (issue order, as RCCL)
  void swap_parts(hipStream_t s, int64_t kind, const std::vector<Part>& out, const std::vector<Part>& in,
                  int width, bool add) {
    ShmSlotHead* me = head(rank);
    NSFEM_REQUIRE(out.size() <= 64, "too many neighbours for the shared-memory communicator");
    me->kind = kind;
    me->n_part = (int64_t)out.size();
    int64_t o = 0, need = 0;
    for (const Part& p : out) need += p.cnt;
    if (need > capacity()) {
      hd->aborted.store(1, std::memory_order_release);
      throw Error(NSFEM_ERR_COMM, "halo larger than the shared-memory slot");
    }
    ensure_dstage(s, (size_t)need);
    for (size_t k = 0; k < out.size(); ++k) {
      const Part& p = out[k];
      me->peer[k] = p.peer; me->off[k] = o; me->cnt[k] = p.cnt;
      if (p.cnt > 0) {
        if (p.idx) {
          move_idx(s, p.n_idx, width, p.dev, p.idx, dstage.p + o, nullptr, false);
          NSFEM_HIP(hipMemcpyAsync(data(rank) + o, dstage.p + o, sizeof(double) * p.cnt, hipMemcpyDeviceToHost, s));
        } else {
          NSFEM_HIP(hipMemcpyAsync(data(rank) + o, p.dev, sizeof(double) * p.cnt, hipMemcpyDeviceToHost, s));
        }
      }
      o += p.cnt;
    }
    NSFEM_HIP(hipStreamSynchronize(s));
    barrier();
    check_kind(kind);
    int64_t need_in = 0;
    for (const Part& p : in) need_in += p.cnt;
    if (add || std::any_of(in.begin(), in.end(), [](const Part& p) { return p.idx != nullptr; }))
      ensure_dstage(s, (size_t)need_in);
    std::vector<int> taken(size, 0);
    int64_t so = 0;
    for (const Part& p : in) {
      const ShmSlotHead* h = head(p.peer);
      int seen = 0, found = -1;
      for (int64_t k = 0; k < h->n_part; ++k)
        if (h->peer[k] == rank && seen++ == taken[p.peer]) { found = (int)k; break; }
      ++taken[p.peer];
      if (found < 0 || h->cnt[found] != p.cnt)
        throw Error(NSFEM_ERR_ARG, "halo size mismatch: rank " + std::to_string(rank) + " expects " +
                                       std::to_string(p.cnt) + " doubles from rank " + std::to_string(p.peer) +
                                       ", which offers " + std::to_string(found < 0 ? -1 : h->cnt[found]));
      if (p.cnt > 0) {
        const double* src = data(p.peer) + h->off[found];
        if (!add && !p.idx) {
          NSFEM_HIP(hipMemcpyAsync(p.dev, src, sizeof(double) * p.cnt, hipMemcpyHostToDevice, s));
        } else {
          NSFEM_HIP(hipMemcpyAsync(dstage.p + so, src, sizeof(double) * p.cnt, hipMemcpyHostToDevice, s));
          if (p.idx) move_idx(s, p.n_idx, width, dstage.p + so, nullptr, p.dev, p.idx, add);
          else add_range(s, p.cnt, dstage.p + so, p.dev);
        }
      }
      so += p.cnt;
    }
    NSFEM_HIP(hipStreamSynchronize(s));
    barrier();      // the slots may be overwritten
  }
  void exchange(hipStream_t s, const HaloRange& h, double* vec, int width) override {
    count_exchange(h, width);
    std::vector<Part> out, in;
    if (h.lists) {
      const HaloLists& L = *h.lists;
      for (size_t k = 0; k < L.nbr.size(); ++k) {
        const int64_t ns = L.send_ptr[k + 1] - L.send_ptr[k], nr = L.recv_ptr[k + 1] - L.recv_ptr[k];
        out.push_back({L.nbr[k], vec, ns * width, L.send_idx.p + L.send_ptr[k], ns});
        in.push_back({L.nbr[k], vec, nr * width, L.recv_idx.p + L.recv_ptr[k], nr});
      }
      for (Part& p : out) if (p.n_idx == 0) p.idx = nullptr;
      for (Part& p : in) if (p.n_idx == 0) p.idx = nullptr;
      swap_parts(s, 3, out, in, width, false);
      return;
    }
    const int above = up(), below = down();
    NSFEM_REQUIRE(!(periodic && size == 1), "a periodic partition needs at least two ranks");
    // same order as RcclComm::exchange: sends up, down; receives from below, from above
    if (above >= 0 && h.send_up_cnt > 0) out.push_back({above, vec + h.send_up_off * width, h.send_up_cnt * width, nullptr, 0});
    if (below >= 0 && h.send_down_cnt > 0) out.push_back({below, vec + h.send_down_off * width, h.send_down_cnt * width, nullptr, 0});
    if (below >= 0 && h.recv_below_cnt > 0) in.push_back({below, vec + h.recv_below_off * width, h.recv_below_cnt * width, nullptr, 0});
    if (above >= 0 && h.recv_above_cnt > 0) in.push_back({above, vec + h.recv_above_off * width, h.recv_above_cnt * width, nullptr, 0});
    swap_parts(s, 3, out, in, width, false);
  }
  void exchange_add(hipStream_t s, const HaloRange& h, double* vec, int width) override {
    count_exchange_add(h, width);
    std::vector<Part> out, in;
    if (h.lists) {
      const HaloLists& L = *h.lists;
      for (size_t k = 0; k < L.nbr.size(); ++k) {
        const int64_t ns = L.send_ptr[k + 1] - L.send_ptr[k], nr = L.recv_ptr[k + 1] - L.recv_ptr[k];
        out.push_back({L.nbr[k], vec, nr * width, nr ? L.recv_idx.p + L.recv_ptr[k] : nullptr, nr});
        in.push_back({L.nbr[k], vec, ns * width, ns ? L.send_idx.p + L.send_ptr[k] : nullptr, ns});
      }
      swap_parts(s, 4, out, in, width, true);
      return;
    }
    const int above = up(), below = down();
    NSFEM_REQUIRE(!(periodic && size == 1), "a periodic partition needs at least two ranks");
    // ghost ranges go back to their owners (RcclComm::exchange_add's order)
    if (above >= 0 && h.recv_above_cnt > 0) out.push_back({above, vec + h.recv_above_off * width, h.recv_above_cnt * width, nullptr, 0});
    if (below >= 0 && h.recv_below_cnt > 0) out.push_back({below, vec + h.recv_below_off * width, h.recv_below_cnt * width, nullptr, 0});
    if (below >= 0 && h.send_down_cnt > 0) in.push_back({below, vec + h.send_down_off * width, h.send_down_cnt * width, nullptr, 0});
    if (above >= 0 && h.send_up_cnt > 0) in.push_back({above, vec + h.send_up_off * width, h.send_up_cnt * width, nullptr, 0});
    swap_parts(s, 4, out, in, width, true);
  }
};

Comm* make_shm_comm(const char* name, int rank, int size, int64_t slot_bytes) {
  NSFEM_REQUIRE(name && name[0] == '/' && rank >= 0 && rank < size && size <= 64, "bad shared-memory communicator arguments");
  if (slot_bytes <= 0) slot_bytes = 64ll << 20;
  slot_bytes = (slot_bytes + 4095) / 4096 * 4096;
  const size_t total = 4096 + (size_t)size * (size_t)slot_bytes;
  int fd = -1;
  if (rank == 0) {
    shm_unlink(name);
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) throw Error(NSFEM_ERR_COMM, std::string("shm_open(create) failed: ") + std::strerror(errno));
    if (ftruncate(fd, (off_t)total) != 0) {
      close(fd);
      shm_unlink(name);
      throw Error(NSFEM_ERR_COMM, std::string("ftruncate failed: ") + std::strerror(errno));
    }
  } else {
    // rank 0 creates the segment; the caller's rendezvous (bench.py: the broadcast of the name) does not
    // order that creation before the other ranks' open, so wait for it to appear at its full size
    for (int tries = 0; tries < 6000; ++tries) {
      fd = shm_open(name, O_RDWR, 0600);
      if (fd >= 0) {
        struct stat st;
        if (fstat(fd, &st) == 0 && (size_t)st.st_size >= total) break;
        close(fd);
        fd = -1;
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    if (fd < 0) throw Error(NSFEM_ERR_COMM, "shared-memory segment of rank 0 did not appear within 60 s");
  }
  void* base = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (base == MAP_FAILED) throw Error(NSFEM_ERR_COMM, std::string("mmap failed: ") + std::strerror(errno));
  ShmComm* c = new ShmComm();
  c->name = name;
  c->base = base;
  c->total = total;
  c->hd = static_cast<ShmHeader*>(base);
  c->rank = rank;
  c->size = size;
  c->slot_bytes = slot_bytes;
  if (rank == 0) {           // fresh segments are zero-filled: arrived = generation = attached = 0
    c->hd->size = size;
    c->hd->slot_bytes = slot_bytes;
  }
  c->hd->attached.fetch_add(1);
  // first barrier = everybody is attached (and rank 0's header is visible)
  const auto t0 = std::chrono::steady_clock::now();
  while (c->hd->attached.load() < size) {
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
      delete c;
      throw Error(NSFEM_ERR_COMM, "shared-memory communicator: not all ranks attached within 120 s");
    }
  }
  // every rank checks the geometry rank 0 wrote against its own arguments: a mismatch would make the slots overlap
  if (c->hd->size != size || c->hd->slot_bytes != slot_bytes) {
    const std::string msg = "shared-memory communicator: rank " + std::to_string(rank) + " was given size " +
                            std::to_string(size) + " / slot " + std::to_string((long long)slot_bytes) +
                            " B, rank 0 created size " + std::to_string(c->hd->size) + " / slot " +
                            std::to_string((long long)c->hd->slot_bytes) + " B";
    delete c;
    throw Error(NSFEM_ERR_COMM, msg);
  }
  return c;
}

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
