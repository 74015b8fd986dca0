// Sparsity pattern, slot map and inverted index of a finite-element operator ON THE DEVICE (round 4; the host version
// is pattern.cpp::build_pattern, which this reproduces array for array).  What dolfin's SparsityPatternBuilder /
// TensorLayout do inside every dlfn.*VariationalSolver of the reference (source/ns_ipcs_solver.py:136-147).
//
// Every (cell, i, j) contributes the pair (row = rowmap[cell][i], col = colmap[cell][j]).  One stable radix sort of
// the 64-bit keys row << 32 | col with the source number q = cell (nr nc) + i nc + j as payload gives everything:
//   * the distinct keys in sorted order are the CSR pattern (rows ascending, columns ascending inside a row);
//   * the rank of a key among the distinct keys is the CSR slot of its sources          -> slot map (SoA [nr nc][cells]);
//   * the sorted payload IS the inverted index (per slot its sources, ascending because the sort is stable and the
//     payload starts ascending), the first position of every distinct key its pointer  -> cptr, cidx.
// A 13 M-dof tetrahedral mesh has 3.1e8 candidate entries in its P2 x P2 pattern: 1.3 s on 16 host threads plus
// 0.8 s of uploads (slot map and index are 2.5 GB) against ~0.2 s here, nothing crosses PCIe but the CSR pattern the
// host keeps for row blocks, SELL slices and stencil dictionaries.
#include "nsfem_internal.hpp"
#include <hipcub/hipcub.hpp>

namespace nsfem {

// maps are SoA on the device: rowmap[i * n_cells + cell]
__global__ __launch_bounds__(256) void k_pat_keys(int64_t n, int n_cells, int nr, int nc, const int32_t* __restrict__ rowmap,
                                                  const int32_t* __restrict__ colmap, uint64_t* __restrict__ keys,
                                                  uint32_t* __restrict__ vals) {
  const int loc = nr * nc;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cell = q / loc;
    const int ij = (int)(q - cell * loc), i = ij / nc, j = ij - i * nc;
    const uint32_t r = (uint32_t)rowmap[(size_t)i * n_cells + cell], c = (uint32_t)colmap[(size_t)j * n_cells + cell];
    keys[q] = ((uint64_t)r << 32) | c;
    vals[q] = (uint32_t)q;
  }
}

__global__ __launch_bounds__(256) void k_pat_heads(int64_t n, const uint64_t* __restrict__ keys, int32_t* __restrict__ head) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
    head[k] = (k == 0 || keys[k] != keys[k - 1]) ? 1 : 0;
}

// rank[k] = inclusive sum of the head flags: slot of the sorted entry k is rank[k] - 1
__global__ __launch_bounds__(256) void k_pat_fill(int64_t n, int n_rows, int n_cells, int loc, const uint64_t* __restrict__ keys,
                                                  const uint32_t* __restrict__ vals, const int32_t* __restrict__ rank,
                                                  int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                                  int32_t* __restrict__ diag, int32_t* __restrict__ slot,
                                                  int32_t* __restrict__ cptr) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t key = keys[k];
    const int32_t sl = rank[k] - 1;
    const uint32_t q = vals[k];
    const uint32_t cell = q / (uint32_t)loc, ij = q - cell * (uint32_t)loc;
    slot[(size_t)ij * n_cells + cell] = sl;
    const bool head = k == 0 || key != keys[k - 1];
    const int row = (int)(key >> 32);
    if (head) {
      col[sl] = (int32_t)(uint32_t)key;
      cptr[sl] = (int32_t)k;
      if (diag && (uint32_t)row == (uint32_t)key) diag[row] = sl;
      const int prev = k == 0 ? -1 : (int)(keys[k - 1] >> 32);
      for (int r = prev + 1; r <= row; ++r) rowptr[r] = sl;          // (rows without entries point at the next one)
    }
    if (k == n - 1) {
      const int32_t nnz = rank[k];
      for (int r = row + 1; r <= n_rows; ++r) rowptr[r] = nnz;
      cptr[nnz] = (int32_t)n;
    }
  }
}

static int grid_for(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, 1 << 20); }

// d_rowmap / d_colmap: SoA device copies of the cell dof maps ([nr][n_cells], [nc][n_cells]).  Fills every device
// array of `out` build_pattern + upload_pattern would, and the host copies of rowptr / col.
void build_pattern_device(hipStream_t s, int n_rows, int n_cols, int n_cells, const int32_t* d_rowmap, int nr,
                          const int32_t* d_colmap, int nc, bool want_diag, Pattern& out) {
  const int loc = nr * nc;
  const int64_t n = (int64_t)n_cells * loc;
  if (n <= 0 || n > INT32_MAX) throw Error(NSFEM_ERR_ARG, "pattern: candidate entries exceed int32");
  out.n_rows = n_rows;
  out.n_cols = n_cols;
  out.nr = nr;
  out.nc = nc;
  DevBuf<uint64_t> k0, k1;
  DevBuf<uint32_t> v0, v1;
  k0.alloc((size_t)n);
  k1.alloc((size_t)n);
  v0.alloc((size_t)n);
  v1.alloc((size_t)n);
  hipLaunchKernelGGL(k_pat_keys, dim3(grid_for(n)), dim3(256), 0, s, n, n_cells, nr, nc, d_rowmap, d_colmap, k0.p, v0.p);
  NSFEM_HIP(hipGetLastError());
  int row_bits = 1;
  while (row_bits < 32 && ((int64_t)1 << row_bits) < (int64_t)n_rows) ++row_bits;
  size_t tmp_bytes = 0;
  NSFEM_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k0.p, k1.p, v0.p, v1.p, (int)n, 0, 32 + row_bits, s));
  DevBuf<char> tmp;
  tmp.alloc(tmp_bytes);
  NSFEM_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, k0.p, k1.p, v0.p, v1.p, (int)n, 0, 32 + row_bits, s));
  // head flags and their inclusive sum (the buffers of the unsorted keys are free again: reuse)
  int32_t* head = reinterpret_cast<int32_t*>(k0.p);
  int32_t* rank = reinterpret_cast<int32_t*>(v0.p);
  hipLaunchKernelGGL(k_pat_heads, dim3(grid_for(n)), dim3(256), 0, s, n, (const uint64_t*)k1.p, head);
  NSFEM_HIP(hipGetLastError());
  size_t scan_bytes = 0;
  NSFEM_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, head, rank, (int)n, s));
  if (scan_bytes > tmp_bytes) { tmp.alloc(scan_bytes); tmp_bytes = scan_bytes; }
  NSFEM_HIP(hipcub::DeviceScan::InclusiveSum(tmp.p, scan_bytes, head, rank, (int)n, s));
  int32_t nnz = 0;
  NSFEM_HIP(hipMemcpyAsync(&nnz, rank + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, s));
  NSFEM_HIP(hipStreamSynchronize(s));
  out.nnz = nnz;
  out.rowptr.alloc((size_t)n_rows + 1);
  out.col.alloc((size_t)nnz);
  out.slot.alloc((size_t)n);
  out.cptr.alloc((size_t)nnz + 1);
  if (want_diag) {
    out.diag.alloc((size_t)n_rows);
    NSFEM_HIP(hipMemsetAsync(out.diag.p, 0xff, sizeof(int32_t) * (size_t)n_rows, s));      // -1
  }
  hipLaunchKernelGGL(k_pat_fill, dim3(grid_for(n)), dim3(256), 0, s, n, n_rows, n_cells, loc, (const uint64_t*)k1.p,
                     (const uint32_t*)v1.p, (const int32_t*)rank, out.rowptr.p, out.col.p,
                     want_diag ? out.diag.p : (int32_t*)nullptr, out.slot.p, out.cptr.p);
  NSFEM_HIP(hipGetLastError());
  // the sorted payload is the inverted index (uint32 source numbers < 2^31: the same bits as int32)
  out.cidx.alloc((size_t)n);
  NSFEM_HIP(hipMemcpyAsync(out.cidx.p, v1.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, s));
  out.h_rowptr.resize((size_t)n_rows + 1);
  out.h_col.resize((size_t)nnz);
  NSFEM_HIP(hipMemcpyAsync(out.h_rowptr.data(), out.rowptr.p, sizeof(int32_t) * ((size_t)n_rows + 1), hipMemcpyDeviceToHost, s));
  NSFEM_HIP(hipMemcpyAsync(out.h_col.data(), out.col.p, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost, s));
  NSFEM_HIP(hipStreamSynchronize(s));
  build_rowblocks(out, s);
}

// Node-sorted layout of the element VECTORS (MeshDev::nptr / ndst): entry (cell, i) goes to position ndst[i][cell], the
// contributions of node n are the run nptr[n] ... nptr[n + 1], ascending in the source number cell nl + i (what
// build_inverse_index + the transposition loop of nsfem_create produced on the host).  One stable 32-bit radix sort.
__global__ __launch_bounds__(256) void k_node_keys(int64_t n, int n_cells, int nl, const int32_t* __restrict__ map,
                                                   uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cell = q / nl;
    const int i = (int)(q - cell * nl);
    keys[q] = (uint32_t)map[(size_t)i * n_cells + cell];
    vals[q] = (uint32_t)q;
  }
}
__global__ __launch_bounds__(256) void k_node_fill(int64_t n, int n_nodes, int n_cells, int nl, const uint32_t* __restrict__ keys,
                                                   const uint32_t* __restrict__ vals, int32_t* __restrict__ nptr,
                                                   int32_t* __restrict__ ndst) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t q = vals[k], cell = q / (uint32_t)nl, i = q - cell * (uint32_t)nl;
    ndst[(size_t)i * n_cells + cell] = (int32_t)k;
    const int node = (int)keys[k];
    const int prev = k == 0 ? -1 : (int)keys[k - 1];
    for (int r = prev + 1; r <= node; ++r) nptr[r] = (int32_t)k;
    if (k == n - 1)
      for (int r = node + 1; r <= n_nodes; ++r) nptr[r] = (int32_t)n;
  }
}
void build_node_index_device(hipStream_t s, int n_nodes, int n_cells, int nl, const int32_t* d_map, DevBuf<int32_t>& nptr,
                             DevBuf<int32_t>& ndst) {
  const int64_t n = (int64_t)n_cells * nl;
  if (n <= 0 || n > INT32_MAX) throw Error(NSFEM_ERR_ARG, "node index exceeds int32");
  DevBuf<uint32_t> k0, k1, v0, v1;
  k0.alloc((size_t)n); k1.alloc((size_t)n); v0.alloc((size_t)n); v1.alloc((size_t)n);
  hipLaunchKernelGGL(k_node_keys, dim3(grid_for(n)), dim3(256), 0, s, n, n_cells, nl, d_map, k0.p, v0.p);
  NSFEM_HIP(hipGetLastError());
  int bits = 1;
  while (bits < 32 && ((int64_t)1 << bits) < (int64_t)n_nodes) ++bits;
  size_t tmp_bytes = 0;
  NSFEM_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k0.p, k1.p, v0.p, v1.p, (int)n, 0, bits, s));
  DevBuf<char> tmp;
  tmp.alloc(tmp_bytes);
  NSFEM_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, k0.p, k1.p, v0.p, v1.p, (int)n, 0, bits, s));
  nptr.alloc((size_t)n_nodes + 1);
  ndst.alloc((size_t)n);
  hipLaunchKernelGGL(k_node_fill, dim3(grid_for(n)), dim3(256), 0, s, n, n_nodes, n_cells, nl, (const uint32_t*)k1.p,
                     (const uint32_t*)v1.p, nptr.p, ndst.p);
  NSFEM_HIP(hipGetLastError());
  NSFEM_HIP(hipStreamSynchronize(s));
}

}  // namespace nsfem
