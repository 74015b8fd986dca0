// Internal declarations of libnsfem_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <stdexcept>
#include <functional>
#include <map>
#include <memory>
#include "../../include/nsfem.h"

// Knock-out switches of the measurement experiments (NSFEM_LATTICE_DBG, NSFEM_JL_DBG, NSFEM_SPMV_DEBUG: parts of the
// dominant kernels switched off, WRONG results): compiled in only with -DNSFEM_KNOCKOUTS=1 (scripts/build_knockouts.sh);
// in the product build NSFEM_KO(..) is the constant 0 and the environment variables are never read.
#ifndef NSFEM_KNOCKOUTS
#define NSFEM_KNOCKOUTS 0
#endif
#if NSFEM_KNOCKOUTS
#define NSFEM_KO(expr) (expr)
#else
#define NSFEM_KO(expr) (0)
#endif

namespace nsfem {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define NSFEM_HIP(expr)                                                              \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess)                                                            \
      throw ::nsfem::Error(NSFEM_ERR_HIP, std::string(#expr) + ": " +                \
                                              hipGetErrorString(e_));                \
  } while (0)

#define NSFEM_REQUIRE(cond, msg)                                                     \
  do {                                                                               \
    if (!(cond)) throw ::nsfem::Error(NSFEM_ERR_ARG, std::string(msg));              \
  } while (0)

// number of partial sums every reduction kernel emits (= its grid size); the
// consumer kernels re-reduce them in a fixed order => bitwise reproducible dots.
constexpr int kParts = 512;
constexpr int kPartSlots = 14;   // partial-sum slots of the Krylov work space (slots 10, 11: drivers; 12, 13: start sums |r0|^2, |b|^2)
constexpr int kBlock = 256;

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    release();
    n = count;
    // (+64 bytes of slack: the SpMV kernels read 16-byte aligned quads that may end past the last
    // entry of a column / value array)
    if (count) NSFEM_HIP(hipMalloc(&p, count * sizeof(T) + 64));
  }
  void upload(const T* host, size_t count, hipStream_t s) {
    if (count != n) alloc(count);
    if (count) NSFEM_HIP(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void upload(const std::vector<T>& v, hipStream_t s) {
    upload(v.data(), v.size(), s);
    NSFEM_HIP(hipStreamSynchronize(s));   // host vector may die after the call
  }
  void zero(hipStream_t s) {
    if (n) NSFEM_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
  }
};

// ---- host-built sparsity pattern (CSR over block rows/cols) + slot map ---------
struct HostPattern {
  int n_rows = 0, n_cols = 0, nr = 0, nc = 0;   // nr/nc: local rows/cols per cell
  std::vector<int32_t> rowptr, col, diag;       // diag only for square patterns
  std::vector<int32_t> slot;                    // SoA: [nr*nc][n_cells]
  std::vector<int32_t> cptr, cidx;              // inverted index of the slot map (want_contrib)
};
int host_threads();                             // threads of the host set-up loops (NSFEM_HOST_THREADS)
void build_pattern(int n_rows, int n_cols, int n_cells, const int32_t* rowmap, int nr,
                   const int32_t* colmap, int nc, bool want_diag, HostPattern& out, bool want_contrib = false);

struct Pattern {
  int n_rows = 0, n_cols = 0, nnz = 0, nr = 0, nc = 0;
  DevBuf<int32_t> rowptr, col, diag, slot;
  DevBuf<int32_t> cptr, cidx;             // per slot: sources (cell * nr*nc + i * nc + j)
  DevBuf<int32_t> rblk;                   // chunks of the CSR-stream SpMV: n_rblk records {r0, r1, s0, s1}
  DevBuf<int32_t> rblk1;                  // the same chunks as a plain row-start table (v1 kernel)
  int n_rblk = 0;
  int max_chunk_rows = 0;                 // rows of the longest chunk (LDS row-pointer table)
  std::vector<int32_t> h_rblk;            // host copy (interior / halo-adjacent split)
  // partitioned meshes: the row blocks [int_b0, int_b1) reference no ghost column -- they can run
  // while the halo exchange of the input vector is still in flight (mark_interior_blocks)
  int int_b0 = 0, int_b1 = 0;
  // SELL-64 layout (sliced ELLPACK, slices of 64 rows = one wavefront, entries of a slice stored
  // column-major: lane r of a wave walks row r) for patterns whose consecutive rows have (nearly)
  // equal length -- structured numberings that group the P2 nodes by lattice-parity class.  The
  // k-th entries of 64 consecutive rows then sit at consecutive column ids: the x gather of a wave
  // is one contiguous run instead of ~50 scattered cache lines.  Built by build_sell when the
  // padding stays below a few per cent; n_slices = 0 otherwise (CSR-stream kernel).
  DevBuf<int32_t> sell_ptr;               // [n_slices + 1] entry offsets (multiples of 64)
  DevBuf<int32_t> sell_col;               // column ids, padding entries point at the row itself
  DevBuf<int32_t> sell_src;               // CSR slot of every entry, -1 = padding
  int n_slices = 0;
  int64_t sell_len = 0;
  int sell_w0 = 0, sell_w1 = 0;           // interior workgroups (4 slices each), see int_b0/int_b1
  int wg_w0 = 0, wg_w1 = 0;   // longest run of 256-row groups without a ghost column (dictionary kernel)
  // contiguous workgroup ranges of the 8 XCDs, balanced by (padded) nonzeros: the parity-class
  // numberings group rows of very different length (3D vertex rows 65, edge rows 14-32 entries),
  // an equal-count split would leave one XCD with 2-3 times the work of the others
  int sell_xcd[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  std::vector<int32_t> h_rowptr, h_col;   // kept for export
};
void build_rowblocks(Pattern& p, hipStream_t s);
// the same pattern, slot map and inverted index built on the device (pattern_device.hip); d_rowmap / d_colmap: SoA
// device copies of the cell dof maps [nr][n_cells] / [nc][n_cells]
void build_pattern_device(hipStream_t s, int n_rows, int n_cols, int n_cells, const int32_t* d_rowmap, int nr,
                          const int32_t* d_colmap, int nc, bool want_diag, Pattern& out);
// node-sorted layout of the element vectors (MeshDev::nptr, ndst) from the SoA device dof map [nl][n_cells]
void build_node_index_device(hipStream_t s, int n_nodes, int n_cells, int nl, const int32_t* d_map, DevBuf<int32_t>& nptr,
                             DevBuf<int32_t>& ndst);
void build_sell(Pattern& p, hipStream_t s);
// longest run of row blocks none of whose columns is flagged in `ghost_cols` ([n_cols] flags)
void mark_interior_blocks(Pattern& p, const std::vector<uint8_t>& ghost_cols);
void build_inverse_index(int n_targets, int64_t n_sources,
                         const std::function<int32_t(int64_t)>& target_of,
                         std::vector<int32_t>& ptr, std::vector<int32_t>& idx);

// block matrix on a pattern; vals[nnz][BR][BC] row-major blocks
// Stencil dictionary of a scalar matrix family on one pattern (structured meshes): rows whose
// (column offsets, values) agree to 2^-40 (of the largest entry) share one dictionary entry, so a product reads ONE
// 4-byte id per row instead of 12 bytes per nonzero -- on the lattice meshes of the benchmark
// configurations a P2 operator has a few dozen (2D) to a few thousand (3D, slab-blocked parity
// numbering) distinct rows.  Built once per pattern from the constant matrices it will combine
// (mass and stiffness: any a M + b K has the same dictionary); the value table of a matrix is a
// gather from the CSR values of the representative rows (dict_update, device only).  The
// compressed copy differs from the CSR matrix by the dedupe tolerance: unless `exact` it is used by
// multigrid SMOOTHING and Newton-Jacobian products only, never by a residual or a linear operator
// whose solution is returned.
struct StencilDict {
  int n_rows = 0, n_stencils = 0, lmax = 0;
  bool exact = false;      // every row equals its representative BITWISE (meshes with a binary spacing): the
                           // dictionary copy is the matrix itself and every product may use it
  bool tables_only = false; // small lattice levels (< 1024 rows): the tables exist for the fused multigrid legs
                           // (mglegs.hip) only -- products and single-level smoothing keep the CSR kernels
  int bsz = 1;             // doubles per entry (block matrices: br * bc)
  bool rect = false;       // offsets are relative to the row's FIRST column (rectangular P2 x P1 blocks:
                           // the two numberings differ), kept per row in cbase; square: relative to the row
  int max_local = 0;       // most entries any workgroup uses (sizes the LDS copy)
  DevBuf<int32_t> cbase;   // [n_rows] (rect only)
  DevBuf<int32_t> sid;     // [n_rows] dictionary entry of every row
  DevBuf<uint8_t> lid;     // [n_rows] its position in the list of the row's workgroup (256 rows)
  DevBuf<int32_t> wg_ptr, wg_list;   // per workgroup: the entries it uses (<= 32)
  DevBuf<int32_t> len;     // [n_stencils]
  DevBuf<int32_t> off;     // [n_stencils * lmax] column - row
  DevBuf<int32_t> src;     // [n_stencils * lmax] CSR position in the representative row (-1: padding)
  DevBuf<int32_t> dpos;    // [n_stencils] position of the diagonal entry inside the stencil (-1: none); the
                           // smoother kernels take 1 / diagonal from the table instead of streaming a dinv vector
  // 2D lattice structure (lexicographic numbering, row = j * lat_w + i): every offset is dj * lat_w + di
  // with |di|, |dj| <= lat_r and every row's entries stay inside the lat_w x lat_h box.  Set (lat_w > 0)
  // only for square scalar operators with <= 64 stencils: the multi-step lattice smoother
  // (k_cheb_lattice) keeps a tile of x in LDS and needs (dj, di) to address it.
  int lat_w = 0, lat_h = 0, lat_r = 0;
  DevBuf<int32_t> pack;    // [n_stencils * lmax] (dj + 8) * 32 + (di + 8)
  DevBuf<uint8_t> sid8;    // [n_rows] dictionary entry of every row as one byte
  std::vector<int32_t> h_pack, h_len;     // host copies (offset tables of the lattice kernel's LDS layouts)
  struct LatticeOffsets {
    int ewh = 0, ehh = 0;
    DevBuf<int32_t> buf;   // [n_stencils][4 classes][lmax]
  };
  mutable std::vector<LatticeOffsets> loff_cache;   // one table per tile shape in use (built on first use)
  mutable DevBuf<uint8_t> sidm_scratch;             // entry | mask bytes for callers that bring none (test hook, timing)
  // entries with the canonical interior stencil shape of a parity class (launch_cheb_lattice: compile-time LDS
  // offsets): shape 0 = not looked at yet, -1 = none, 1 = P2 right-diagonal lattice, 2 = P1 7-point
  mutable int fixed_shape = 0;
  mutable unsigned long long fixed_mask[4] = {0, 0, 0, 0};
};
// false: the rows do not repeat (unstructured mesh) -- no dictionary
// min_rows > 0: patterns down to that many rows are accepted when they turn out to be 2D lattices (tables_only)
void ensure_fixed_masks(const StencilDict& d);   // fills d.fixed_shape / fixed_mask (entries of canonical interior shape)
bool build_stencil_dict(hipStream_t s, const Pattern& p, const double* dev_a, const double* dev_b,
                        StencilDict& d, int bsz = 1, bool rect = false, int min_rows = 0);

struct BlockMat {
  const Pattern* pat = nullptr;
  int br = 1, bc = 1;
  DevBuf<double> vals;
  const StencilDict* dict = nullptr;      // set: sell_update() also refreshes dict_vals
  DevBuf<double> dict_vals;               // [n_stencils * lmax]
  DevBuf<double> dict_dinv;               // [n_stencils] 1 / diagonal (scalar square operators)
  DevBuf<double> lat_vals;                // [n_stencils * lp] zero-padded rows for the lattice kernel
  bool dict_ready = false;
  // copy of the values in the pattern's SELL-64 order (scalar matrices on patterns that have one);
  // refreshed by sell_update() after every change of `vals` -- the SpMV kernels use it only while
  // sell_ready is set
  DevBuf<double> sell_vals;
  bool sell_ready = false;
  void init(const Pattern* p, int br_, int bc_, hipStream_t s) {
    pat = p; br = br_; bc = bc_;
    vals.alloc((size_t)p->nnz * br * bc);
    vals.zero(s);
    sell_ready = false;
  }
  void sell_update(hipStream_t s);        // linalg.hip
};

// reference-element tables (7-point degree-5 rule)
struct QuadTables {
  double w[7];
  double phi2[7][6];
  double dphi2[7][6][2];
  double phi1[7][3];
  // dphi1 is constant: (-1,-1), (1,0), (0,1)
  // CFL diagnostic: P2 basis at the 6 points of the degree-4 Strang-Fix rule and its inverse
  // (the local L2 projection onto P2 with that rule is interpolation at its points)
  double cfl_phi[6][6];
  double cfl_inv[6][6];
};
void fill_quad_tables(QuadTables& t);
void upload_quad_tables(const QuadTables& t);
struct MeshDev;
void launch_cfl(hipStream_t s, const MeshDev& m, const double* u, double scale, double* parts,
                int n_parts);

// SpMV row-mask modes
enum MaskMode { MASK_NONE = 0, MASK_IDENTITY = 1, MASK_ZERO = 2 };

// -------------------------- kernel launch wrappers ------------------------------
// y = A x.  nv = number of interleaved right-hand sides the scalar blocks act on
// (scalar P2 matrices applied to both velocity components use br=bc=1, nv=2).
// ghost: treatment of rows flagged 2 (ghost rows of a partitioned mesh): 0 output 0, 2 computed
// phase (partitioned meshes, halo exchange overlapped with the product): 0 all rows, 1 only the
// row blocks that touch no ghost column (Pattern::int_b0..int_b1), 2 the remaining row blocks
// dict_ok: the product may run on the matrix's stencil-dictionary copy (equal to the CSR values to
// 2^-40 of the largest entry): Newton-Jacobian products and smoothing steps only
void launch_spmv(hipStream_t s, const BlockMat& A, int nv, const double* x, double* y,
                 const uint8_t* rowmask, int maskmode, int ghost = 0, int phase = 0, int dict_ok = 0);
bool launch_spmv_with_gather(hipStream_t s, const BlockMat& A, int nv, const double* x, double* y,
                             const uint8_t* rowmask, int maskmode, const int32_t* gptr, const double* gbuf);
void launch_spmv_cheb_first(hipStream_t s, const BlockMat& A, int nv, const double* x, double* y,
                            const uint8_t* rowmask, const double* dinv, double c2, double* d,
                            double* x1);
// y = b - A x  (same arguments + b)
void launch_residual(hipStream_t s, const BlockMat& A, int nv, const double* x,
                     const double* b, double* y, const uint8_t* rowmask, int maskmode, int phase = 0);

// y += A x (masked rows -> 0);  Chebyshev/Jacobi smoother step
//   d = c1 d + c2 dinv (b - A x) ; xout = x + d   (masked rows -> 0)
void launch_spmv_accumulate(hipStream_t s, const BlockMat& A, int nv, const double* x, double* y,
                            const uint8_t* rowmask, int ghost = 0 /* 2: ghost rows accumulate too */);
// y = scale * A x ;  y += scale * A x on rows not flagged in skipmask (flagged rows untouched)
void launch_spmv_scaled(hipStream_t s, const BlockMat& A, int nv, double scale, const double* x,
                        double* y, int dict_ok = 0);
void launch_spmv_axpy(hipStream_t s, const BlockMat& A, int nv, double scale, const double* x,
                      double* y, const uint8_t* skipmask, int dict_ok = 0);
void launch_cheb_step(hipStream_t s, const BlockMat& A, int nv, const double* x, const double* b,
                      const double* dinv, double* d, double c1, double c2, double* xout,
                      const uint8_t* rowmask, int ghost = 0 /* 1: ghost rows keep x */, int phase = 0,
                      int ident = 0 /* 1: rows flagged 1 take xout = b (identity rows) */);

// multi-step lattice smoother (2D lexicographic lattices with a stencil dictionary; linalg.hip)
void refresh_env_switches();            // NSFEM_LATTICE, NSFEM_LATTICE_TRANSFERS (re-read by nsfem_create)
void refresh_assembly_switches();       // NSFEM_JAC_LATTICE
bool partitioned_lattice_kernels();      // NSFEM_PARTITIONED_LATTICE (default on): k_jac_lattice / k_cheb_lattice on strips
void refresh_leg_switches();            // NSFEM_MG_LEGS, NSFEM_LEG_GROUP, NSFEM_LEG_T (mglegs.hip)
bool lattice_transfers_enabled();
bool lattice_smoother_available(const BlockMat& A, int nv);
bool lattice_tables_available(const BlockMat& A, int nv);   // dictionary tables of a 2D lattice operator are in place
int lattice_smoother_max_steps(const BlockMat& A, bool from_zero, bool with_resid);
void launch_cheb_lattice(hipStream_t s, const BlockMat& A, int nv, const double* x_in, const double* b,
                         const double* d_in, double* x_out, double* d_out, double* r_out,
                         const uint8_t* mask, int steps, const double* c1, const double* c2, int ident,
                         const uint8_t* sidm = nullptr, const double* xc = nullptr, const double* rf = nullptr,
                         double* b_out = nullptr, int gh_lo = 0, int gh_hi = 0, int gh_zero = 0);
// out[row] = dictionary entry | (mask of component c) << (6 + c): one byte per row for the lattice kernel
void launch_lattice_sidm(hipStream_t s, const BlockMat& A, int nv, const uint8_t* mask, uint8_t* out);
// out = R rf on a lattice hierarchy (rows flagged in the coarse mask: 0); false = shapes do not nest, nothing launched
bool launch_restrict_lattice(hipStream_t s, int nv, int Wc, int Hc, int Wf, int Hf, const double* rf,
                             const uint8_t* mask, double* out);
// two levels of a restriction chain in one launch: b1 = R1 rf, b2 = R2 b1
bool launch_restrict_lattice2(hipStream_t s, int nv, int W2, int H2, int W1, int H1, int Wf, int Hf,
                              const double* rf, const uint8_t* mask1, const uint8_t* mask2, double* b1, double* b2);

int64_t lattice_launch_bytes(const BlockMat& A, int nv, bool from_zero, bool d_in, bool d_out, bool r_out);

// element kernels
// 2D lattice meshes in rectangle_mesh numbering (cell 2 (sy nx + sx) + t, P2 node j W + i): the node positions of
// a cell are a template of its type -- verified cell by cell on the host (build_cell_lattice), used by k_jac_lattice
struct CellLattice {
  bool ok = false, tried = false;
  int nx = 0, ny = 0, W = 0, H = 0;
  int di[2][6] = {{0}}, dj[2][6] = {{0}};   // lattice offset of local node k of cell type t from the square's corner
  int rank[2][6] = {{0}};                   // position of the cell among the cells around that node (ascending)
  // uniform lattices (every cell of a type has the geometry of the first one BIT FOR BIT -- checked on the device
  // with the kernels' own load_geo): {J^-1, |det|} of the two cell types; the one-launch kernel then takes the
  // geometry from 10 scalar loads instead of six strided coordinate loads per cell (48 B per cell of the launch)
  bool geo_uniform = false;
  DevBuf<double> ugeo;                      // [2][5]
};
bool build_cell_lattice(const int32_t* p2map /* [cell][6] */, int nc, int W, int H, CellLattice& cl);
struct MeshDev;
void check_uniform_geometry(hipStream_t s, MeshDev& m);   // fills m.cl.geo_uniform / ugeo (after build_cell_lattice)

struct MeshDev {
  int dim = 2;              // 2: triangles (6 + 3 nodes per cell), 3: tetrahedra (10 + 4)
  int n_cells = 0, n_p2 = 0, n_p1 = 0, n_vertices = 0;
  DevBuf<double> vx;        // SoA vertex coords per cell: [6][n_cells] (x0,y0,x1,y1,x2,y2)
  DevBuf<int32_t> p2;       // SoA [6][n_cells]
  DevBuf<int32_t> p1;       // SoA [3][n_cells]
  // vector assembly without atomics: the element vectors are stored NODE-SORTED -- entry (cell, i)
  // goes to position ndst[i][cell] (SoA [6 | 10][n_cells]) of rbuf, the contributions of node n are
  // the contiguous run nptr[n] .. nptr[n + 1] (ascending cell order: deterministic sums, and the
  // gather kernel streams the buffer instead of chasing 16 / 24-byte pieces through it)
  DevBuf<int32_t> nptr, ndst;
  DevBuf<double> ebuf;      // element Jacobian blocks [cell][6][6][4] (plain stores)
  DevBuf<double> rbuf;      // element residual [cell][6][2]
  CellLattice cl;
};
void launch_assemble_p2_scalar(hipStream_t s, const MeshDev& m, const Pattern& p22,
                               double* mass, double* stiff);
// tetrahedral meshes (assembly3d.hip); the launch_* wrappers dispatch on MeshDev::dim
struct QuadTables3 {          // 15-point degree-5 Keast rule, P2 (10 nodes) / P1 (4 nodes) tables
  double w[15];
  double phi2[15][10];
  double dphi2[15][10][3];
  double phi1[15][4];
};
void fill_quad_tables_3d(QuadTables3& t);
void upload_quad_tables_3d();
void launch_cfl_3d(hipStream_t s, const MeshDev& m, const double* u, double scale, double* parts,
                   int n_parts);
// extreme eigenvalues of diag(M_e)^{-1} M_e for the P2 element mass matrix (Wathen: they bound
// the spectrum of the Jacobi-scaled assembled mass matrix on any affine mesh)
void p2_mass_jacobi_bounds(int dim, double& lmin, double& lmax);
// max_i K_ii / M_ii of two scalar matrices on the same pattern (stored diagonals)
double diag_ratio_max(hipStream_t s, const Pattern& pat, const double* M, const double* K, double* parts);
void assemble_p2_scalar_3d(hipStream_t s, const MeshDev& m, const Pattern& p22, double* mass,
                           double* stiff);
void assemble_p1_scalar_3d(hipStream_t s, const MeshDev& m, const Pattern& p11, double* stiff,
                           double* mass);
void assemble_div_grad_3d(hipStream_t s, const MeshDev& m, const Pattern& p12, const Pattern& p21,
                          double* div, double* grad, double* divT);
void jacobian_init_3d(hipStream_t s, int nnz, const double* L, const double* E, double cvE,
                      double* J);
void assemble_viscous_extra_3d(hipStream_t s, const MeshDev& m, const Pattern& p22, double* extra);
void convection_jacobian_3d(hipStream_t s, const MeshDev& m, const Pattern& p22, const double* u,
                            double cc, const double* L, const double* E, double cvE, double* J,
                            int form, bool picard);
void convection_residual_3d(hipStream_t s, const MeshDev& m, const double* u, double cc, double* b,
                            int form);
void convection_action_3d(hipStream_t s, const MeshDev& m, const double* u, const double* v,
                          double cc, double* y, int form, bool picard, const uint8_t* skipmask = nullptr);
void launch_assemble_p1_scalar(hipStream_t s, const MeshDev& m, const Pattern& p11,
                               double* stiff, double* mass);
void launch_assemble_div_grad(hipStream_t s, const MeshDev& m, const Pattern& p12,
                              const Pattern& p21, double* div, double* grad, double* divT);
void launch_assemble_viscous_extra(hipStream_t s, const MeshDev& m, const Pattern& p22,
                                   double* extra);
// J(2x2 blocks) = L (scalar) (x) I_2 [+ cv_extra * E]  then  += cc * conv'(u)
void launch_jacobian_init(hipStream_t s, int nnz, const double* L, const double* E,
                          double cvE, double* J);
// J = L (x) I_2 + cvE * E + cc * conv'(u): element blocks are stored, then gathered per slot
void launch_convection_jacobian(hipStream_t s, const MeshDev& m, const Pattern& p22,
                                const double* u, double cc, const double* L, const double* E,
                                double cvE, double* J, int form, bool picard);
// skipmask != nullptr: vector entries with a nonzero flag are left untouched by the node gather
void launch_convection_action(hipStream_t s, const MeshDev& m, const double* u, const double* v,
                              double cc, double* y, int form, bool picard,
                              const uint8_t* skipmask = nullptr);
// only the element kernel of the action: the element vectors land node-sorted in m.rbuf (runs m.nptr); the
// caller sums them per node (launch_spmv_with_gather)
bool launch_residual_lattice(hipStream_t s, const MeshDev& m, const BlockMat& L, const double* u, const double* g,
                             double cc, int form, double* y);   // y = L u + g + c_c conv(u) (exact dictionaries)
// y = L x + c_c [d conv(u)/du] x, identity on the rows flagged in mask: ONE launch on 2D lattice meshes (m.cl.ok and a
// lattice stencil dictionary of L); false = not available, nothing launched
bool jacobian_lattice_available(const MeshDev& m, const BlockMat& L);
int64_t jacobian_lattice_bytes(const MeshDev& m);   // u, x read, y written, ids + masks, vertex coordinates
bool launch_jacobian_lattice(hipStream_t s, const MeshDev& m, const BlockMat& L, const double* u, const double* x,
                             double cc, int form, bool picard, const uint8_t* mask, double* y, int phase = 0,
                             int gh_lo = 0, int gh_hi = 0);
// partitioned strips: can the launch be cut into tile rows that read no ghost line (phase 1, under the halo
// exchange) and the rest (phase 2)?
bool jacobian_lattice_split(const MeshDev& m, int gh_lo, int gh_hi);
// are the ghost nodes of a W x H lattice whole lines at its bottom (lo of them) and top (hi)?
bool ghost_lattice_lines(const std::vector<uint8_t>& ghost, int W, int H, int& lo, int& hi);
void launch_convection_cells(hipStream_t s, const MeshDev& m, const double* u, const double* v, double cc,
                             int form, bool picard);
void launch_convection_residual(hipStream_t s, const MeshDev& m, const double* u, double cc,
                                double* b, int form);
// per-facet surface force / flux / measure (boundary.hip): out[nf][dim + 2]
void launch_boundary_force(hipStream_t s, const MeshDev& m, int nf, const int32_t* fcell,
                           const int32_t* flocal, const double* u, const double* p, double nu,
                           double sym, double* out);
// diag extraction: d[(i,a)] = 1 / A_ii[a][a]  (mask rows -> 1)
void launch_inv_diag(hipStream_t s, const BlockMat& A, int nv, const uint8_t* rowmask,
                     double* dinv);

// vector kernels (n = number of doubles)
void launch_rot90(hipStream_t s, int64_t n_nodes, double g, const double* u, double* out);
void launch_jac_add_skew(hipStream_t s, int64_t nnz, double g, const double* M, double* J);
void launch_rot_field(hipStream_t s, const MeshDev& m, double* out);
void launch_coord_field_3d(hipStream_t s, const MeshDev& m, double* out);
void launch_cross3(hipStream_t s, int64_t n_nodes, const double g[3], const double* u, double* out);
void launch_jac_add_skew3(hipStream_t s, int64_t nnz, const double g[3], const double* M, double* J);
void launch_axpby(hipStream_t s, int64_t n, double a, const double* x, double b, const double* y,
                  double* z);                                     // z = a x + b y
void launch_lincomb3(hipStream_t s, int64_t n, double a, const double* x, double b,
                     const double* y, double c, const double* z, double* out);
void launch_scale_combine(hipStream_t s, int nnz, double a, const double* A, double b,
                          const double* B, double* C);            // C = a A + b B (values)
void launch_dot(hipStream_t s, int64_t n, const double* x, const double* y, double* parts);
void launch_sum_sub_mean(hipStream_t s, int64_t n, double* x, double* parts);
void launch_correction_setup(hipStream_t s, int64_t n, double a, double* rhs, const double* t, const uint8_t* mask,
                             double* r0, double* parts_r, double* parts_b);
// the same together with |b|^2 (512 partial sums) in one launch; rows with mask != 0 must be exactly the nbc dofs
void launch_bc_residual_norm(hipStream_t s, int64_t n, double* b, const uint8_t* mask, int nbc, const int32_t* dofs,
                             const double* g, const double* x, double* parts);
void launch_set_bc_residual(hipStream_t s, int nbc, const int32_t* dofs, const double* g,
                            const double* x, double* b);          // b[d] = x[d] - g[d]
void launch_set_values(hipStream_t s, int nbc, const int32_t* dofs, const double* g, double* x);
void launch_fill_mask(hipStream_t s, int nbc, const int32_t* dofs, uint8_t* mask);
void launch_mask_zero(hipStream_t s, int64_t n, const uint8_t* mask, double* x);
void launch_add_scalar(hipStream_t s, int64_t n, double a, double* x);

// Krylov scratch + drivers
struct LinOp;
// identity of one captured Krylov iteration body (all baked kernel arguments)
struct GraphKey {
  const void *op, *prec, *x, *dinv;
  int64_t n;
  int parity;
  uint64_t epoch;
  bool operator==(const GraphKey& o) const {
    return op == o.op && prec == o.prec && x == o.x && dinv == o.dinv && n == o.n &&
           parity == o.parity && epoch == o.epoch;
  }
};
struct KrylovWork {
  int64_t n = 0;
  DevBuf<double> r, rhat, p, v, s, t, phat, shat, z, q;
  DevBuf<double> parts;     // [kPartSlots][kParts]
  DevBuf<double> scal;      // device scalars
  double* h_parts = nullptr;   // pinned host [kPartSlots][kParts]
  // reduced sums for the host (convergence checks): a one-workgroup kernel re-reduces the partial sums of up to four
  // slots in the order the device kernels use (sum_parts) and stores them, then a sequence number, into coherent
  // pinned host memory; the host spins on the number.  Replaces hipMemcpyAsync + hipStreamSynchronize: on the
  // MI355X the copy path leaves the GPU idle ~20 us before and ~25 us after the copy (rocprofv3 trace, round 4)
  struct HostResult { double v[4]; uint64_t seq; };
  HostResult* h_res = nullptr;
  uint64_t seq_no = 0;
  // captured HIP graphs of the iteration bodies
  std::vector<std::pair<GraphKey, hipGraphExec_t>> graphs;
  std::vector<GraphKey> seen;   // bodies that ran eagerly once (lazy table builds synchronise): captured at their second run
  uint64_t epoch = 0;
  // replay is worth its capture only while the baked arguments stay put: contexts whose operators change every few
  // iterations (variable time steps) fall back to eager launches (graphs_off); profiling windows that record HIP
  // events inside the bodies suspend it
  uint64_t touch = 0;           // bumped by every solver that uses the work vectors (who may rely on what another left in them)
  double last_target = 0.0;     // absolute residual target of the last solve (max(atol, rtol |b|)): the drivers' iteration predictor
  bool graphs_off = false, graphs_suspended = false;
  int64_t replays_in_epoch = 0;
  int short_epochs = 0;
  bool graphs_enabled(const LinOp& op) const;
  void replay(hipStream_t s, const GraphKey& key, const std::function<void()>& body);
  void clear_graphs();
  void ensure(int64_t n_);
  ~KrylovWork();
};

// general preconditioner z = M^{-1} r (multigrid); nullptr in LinOp => Jacobi (dinv)
struct Precond {
  virtual ~Precond() {}
  virtual void apply(hipStream_t s, const double* r, double* z) = 0;
};

// ---- partitioned meshes: halo ranges (node units) and the communicator interface ---------
// general halo of an unstructured partition: per neighbouring rank the nodes this rank sends
// (owned here, ghost there) and receives (ghost here), both in the order of the global node ids
struct HaloLists {
  std::vector<int32_t> nbr;
  std::vector<int64_t> send_ptr, recv_ptr;       // [nbr.size() + 1]
  DevBuf<int32_t> send_idx, recv_idx;
  int64_t n_send() const { return send_ptr.empty() ? 0 : send_ptr.back(); }
  int64_t n_recv() const { return recv_ptr.empty() ? 0 : recv_ptr.back(); }
  int find(int rank) const {
    for (size_t k = 0; k < nbr.size(); ++k)
      if (nbr[k] == rank) return (int)k;
    return -1;
  }
};
// strips / slabs: two neighbours, contiguous ranges (no packing); `lists` set: index lists instead
struct HaloRange {
  int64_t send_up_off = 0, send_up_cnt = 0, recv_above_off = 0, recv_above_cnt = 0;
  int64_t send_down_off = 0, send_down_cnt = 0, recv_below_off = 0, recv_below_cnt = 0;
  const HaloLists* lists = nullptr;
};
struct Comm {
  int rank = 0, size = 1;
  bool periodic = false;     // the partition closes periodically: neighbours wrap around
  int up() const { return rank + 1 < size ? rank + 1 : (periodic ? 0 : -1); }
  int down() const { return rank > 0 ? rank - 1 : (periodic ? size - 1 : -1); }
  virtual ~Comm() {}
  // in-place sum over the ranks of `count` doubles in device memory, ordered on stream s
  virtual void allreduce_sum(hipStream_t s, double* dev, int64_t count) = 0;
  virtual void allreduce_max(hipStream_t s, double* dev, int64_t count) = 0;
  // fill the ghost ranges of `vec` (width entries per node) from the neighbouring ranks
  virtual void exchange(hipStream_t s, const HaloRange& h, double* vec, int width) = 0;
  // the reverse of `exchange`: the values this rank holds in its ghost ranges are ADDED to the
  // owners' entries (their send ranges) -- products with operators that are stored as rank-local
  // additive parts (A = sum_r A_r), where every rank computes partial sums for its ghost rows
  virtual void exchange_add(hipStream_t s, const HaloRange& h, double* vec, int width) = 0;
  // ---- overlap of the exchange with the interior rows of the product that consumes `vec`:
  // exchange_begin orders the exchange after everything queued on `s` so far and runs it on the
  // communicator's own stream; exchange_end makes `s` wait for its completion.  Between the two
  // calls the caller launches (on `s`) only work that neither reads the ghost ranges nor writes
  // the send ranges of `vec`.
  bool overlap = false;
  hipStream_t cs = nullptr;
  hipEvent_t ev_in = nullptr, ev_done = nullptr;
  int64_t n_overlapped = 0;
  void exchange_begin(hipStream_t s, const HaloRange& h, double* vec, int width) {
    if (!cs) {
      NSFEM_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
      NSFEM_HIP(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
      NSFEM_HIP(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
    }
    NSFEM_HIP(hipEventRecord(ev_in, s));
    NSFEM_HIP(hipStreamWaitEvent(cs, ev_in, 0));
    exchange(cs, h, vec, width);
    NSFEM_HIP(hipEventRecord(ev_done, cs));
    ++n_overlapped;
  }
  void exchange_end(hipStream_t s) { NSFEM_HIP(hipStreamWaitEvent(s, ev_done, 0)); }
  void release_streams() {
    if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); cs = nullptr; }
    if (ev_in) { (void)hipEventDestroy(ev_in); ev_in = nullptr; }
    if (ev_done) { (void)hipEventDestroy(ev_done); ev_done = nullptr; }
  }
  // traffic counters (calls / payload bytes this rank sends), read through nsfem_comm_stats
  int64_t n_allreduce = 0, n_exchange = 0, bytes_allreduce = 0, bytes_exchange = 0;
  // NSFEM_COMM_HIST=1: calls by (kind, payload bytes) since the last reset, printed by nsfem_comm_stats --
  // the payload size identifies the level / vector a message belongs to
  std::map<std::pair<int, int64_t>, int64_t> hist;
  void tally(int kind, int64_t bytes) {
    static const bool on = std::getenv("NSFEM_COMM_HIST") != nullptr;
    if (on) ++hist[{kind, bytes}];
  }
  void count_allreduce(int64_t count) { ++n_allreduce; bytes_allreduce += 8 * count; tally(0, 8 * count); }
  void count_exchange_add(const HaloRange& h, int width) {
    ++n_exchange;
    const int64_t before = bytes_exchange;
    if (h.lists) bytes_exchange += 8 * (int64_t)width * h.lists->n_recv();
    else bytes_exchange += 8 * (int64_t)width * ((up() >= 0 ? h.recv_above_cnt : 0) + (down() >= 0 ? h.recv_below_cnt : 0));
    tally(2, bytes_exchange - before);
  }
  void count_exchange(const HaloRange& h, int width) {
    ++n_exchange;
    const int64_t before = bytes_exchange;
    if (h.lists) bytes_exchange += 8 * (int64_t)width * h.lists->n_send();
    else bytes_exchange += 8 * (int64_t)width * ((up() >= 0 ? h.send_up_cnt : 0) + (down() >= 0 ? h.send_down_cnt : 0));
    tally(1, bytes_exchange - before);
  }
};
void launch_zero_ghost(hipStream_t s, int64_t n, const uint8_t* mask, double* x);  // x[mask==2]=0

// product with a halo-dependent input: exchange, then launch(0) -- or, when the communicator
// overlaps and the pattern has interior row blocks, exchange on the communicator's stream under
// launch(1) (interior blocks), then launch(2) (halo-adjacent blocks) after the event wait
template <class F>
inline void product_with_halo(Comm* comm, const HaloRange* halo, int width, hipStream_t s,
                              const double* v, const Pattern* pat, F&& launch) {
  if (!comm || !halo) { launch(0); return; }
  double* vec = const_cast<double*>(v);
  if (!comm->overlap || !pat || pat->n_rblk == 0 || pat->int_b1 <= pat->int_b0) {
    comm->exchange(s, *halo, vec, width);
    launch(0);
    return;
  }
  comm->exchange_begin(s, *halo, vec, width);
  launch(1);
  comm->exchange_end(s);
  launch(2);
}

// abstract operator (block systems): y = A x on vectors of length n
struct Operator {
  int64_t n = 0;
  virtual ~Operator() {}
  virtual void apply(hipStream_t s, const double* x, double* y) = 0;
};

struct LinOp {
  Operator* custom = nullptr;       // overrides A / nv / rowmask when set (BiCGStab only)
  const BlockMat* A = nullptr;
  int nv = 1;
  const uint8_t* rowmask = nullptr;
  int maskmode = MASK_NONE;
  const double* dinv = nullptr;     // Jacobi
  Precond* prec = nullptr;          // overrides dinv when set
  // partitioned meshes (all null / 0 in serial runs)
  Comm* comm = nullptr;
  const HaloRange* halo = nullptr;
  int halo_width = 1;               // vector entries per node
  const uint8_t* ghostmask = nullptr;
  int64_t n_global = 0;             // global length (mean projection)
  uint64_t graph_epoch = 0;         // bumped whenever baked kernel arguments may have changed
  bool x_zero = false;              // the caller's start vector is all zeros: the start residual is b itself
                                    // (BiCGStab skips the operator application that would compute b - A 0)
  double known_bnorm = -1.0;        // >= 0: |b|_2 (all ranks), already on the host -- with x_zero it is the start
                                    // residual too and the solve begins without a device -> host round trip
};

// ---- multigrid ------------------------------------------------------------------
// P (rows = finer level, cols = coarser level) and R = P^T as scalar CSR operators
struct Transfer {
  Pattern patP, patR;
  BlockMat P, R;
  std::vector<int32_t> h_inj;   // coarse node -> coinciding finer-level node
  std::vector<double> h_val;    // host copy of P's values (lattice check)
  int lattice = -1;             // P is the lattice interpolation of a (2 Wc - 1)-wide fine lattice: -1 unknown, 0 no, 1 yes
  bool is_lattice(int wf, int hf);
  // kind: 1 nested, 2 non-nested interpolation, 0 decide from the values (nsfem_mg_level_desc::transfer_kind)
  void build(hipStream_t s, int n_fine, int n_coarse, const int32_t* rowptr, const int32_t* col,
             const double* val, int kind = 0);
};

struct MGLevel {
  const BlockMat* A = nullptr;   // scalar operator of this level
  int n = 0;                     // scalar unknowns
  const uint8_t* mask = nullptr; // [n * nv]; level 0 uses the context's mask
  DevBuf<uint8_t> own_mask;
  DevBuf<double> dinv, xa, xb, x, b, r, d;
  DevBuf<double> xc, d2;         // lattice smoother: out-of-place iterate of a cycle leg, second direction buffer
  DevBuf<uint8_t> sidm;          // lattice smoother: dictionary entry | mask bits, one byte per row
  const void* sidm_for = nullptr; // (dictionary, mask) pair the byte array was built for
  const void* sidm_mask = nullptr;
  double lmax = 2.0;
  double ratio = 0.0;            // > 0: Chebyshev interval [lmax / ratio, lmax] of THIS level (truncated solve)
  const BlockMat* P = nullptr;   // transfer to / from the next coarser level
  const BlockMat* R = nullptr;
  Transfer* transfer = nullptr;  // the object P and R belong to (lattice check), when known
  const std::vector<int32_t>* h_inj = nullptr;
  // partitioned meshes
  const std::vector<uint8_t>* h_ghost = nullptr;   // per node: nonzero = ghost
  int ghost_lo = -2, ghost_hi = -2;                // ghost lattice lines at the bottom / top of a strip (-2: not looked
                                                   // at yet, -1: the ghost nodes are not whole lines)
  HaloRange halo;
  bool has_halo = false;
  // partitioned meshes, operators without an element-local form (the algebraic Schur Laplacian
  // D diag(M)^-1 D^T): A holds this rank's ADDITIVE part -- the sum over the ranks of the parts,
  // ghost rows included, is the operator.  Products: fill the ghosts of x, multiply every local
  // row, add the ghost rows at their owners (Comm::exchange_add).
  bool additive = false;
  DevBuf<double> t;              // additive levels: the product before the smoother update
};


// ---- fused multi-level launches on 2D lattice hierarchies (mglegs.hip) ---------------------------
// A "leg" is a short program of level operations (load, restrict, Chebyshev step, residual, prolong, store, dense
// coarse solve) that ONE launch of k_mg_leg runs from LDS: a workgroup owns a tile of the coarsest level of the leg
// (the anchor) and the matching nodes of the finer levels; every operation works on the tile widened by the halo the
// later operations need (planned backwards on the host), halo nodes are computed redundantly by the neighbours.
// Replaces the 7 - 13 us launches of the small multigrid levels (one per smoothing sequence, transfer and coarse
// solve) by one launch per cycle leg.
enum LegOpType { LEG_LOADB = 0, LEG_LOADX, LEG_RESTRICT_G, LEG_RESTRICT_L, LEG_CHEB0, LEG_CHEB, LEG_RESID,
                 LEG_PROLONG, LEG_STORE, LEG_DENSE };
enum LegFlags { LEGF_OLD_ZERO = 1, LEGF_IDENT = 2, LEGF_ADD = 4, LEGF_EXT_IN = 8, LEGF_EXT_OUT = 16, LEGF_SYNC = 32 };
constexpr int kLegMaxLevels = 8, kLegMaxOps = 96;
// builder record of one operation (host): level / array numbers, resolved by LegPlan::finalize
struct LegOpDev {
  int type, lev, dst, src, halo, flags, n_dense, pad_;
  double c1, c2;
  const double* gin;       // LOADB / LOADX / RESTRICT_G source, DENSE: the inverse [nv][n][n]
  double* gout;            // STORE / RESTRICT_L target (own nodes only)
};
// what the kernel reads per operation: everything resolved to plain numbers, one contiguous (scalar) load.
// LDS arrays: node (i, j) of a level sits at off + ((j - ly + g) * pitch + (i - lx + g)) * NV, (lx, ly) = first own
// node of the workgroup's tile on that level, g = halo + reach (zeroed guard ring)
struct LegOpRec {
  int type, flags, halo, shift;
  int W, H, np, n_dense;
  int d_off, d_pitch, d_g, s_off, s_pitch, s_g;       // destination / source array
  int b_off, b_pitch, b_g, e_off, e_pitch, e_g;       // right-hand side array; entry | mask bytes (offset in bytes)
  int t_off, o_shift, o_W, o_H;                       // value table; the other level of a transfer ...
  int o_off, o_pitch, o_g, R;                         // ... and its array
  double c1, c2;
  const double* gin;
  double* gout;
};
struct LegLevelDev {
  int W, H, shift, R;
  int n_st, lmax, lp, np;          // dictionary entries, longest row, value-table stride, (2 R + 1)^2
  int t_off, e_off, e_pitch, e_g;  // dense value table (offset in doubles), entry | mask bytes (offset in bytes)
  const uint8_t* sidm;
  const double* tval;              // [n_st][lp]
  const int32_t* pack;             // [n_st][lmax]  (dj + 8) * 32 + (di + 8)
  const int32_t* len;
  const double* dinv;
};
struct LegPlanDev {
  int n_levels, n_ops, T, ntx, nty, Wa, Ha, lds_doubles;
  LegLevelDev lv[kLegMaxLevels];
  LegOpRec op[kLegMaxOps];
};
struct MGLevel;
struct LegPlan {
  int nv = 1, threads = 1024;
  std::vector<MGLevel*> lv;        // finest first; the last one is the anchor (tiles are cut on it)
  std::vector<LegOpDev> ops;
  LegPlanDev h;                    // host copy of what the kernel reads
  DevBuf<LegPlanDev> dev;
  size_t lds_bytes = 0;
  double redundancy = 1.0;         // node operations of the launch / the same without halos
  int64_t launches = 0;
  // ops appended by the builders; `lev` indexes lv
  void add(int type, int lev, int dst, int src, int flags = 0, double c1 = 0.0, double c2 = 0.0,
           const double* gin = nullptr, double* gout = nullptr, int n_dense = 0);
  // plan the halos backwards, lay the LDS out for tiles of T anchor nodes; false: does not fit `lds_limit`
  bool finalize(hipStream_t s, int T, size_t lds_limit);
  void launch(hipStream_t s, const double* ext_in, double* ext_out);
};

struct Multigrid : Precond {
  int nv = 1;
  std::vector<MGLevel> lv;
  int degree = 2;                // Chebyshev steps per pre/post smoothing
  int pre_degree = -1;           // >= 0: pre-smoothing steps differ from `degree` (non-symmetric cycle)
  double eig_ratio = 4.0;        // smoothing interval [lmax / ratio, lmax]
  int coarse_dense_max = 1200;
  int coarse_steps = 30;
  bool dense_coarse = true, ready = false;
  DevBuf<double> coarse_inv, parts;
  // partitioned meshes: communicator + replicated global coarsest problem
  Comm* comm = nullptr;          // set only when the context is partitioned (or forced)
  bool comm_active() const { return comm != nullptr; }
  const BlockMat* globA = nullptr;
  int n_glob = 0;
  int64_t glob_off = 0;
  // unstructured partitions: global id of every local coarsest node instead of offset + i
  const int32_t* glob_idx = nullptr;
  const std::vector<int32_t>* h_glob_idx = nullptr;
  size_t glob_of(size_t i) const {
    return h_glob_idx ? (size_t)(*h_glob_idx)[i] : ((size_t)glob_off + i) % (size_t)n_glob;
  }
  bool glob_wrap = false;        // periodic partitions: the local coarsest level wraps around the global numbering
  DevBuf<double> gb, gx;
  // when the global coarsest mesh is too large for a dense solve it carries its own REPLICATED
  // hierarchy: a serial multigrid (no communicator) every rank runs redundantly on the
  // all-reduced right-hand side -- the small levels cost no halo exchanges at all
  Multigrid* tail = nullptr;     // not owned
  // in-situ timing of the finest-level smoothing launches (bench.py's roofline figure): one
  // HIP-event pair per launch while enabled
  bool prof = false;
  std::vector<hipEvent_t> prof_ev;
  size_t prof_n = 0;
  bool prof_open = false;        // an event pair is open inside the current smooth() call
  int64_t prof_launches = 0;     // launches covered by the recorded pairs
  int64_t lattice_launches = 0;  // launches of the multi-step lattice kernel by this hierarchy (nsfem_mg_info)
  int64_t prof_steps = 0;        // smoothing steps covered (the lattice kernel runs several per launch)
  int64_t prof_bytes = 0;        // algorithmic bytes of the covered launches (lattice kernel: per launch shape)
  bool own_mask0 = false;        // level 0 keeps its own mask buffer (tails)
  bool smoother_only = false;    // the last level is smoothed (coarse_steps), never solved globally
  // preconditioner of a Newton matrix with dolfin-style Dirichlet rows: the result carries z = r on
  // the rows flagged 1 of level 0 -- written by the epilogue of the last finest-level smoothing step
  // instead of a separate copy kernel
  bool identity_rows = false;
  // partitioned meshes, relaxed mode: a smoothing sequence exchanges the ghost values ONCE (at
  // its first step that reads them) and keeps them frozen afterwards -- Chebyshev iteration on
  // the rank-local operator (block-Jacobi across ranks) around the true residual, still a
  // symmetric preconditioner; prolongations compute the ghost rows themselves.  Default (false):
  // one exchange per SpMV, the partitioned cycle IS the serial one.
  bool relaxed_halo = false;
  // truncated cycle (mass-dominated operators): only the first `active` levels are used and the
  // last of them is SOLVED by `trunc_steps` Chebyshev steps over its whole spectrum
  // [trunc_lmin, lmax] -- no coarser level, no dense / global coarse solve, no all-reduce
  size_t active = 0;             // 0: all levels
  double trunc_lmin = 0.0, trunc_tol = 0.1;
  int trunc_steps = 0;
  bool truncated() const { return active > 0 && active < lv.size(); }
  void refresh_global_coarse(hipStream_t s, const std::vector<uint8_t>& cur, bool singular);
  void halo_fill(hipStream_t s, const MGLevel& L, const double* v);
  void apply_additive(hipStream_t s, MGLevel& L, const double* x, double* t);
  void smooth_additive(hipStream_t s, MGLevel& L, const double* b, const double* x_in, double* x_out,
                       int steps, bool first_done);
  void setup_work(hipStream_t s);
  void refresh(hipStream_t s, const std::vector<uint8_t>& mask0, bool singular);
  void apply(hipStream_t s, const double* r, double* z) override;
  // result in x (level 0: always) or in a buffer of the level: the returned pointer says where
  const double* vcycle(hipStream_t s, size_t l, const double* b, double* x, bool first_done = false);
  // multi-step lattice kernel (2D lexicographic lattices, serial levels): `steps` Chebyshev steps out of
  // place, optionally the residual of the result as well
  bool lattice_ok(const MGLevel& L) const;
  // partitioned level in relaxed halo mode whose ghost nodes are whole lattice lines: smoothing sequences run in
  // the multi-step lattice kernel with the ghost lines frozen (transfers stay explicit products)
  bool lattice_ok_relaxed(MGLevel& L);
  bool chain_child_forms_b(size_t l);
  size_t restricted_to = 0;      // level whose b the two-level restriction kernel has already formed in this leg
  // fused legs (mglegs.hip): built on first use after a refresh; legs_kind 0 = none (separate launches),
  // 1 = truncated cycle without pre-smoothing: ONE launch for everything below the finest level,
  // 2 = V(pre, post) cycle with a dense coarsest level: down-legs, one single-workgroup tail, up-legs
  int legs_kind = -1;            // -1: not tried since the last refresh
  std::vector<std::unique_ptr<LegPlan>> legs_down, legs_up;
  std::unique_ptr<LegPlan> leg_tail, leg_coarse;
  int64_t leg_launches = 0;
  void ensure_sidm(hipStream_t s, MGLevel& L);
  bool leg_level_ok(size_t l) const;
  void build_legs(hipStream_t s);
  bool vcycle_legs(hipStream_t s, const double* b, double* x);
  // xc: the start vector is [x_in +] P xc (prolongation fused into the staging); rf: b = R rf is computed by the
  // first launch and stored to `b` (restriction fused)
  void smooth_lattice(hipStream_t s, MGLevel& L, const double* b, const double* x_in, double* x_out,
                      int steps, bool ident_last, double* r_out, const double* xc = nullptr,
                      const double* rf = nullptr);
  bool transfer_lattice(size_t l);      // levels l, l + 1: lattice operators and a lattice interpolation between them
  const double* vcycle_lattice(hipStream_t s, size_t l, const double* b, double* x, const double* rf);
  void smooth(hipStream_t s, MGLevel& L, const double* b, const double* x_in, double* x_out,
              int steps, bool ghosts_valid = false, bool ident_last = false, bool first_done = false);
  // restriction to level l + 1 fused with the first smoothing step there (when that level starts
  // from zero): returns true when L[l+1].xa / .d already hold step 0
  bool restrict_to(hipStream_t s, size_t l, const double* src);
  bool starts_from_zero(size_t l) const;
  void cheb_coeffs(const MGLevel& L, int k, double rho_prev, double& c1, double& c2,
                   double& rho) const;
};

// Fast diagonalisation of the P1 stiffness matrix of a tensor-product lattice (fastdiag.hip, poisson_fd.py):
// z = A^+ r by four dense products on the matrix cores
struct FastDiag : Precond {
  int W = 0, H = 0;
  // partitioned strips: this rank holds the lattice lines j0 ... j0 + h_loc - 1 (ghost lines included) of the GLOBAL
  // W x H lattice; Vy then keeps those rows of V_y only.  h_loc == 0: the whole lattice (one rank)
  int j0 = 0, h_loc = 0;
  DevBuf<double> Vx, Vy, inv, t1, t2;
  int64_t applications = 0;
  bool ready() const { return W > 0 && Vx.p && Vy.p && inv.p; }
  bool strip() const { return h_loc > 0; }
  void set(hipStream_t s, int W_, int H_, const double* vx, const double* vy, const double* inv_);
  void set_rows(hipStream_t s, int W_, int H_, int j0_, int h_loc_, const double* vx, const double* vy,
                const double* inv_);
  void apply(hipStream_t s, const double* r, double* z) override;
  // strips: r with zero ghost rows in, z on EVERY local row (ghost lines included) out; one all-reduce of the
  // H x W transformed array in between
  void apply_strip(hipStream_t s, Comm* comm, const double* r, double* z);
};
// x -= sum(parts) / count on n entries (k_sum fills the slot: launch_sum)
void launch_sum(hipStream_t s, int64_t n, const double* x, double* parts);
void launch_gather_diag(hipStream_t s, int n, const int32_t* diag, const double* vals, double* out);   // out[i] = vals[diag[i]]
void launch_sub_mean(hipStream_t s, int64_t n, int64_t count, const double* parts, double* x);

// z[dofs] = r[dofs]
void launch_copy_at(hipStream_t s, int n, const int32_t* dofs, const double* r, double* z);
// mask[i] = 2 where ghost[i] != 0
void launch_overlay_ghost(hipStream_t s, int64_t n, const uint8_t* ghost, uint8_t* mask);
// communicators (comm.hip)
Comm* make_local_comm(void* group, int rank);
Comm* make_rccl_comm(const char* id128, int rank, int size);
Comm* make_shm_comm(const char* name, int rank, int size, int64_t slot_bytes);

// Jacobi-preconditioned BiCGStab; x holds the initial guess on entry
int bicgstab(hipStream_t s, KrylovWork& w, const LinOp& op, const double* b, double* x,
             const nsfem_krylov_opts& o, nsfem_solve_info& info);
// Jacobi-preconditioned CG (rowmask in MASK_ZERO mode => symmetric elimination);
// project_mean: remove the mean of residuals (singular Neumann problem)
int pcg(hipStream_t s, KrylovWork& w, const LinOp& op, const double* b, double* x,
        const nsfem_krylov_opts& o, nsfem_solve_info& info, bool project_mean);

double host_sum_parts(hipStream_t s, KrylovWork& w, int which);   // sync + sum
void host_sum_parts2(hipStream_t s, KrylovWork& w, int a, int b, double& ra, double& rb);   // two slots, one round trip
void host_sum_parts3(hipStream_t s, KrylovWork& w, int a, int b, int c, double& ra, double& rb, double& rc);

}  // namespace nsfem

// the context
struct nsfem_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  nsfem::MeshDev mesh;
  nsfem::Pattern p22, p11, p12, p21;
  nsfem::BlockMat M2, K2, Ap, Mp, Dv, Gr, DT, L, J, E;   // constant + per-step operators
  bool have_E = false;
  int traction_form = 0;
  double coef[6] = {1.0, 1.0, 1.0, NAN, NAN, NAN};
  double alpha[3] = {1.0, -1.0, 0.0};
  // rotating frame: angular velocity and its time derivative (2D: scalars about e_z, rot_field =
  // nodal (-y, x); 3D: vectors, rot_field = nodal coordinates x)
  double omega = 0.0, omega_dot = 0.0;
  double omega3[3] = {0.0, 0.0, 0.0}, omega_dot3[3] = {0.0, 0.0, 0.0};
  nsfem::DevBuf<double> rot_field, rot_tmp;
  double k = 1.0;
  bool L_dirty = true;
  nsfem::DevBuf<double> state[NSFEM_N_SLOTS];
  bool have_body_force = false, have_traction = false;
  // Dirichlet data
  int nbc_v = 0, nbc_p = 0;
  nsfem::DevBuf<int32_t> bc_v_dofs, bc_p_dofs;
  nsfem::DevBuf<double> bc_v_vals, bc_p_vals;
  nsfem::DevBuf<uint8_t> mask_v, mask_p;
  // per-system work vectors
  nsfem::DevBuf<double> rhs_v, rhs_p, dx_v, gconst, tmp_v, tmp_p, dinv_v, dinv_p, dinv_m;
  bool dinv_p_ready = false, dinv_m_ready = false;
  int64_t mass_dinv_epoch = 0, mass_dinv_copied_epoch = -1;   // dinv_m recomputed / copied into the mass smoother
  nsfem::KrylovWork kw;
  int assembled_system = -1;
  double area = 0.0;
  uint64_t graph_epoch = 1; // captured iteration graphs are valid within one epoch
  int conv_form = 0;        // 0 standard, 1 rotational, 2 divergence, 3 skew-symmetric
  bool picard = false;      // Picard instead of Newton linearisation of the convection
  // ---- multigrid hierarchy (optional; nsfem_mg_add_level / nsfem_mg_finalize)
  struct P1Level {
    int n = 0;
    nsfem::MeshDev mesh;
    nsfem::Pattern pat;
    nsfem::BlockMat K, M, Lc;
    nsfem::StencilDict dict;                   // lattice levels: rows of K, M and every combination of them
    nsfem::Transfer to_finer;
    std::vector<uint8_t> h_ghost;              // per node (partitioned meshes)
    nsfem::HaloRange halo;
    bool has_halo = false;
  };
  // ---- partitioned meshes (nsfem_set_partition + nsfem_comm_attach_*)
  nsfem::Comm* comm = nullptr;                 // owned
  nsfem::HaloRange halo_p2, halo_p1;
  std::vector<uint8_t> h_ghost_p2, h_ghost_p1; // per node: nonzero = ghost
  int p2_gh_lo = -2, p2_gh_hi = -2;            // ghost lattice lines of the P2 lattice of a strip (-2: not looked at, -1: none such)
  nsfem::DevBuf<uint8_t> ghost_v, ghost_p;     // per vector entry: 0 / 2
  int64_t n_p2_global = 0, n_p1_global = 0;
  P1Level* global_coarse = nullptr;            // replicated global coarsest mesh (owned)
  std::vector<P1Level*> global_tail;           // its coarser levels, finest first (owned)
  nsfem::Multigrid mg_p_tail, mg_v_tail;       // replicated hierarchies below global_coarse
  nsfem::Multigrid mg_mv;                      // one level: Chebyshev solver of the velocity mass matrix
  double mass_kappa = 0.0;
  double mg_trunc_ratio = 4.0;                 // > 0: truncate the velocity cycle where nu K_ii <= ratio * alpha M_ii
  double mg_trunc_tol = 0.1;
  int64_t glob_off = 0;
  std::vector<int32_t> h_glob_idx;             // unstructured partitions: local coarsest node -> global id
  nsfem::DevBuf<int32_t> glob_idx;
  std::vector<nsfem::HaloLists*> halo_lists;   // owned (nsfem_set_halo_lists)
  bool partition_periodic = false;
  bool overlap = false;                        // halo exchanges under the interior rows (nsfem_set_overlap)
  double area_global = 0.0;                    // measure of the whole (partitioned) domain, lazily all-reduced
  // NSFEM_FORCE_COMM=1 routes a single-rank run through the communicator as well (lets a
  // one-GPU box exercise the RCCL all-reduce calls)
  bool distributed() const {
    static const bool force = std::getenv("NSFEM_FORCE_COMM") != nullptr;
    return comm && (comm->size > 1 || force);
  }
  std::vector<int32_t> h_p2map, h_p1map;       // host copies of the fine dof maps
  std::vector<P1Level*> coarse;                // owned
  nsfem::Transfer t_p2p1;                      // P2 (fine mesh) <- P1 (fine mesh)
  nsfem::BlockMat Lc0;                         // alpha0/k M_p + c_v A_p on the fine P1 space
  nsfem::Multigrid mg_p, mg_v;
  bool cor_start_ready = false;                // correction_assemble produced the mass solve's start residual and sums
  uint64_t cor_start_touch = 0;                //   ... and nobody has used the Krylov work vectors since (kw.touch)
  nsfem::FastDiag fd_p;                        // direct projection-step solver on tensor-product lattices
  bool fd_p_singular = false;
  bool mg_built = false, mg_p_dirty = true, mg_v_dirty = true;
  std::vector<int32_t> h_bc_v, h_bc_p;         // host copies of the Dirichlet dof sets
  struct MomentumPrec : nsfem::Precond {
    nsfem_ctx* c = nullptr;
    void apply(hipStream_t s, const double* r, double* z) override;
  } mom_prec;
  // host-supplied CSR operators (algebraic Schur Laplacian per multigrid level)
  struct CsrOp {
    nsfem::Pattern pat;
    nsfem::BlockMat mat;
    nsfem::StencilDict dict;                   // lattice meshes: smoothing steps run on the dictionary copy
  };
  std::vector<CsrOp*> schur_ops;               // owned
  nsfem::StencilDict dict22;                   // rows of the scalar P2 operators (ensure_L)
  bool dict22_tried = false;
  nsfem::StencilDict dict11;                   // rows of the scalar P1 operators of the fine mesh
  bool dict11_tried = false;
  nsfem::StencilDict dict21, dict12, dict21g;  // divergence-transpose / divergence / gradient blocks
  bool dictD_tried = false;
  int bc_p_any = -1;                           // partitioned: pressure Dirichlet dofs on any rank (-1 unknown)
  bool schur_additive = false;                 // operators of nsfem_mg_set_schur_operator are rank parts
  int schur_singular = -1;                     // -1: geometric hierarchy (singular iff no Dirichlet set)
  // monolithic BDF system: mixed operator, block preconditioner and their data
  nsfem::Multigrid mg_s, mg_m;                 // Schur Laplacian V-cycle, pressure-mass smoother
  bool mg_s_dirty = true;
  std::vector<int32_t> h_bc_s;
  nsfem::DevBuf<uint8_t> mask_s;
  nsfem::DevBuf<double> rhs_m, dx_m;
  // matrix-free velocity Jacobian (tetrahedral meshes): y = L x + c_c [d conv(u)/du] x with
  // identity rows on Dirichlet dofs and zero ghost rows; u = state[vel_slot]
  struct MomentumMF : nsfem::Operator {
    nsfem_ctx* c = nullptr;
    int vel_slot = 3;               // NSFEM_USTAR
    void apply(hipStream_t s, const double* x, double* y) override;
  } mom_mf;
  nsfem::DevBuf<uint8_t> mask_m;     // ghost flags of the pressure mass smoother (partitioned)
  double prec_shift = 0.0;          // mass shift of the velocity / Schur preconditioners
  nsfem::BlockMat Lprec;            // (alpha0/k + shift) M + c_v K when shift != 0
  // iteration predictor of a recurring solve (see hinted / next_hint in api.hip): the count the same solve needed
  // in the previous step, how many solves in a row needed exactly that count, and the number of solves so far
  struct SolveHint { int its = 0, same = 0; int64_t count = 0; };
  SolveHint hint_mom[4], hint_poi, hint_cor;
  // in-situ timing of the matrix-free convection action (k_conv_cell + k_res_gather): one HIP-event
  // pair per application while enabled (nsfem_profile_convection; bench.py's assembly roofline)
  struct Probe {
    bool on = false;
    std::vector<hipEvent_t> ev;
    size_t n = 0;
    ~Probe() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
  } conv_probe;
  int64_t jac_lattice_launches = 0;   // applications of the matrix-free Jacobian through k_jac_lattice
  bool mf_active = false;           // the running step driver applies the Jacobian matrix-free
  int pressure_history = 0;         // IPCS: pressure levels shifted since the state was last set (0..2)
  struct MixedOp : nsfem::Operator {
    nsfem_ctx* c = nullptr;
    void apply(hipStream_t s, const double* x, double* y) override;
  } mixed_op;
  struct BlockPrec : nsfem::Precond {
    nsfem_ctx* c = nullptr;
    void apply(hipStream_t s, const double* r, double* z) override;
  } block_prec;
  ~nsfem_ctx() {
    if (comm) comm->release_streams();
    for (P1Level* p : coarse) delete p;
    for (nsfem::HaloLists* h : halo_lists) delete h;
    for (CsrOp* p : schur_ops) delete p;
    for (P1Level* p : global_tail) delete p;
    delete global_coarse;
    delete comm;
  }
};
