/*
 * nsfem.h -- C ABI of libnsfem_hip.so: MI355X (gfx950) implementation of the
 * per-time-step Taylor-Hood (P2/P1) assembly + sparse solve that the reference
 * (LKM-code-base/NavierStokes-with-Fenics) delegates to FEniCS/PETSc.
 *
 * Nothing like this interface exists in the reference (it is pure Python on top
 * of dolfin).  Every entry point names the reference call it replaces; the
 * Python binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; every function returns int (0 = ok, <0 = nsfem_status),
 *     no C++ exception crosses the boundary; nsfem_last_error() gives the text.
 *   - the caller owns every host buffer it passes (read during the call only);
 *     the library owns all device memory; nothing returned outlives
 *     nsfem_destroy().
 *   - one context per process x device, one HIP stream per context; calls on a
 *     context are not re-entrant (the reference is single threaded).
 *   - all floating point data is fp64, all indices int32.
 *   - dim = 2 (triangles) or 3 (tetrahedra); velocity vectors are node-interleaved:
 *     index = dim * p2_node + component; "mixed" vectors are
 *     [velocity (dim * n_p2) | pressure (n_p1)].
 */
#ifndef NSFEM_H
#define NSFEM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nsfem_ctx nsfem_ctx;

enum nsfem_status {
  NSFEM_OK = 0,
  NSFEM_ERR_ARG = -1,         /* bad argument / wrong call order            */
  NSFEM_ERR_HIP = -2,         /* HIP runtime error (no device, OOM, ...)     */
  NSFEM_ERR_BREAKDOWN = -3,   /* Krylov breakdown (rho / omega = 0, NaN)     */
  NSFEM_ERR_NOT_CONVERGED = -4, /* Newton / Krylov hit its iteration limit   */
  NSFEM_ERR_COMM = -5         /* RCCL error                                  */
};

/* Mesh + dof maps: what dolfin.Mesh / FunctionSpace(mesh, P2^d x P1) hold in the
 * reference (source/ns_solver_base.py:501-524).  Affine simplices: triangles (dim = 2) or
 * tetrahedra (dim = 3); local P2 order = vertices, then the edge midpoints in UFC edge order
 * (triangle: e(v1v2), e(v0v2), e(v0v1); tetrahedron: e(v2v3), e(v1v3), e(v1v2), e(v0v3), e(v0v2),
 * e(v0v1)). */
typedef struct {
  int32_t dim;               /* 2 or 3                                        */
  int32_t n_cells;
  int32_t n_vertices;
  int32_t n_p2;              /* scalar P2 nodes                               */
  int32_t n_p1;              /* P1 nodes                                      */
  const double* coords;      /* [n_vertices * dim]                            */
  const int32_t* cells;      /* [n_cells * (dim + 1)] vertex ids              */
  const int32_t* p2_dofmap;  /* [n_cells * 6 | 10]                            */
  const int32_t* p1_dofmap;  /* [n_cells * (dim + 1)]                         */
} nsfem_mesh_desc;

/* state slots (device-resident vectors) */
enum nsfem_slot {
  NSFEM_U0 = 0,      /* velocity at t_{n+1}   (IPCS _velocities[0]) [dim*n_p2] */
  NSFEM_U1 = 1,      /* velocity at t_n                                       */
  NSFEM_U2 = 2,      /* velocity at t_{n-1}                                   */
  NSFEM_USTAR = 3,   /* IPCS _intermediate_velocity                           */
  NSFEM_P = 4,       /* pressure                                    [n_p1]    */
  NSFEM_P_OLD = 5,   /* IPCS _old_pressure / BDF pressure at t_n              */
  NSFEM_BODY_FORCE = 6, /* nodal P2 interpolant of f             [dim*n_p2]   */
  NSFEM_TRACTION = 7,   /* assembled boundary traction vector    [dim*n_p2]   */
  NSFEM_P2_OLD = 8,  /* BDF: pressure at t_{n-1} (keeps _solutions[2] whole)  */
  NSFEM_N_SLOTS = 9
};

/* fields for Dirichlet sets */
enum nsfem_field {
  NSFEM_VELOCITY = 0,
  NSFEM_PRESSURE = 1,
  NSFEM_PRESSURE_PRECOND = 2   /* Dirichlet set of the pressure Laplacian used inside the
                                  Schur-complement preconditioner of the monolithic scheme
                                  (P1 nodes on open boundaries + true pressure conditions) */
};

/* operators that can be exported / applied (parity tests, _assemble_system) */
enum nsfem_operator {
  NSFEM_OP_MASS_P2 = 0,      /* scalar P2 mass               n_p2 x n_p2      */
  NSFEM_OP_STIFF_P2 = 1,     /* scalar P2 stiffness                           */
  NSFEM_OP_STIFF_P1 = 2,     /* (grad p, grad q)  ns_ipcs_solver.py:160       */
  NSFEM_OP_MASS_P1 = 3,
  NSFEM_OP_DIV = 4,          /* (div u, q)       n_p1 x dim n_p2              */
  NSFEM_OP_GRAD = 5,         /* (grad p, w)      dim n_p2 x n_p1              */
  NSFEM_OP_DIVT = 6,         /* (p, div w)       dim n_p2 x n_p1              */
  NSFEM_OP_MOMENTUM_JAC = 7, /* IPCS/BDF velocity block of the Newton matrix  */
  NSFEM_OP_VISCOUS_EXTRA = 8, /* traction-form extra block (grad u^T : grad v) */
  NSFEM_OP_MOMENTUM_JAC_MF = 9, /* nsfem_operator_apply only: matrix-free velocity Jacobian */
  NSFEM_OP_MOMENTUM_SMOOTHER = 10, /* nsfem_time_spmv only: finest-level Chebyshev step of the
                                     velocity multigrid (scalar P2 operator, fused epilogue) */
  NSFEM_OP_CONVECTION_ACTION = 11 /* nsfem_time_spmv only: matrix-free convection action
                                     (element kernel + node gather), cache-cold            */
};

enum nsfem_system {
  NSFEM_SYS_MOMENTUM = 0,    /* IPCS diffusion step  ns_ipcs_solver.py:106-147 */
  NSFEM_SYS_POISSON = 1,     /* projection step      ns_ipcs_solver.py:149-171 */
  NSFEM_SYS_CORRECTION = 2,  /* velocity correction  ns_ipcs_solver.py:173-196 */
  NSFEM_SYS_MONOLITHIC = 3   /* BDF mixed system     ns_bdf_solver.py:36-100   */
};

typedef struct {
  double rtol;          /* relative residual (to |b|) tolerance                */
  double atol;          /* absolute residual tolerance                         */
  int32_t max_iter;
  int32_t precond;      /* 0 = Jacobi, 1 = multigrid (where available), 2 = (velocity mass
                           solve only) Chebyshev iteration with a-priori element bounds:
                           no dot products, no all-reduce inside the iteration;
                           3 = (projection step only) direct solve by fast diagonalisation on
                           tensor-product lattices, see nsfem_poisson_set_fast_diag            */
  int32_t check_every;  /* host convergence check interval (>=1)               */
  int32_t first_check;  /* iterations before the first host convergence check (every check is a
                           device -> host round trip; the step drivers set it from the iteration
                           count the same solve needed in the previous step)     */
} nsfem_krylov_opts;

typedef struct {
  int32_t iterations;
  int32_t converged;
  double residual;      /* final |r|_2                                          */
  double residual0;     /* initial |r|_2                                        */
} nsfem_solve_info;

typedef struct {
  double newton_atol;   /* reference: tol (1e-10)   ns_ipcs_solver.py:144       */
  double newton_rtol;   /* reference: 10 * tol                                  */
  int32_t newton_max_iter; /* reference: 50                                     */
  int32_t convective_form; /* 0 standard, 1 rotational, 2 divergence, 3 skew-symmetric */
  nsfem_krylov_opts momentum;   /* BiCGStab */
  nsfem_krylov_opts poisson;    /* CG       */
  nsfem_krylov_opts correction; /* CG       */
  int32_t picard;               /* monolithic step: Picard instead of Newton matrix
                                   (StationarySolverBase, ns_solver_base.py:930-934)        */
  int32_t allow_nonconvergence; /* monolithic step: return instead of failing when the
                                   iteration limit is hit (error_on_nonconvergence=False)   */
  double newton_forcing;        /* 0: every Newton linear solve to the Krylov tolerances given
                                   (direct-solver accuracy, parity runs).  eta > 0: inexact
                                   Newton -- linear residual reduced by eta, at most down to a
                                   tenth of the nonlinear target; the Newton loop still stops on
                                   the reference's criterion (throughput runs)              */
  int32_t matrix_free;          /* velocity Jacobian inside the step drivers: 0 auto (= matrix-
                                   free: L x + linearised convection by an element kernel),
                                   1 assembled block CSR, 2 matrix-free                      */
  int32_t pressure_extrapolation; /* IPCS projection step: start the CG iteration from the linear
                                   extrapolation 2 p_n - p_(n-1) instead of p_n (the reference's direct
                                   solve has no initial guess; the converged pressure is the same, the
                                   iteration starts closer to it).  0 = off (default)               */
} nsfem_step_opts;

#define NSFEM_MAX_NEWTON 64
typedef struct {
  int32_t newton_iterations;
  int32_t krylov_iterations_momentum;   /* summed over Newton iterations        */
  int32_t krylov_iterations_poisson;
  int32_t krylov_iterations_correction;
  double newton_residuals[NSFEM_MAX_NEWTON]; /* |b| after 0,1,.. updates        */
  int32_t converged;
  int32_t reserved;
} nsfem_step_info;

/* ---- life cycle: replaces FunctionSpace/Function construction
 * (source/ns_solver_base.py:501-524,1018-1025; ns_ipcs_solver.py:66-82) ------ */
int nsfem_create(const nsfem_mesh_desc* mesh, int device, nsfem_ctx** out);
void nsfem_destroy(nsfem_ctx* ctx);
const char* nsfem_last_error(const nsfem_ctx* ctx);   /* ctx may be NULL: last create error */
int nsfem_version(void);

/* ---- coefficients: replaces set_equation_coefficients (ns_solver_base.py:829-855)
 * c = {convective, pressure, viscous, body_force, coriolis, euler}; NaN = None */
int nsfem_set_coeffs(nsfem_ctx* ctx, const double c[6]);
/* replaces _update_time_stepping_coefficients (ns_ipcs_solver.py:210-227) */
int nsfem_set_bdf(nsfem_ctx* ctx, const double alpha[3], double k);
/* replaces DirichletBC lists (ns_solver_base.py:546-660); re-callable each step
 * (time dependent values, _set_time ns_solver_base.py:1033-1104).  dofs index the
 * velocity (interleaved) or pressure vector; later entries win on duplicates. */
int nsfem_set_dirichlet(nsfem_ctx* ctx, int field, int32_t n, const int32_t* dofs,
                        const double* vals);
/* convective term form (ns_solver_base.py:370-390): 0 standard, 1 rotational, 2 divergence,
 * 3 skew-symmetric; picard != 0 selects the Picard linearisation (:478-499) for the matrices
 * assembled through nsfem_assemble (the fused step drivers take the form from their options) */
int nsfem_set_convective_form(nsfem_ctx* ctx, int form, int picard);
/* viscous term form: 0 reduced, 1 traction (ns_solver_base.py:662-673) */
int nsfem_set_viscous_form(nsfem_ctx* ctx, int traction_form);

/* ---- state transfer: Function.vector() get/set ------------------------------ */
int nsfem_set_state(nsfem_ctx* ctx, int slot, const double* host, int64_t n);
int nsfem_get_state(nsfem_ctx* ctx, int slot, double* host, int64_t n);
int64_t nsfem_state_size(const nsfem_ctx* ctx, int slot);
void* nsfem_state_devptr(nsfem_ctx* ctx, int slot);   /* device pointer (plumbing) */

/* ---- the explicit assembly seam (_assemble_system) ------------
 * Assembles matrix and right-hand side / residual of one system from the current
 * state; replaces the implicit dolfin assemble() inside *VariationalSolver.solve() */
int nsfem_assemble(nsfem_ctx* ctx, int system, uint32_t flags);
/* 2-norm of the assembled residual / rhs (Dirichlet rows: x_i - g_i for Newton) */
int nsfem_residual_norm(nsfem_ctx* ctx, int system, double* out);
int nsfem_get_rhs(nsfem_ctx* ctx, int system, double* host, int64_t n);
/* Krylov solve of the assembled system; replaces PETSc LU */
int nsfem_solve(nsfem_ctx* ctx, int system, const nsfem_krylov_opts* opts,
                nsfem_solve_info* info);

/* ---- operator introspection for parity tests -------------------------------- */
int nsfem_operator_shape(nsfem_ctx* ctx, int op, int64_t* n_rows, int64_t* n_cols,
                         int64_t* nnz_scalar);
/* scalar CSR copy (blocks expanded); arrays sized from nsfem_operator_shape */
int nsfem_operator_export(nsfem_ctx* ctx, int op, int32_t* rowptr, int32_t* col, double* val);
/* diagonal of a square scalar operator (n_rows doubles) without exporting the matrix */
int nsfem_operator_diagonal(nsfem_ctx* ctx, int op, double* out);
/* y = op * x on the device through the production SpMV kernel (host in/out) */
int nsfem_operator_apply(nsfem_ctx* ctx, int op, const double* x, double* y);
/* Test hook (parity tests): ONE product / residual / Chebyshev-Jacobi smoothing sequence of the scalar
 * lattice operator  a M + b K  (space 0: P2 mass / stiffness of the fine mesh on nv interleaved
 * components; space 1: P1 mass / stiffness) through a CHOSEN kernel family, on host data -- so that every
 * SpMV kernel family and every epilogue can be pinned to the oracle's matrices directly.
 *   family   0 library default, 1 CSR (stream / lane-group), 2 SELL-64, 3 stencil dictionary (one step per
 *            launch), 4 multi-step lattice kernel (all steps in one launch; epilogue 3 only)
 *   epilogue 0  y = A x   1  y = b - A x   3  `steps` steps  d = c1[k] d + c2[k] D^-1 (b - A x), x += d
 *   maskmode 0 none, 1 identity rows, 2 zero rows (flags in `mask`, one per vector entry; flag 2 = ghost)
 * Outputs: y (result / last iterate), d_out, r_out (b - A y when with_residual), and which family ran. */
typedef struct {
  int32_t space, nv, family, epilogue, steps, maskmode, ghost, ident, from_zero, with_residual, dict_ok;
  int32_t used_family;        /* out */
  int32_t dict_entries, dict_exact, lattice_w;   /* out: dictionary of the pattern (0: none) */
  int32_t reserved;
  double a, b_coef;
  double c1[8], c2[8];
  const double *x, *b, *d;    /* host [n * nv]; b, d may be NULL */
  const uint8_t* mask;        /* host [n * nv] or NULL */
  double *y, *d_out, *r_out;  /* host [n * nv]; d_out, r_out may be NULL */
} nsfem_kernel_test;
int nsfem_kernel_apply(nsfem_ctx* ctx, nsfem_kernel_test* t);

/* ---- multigrid hierarchy (optional).  Coarse P1 levels are added finest-first; each
 * carries its mesh and the prolongation P (CSR, rows = nodes of the previous finer P1
 * level, n_fine of them; cols = nodes of this level).  Spaces must be nested (every
 * coarse node coincides with a finer node: a row of P with the single entry 1).  The
 * library integrates the coarse operators on the device, adds the P2 <- P1 transfer of
 * the fine mesh itself and builds two V-cycle preconditioners: pressure Poisson and
 * alpha0/k M + c_v K.  Selected per solve with nsfem_krylov_opts.precond = 1. */
/* contiguous halo ranges of a strip (2D) / slab (3D) partition, in node units (offset, count) */
typedef struct {
  int64_t send_up_off, send_up_cnt, recv_above_off, recv_above_cnt;
  int64_t send_down_off, send_down_cnt, recv_below_off, recv_below_cnt;
} nsfem_halo;
typedef struct {
  int32_t n_vertices, n_cells;
  const double* coords;      /* [n_vertices * dim]     */
  const int32_t* cells;      /* [n_cells * (dim + 1)]  */
  int32_t n_fine;
  const int32_t* p_rowptr;   /* [n_fine + 1]     */
  const int32_t* p_col;
  const double* p_val;
  const uint8_t* ghost;      /* one flag per P1 dof of the level ([n_vertices], or [n_dofs] with a
                                dofmap): nonzero = ghost; NULL on unpartitioned meshes */
  nsfem_halo halo;           /* used when ghost != NULL */
  /* constrained (periodic) spaces: the P1 dof of every cell vertex, [n_cells * (dim + 1)], with
   * n_dofs < n_vertices distinct ids (slaves share their master's dof); the geometry still comes
   * from coords[cells].  NULL / 0: dof = vertex id.  P then has n_dofs columns. */
  const int32_t* dofmap;
  int32_t n_dofs;
  int32_t transfer_kind;     /* how Dirichlet / ghost flags travel down this transfer: 1 = nested levels (a coarse node
                                takes the flag of the finer node it coincides with: the row of P with the single entry
                                1), 2 = non-nested interpolation (the finer node its hat function weighs most, weight
                                >= 1/2), 0 = decide from the values of P (every entry 1 or 1/2 -> nested) */
} nsfem_mg_level_desc;
typedef struct {
  int32_t smoother_degree;   /* Chebyshev steps per pre/post smoothing (default 2) */
  int32_t coarse_dense_max;  /* dense coarse solve up to this many unknowns (1200)  */
  double eig_ratio;          /* smoothing interval [lmax/ratio, lmax] (default 4)   */
} nsfem_mg_opts;
int nsfem_mg_add_level(nsfem_ctx* ctx, const nsfem_mg_level_desc* level);
/* monolithic scheme: algebraic pressure Laplacian  D_f diag(M_v)^{-1} D_f^T  (and its Galerkin
 * coarsenings) for level `level` of the Schur-complement hierarchy; scalar CSR with a stored
 * diagonal, n = number of P1 nodes of that level (level 0 = fine mesh); singular != 0 when no
 * boundary is open (constants in the kernel: the coarse solve uses the pseudo-inverse) */
int nsfem_mg_set_schur_operator(nsfem_ctx* ctx, int level, int32_t n, const int32_t* rowptr,
                                const int32_t* col, const double* val, int singular);
/* partitioned meshes (additive != 0, call before nsfem_mg_set_schur_operator): the operators are
 * the rank's ADDITIVE parts  D_r W_r D_r^T  (W_r = 1 / M_v,jj on the velocity dofs the rank owns,
 * 0 on ghosts and Dirichlet dofs), ghost rows included, and their Galerkin coarsenings with the
 * rank-local prolongations: their sum over the ranks is the operator.  Products run as forward
 * halo exchange -> local product -> reverse (add) exchange; the global coarsest matrix is the
 * all-reduced dense sum of the coarsest parts (the partition must not carry a replicated tail).
 * `singular` must be the same on every rank (nsfem_comm_allreduce). */
int nsfem_mg_set_schur_mode(nsfem_ctx* ctx, int additive);
/* sum (op = 0) / max (op = 1) over the ranks of `count` (<= 1024) host doubles, in place;
 * single contexts: a no-op */
int nsfem_comm_allreduce(nsfem_ctx* ctx, double* values, int count, int op);
/* partitioned hierarchies: the GLOBAL coarsest mesh (solved redundantly on every rank);
 * offset = global id of this rank's local coarsest node 0 */
int nsfem_mg_set_global_coarse(nsfem_ctx* ctx, int32_t n_vertices, int32_t n_cells,
                               const double* coords, const int32_t* cells, int64_t offset);
/* coarser levels of a REPLICATED hierarchy below the global coarsest mesh (finest first): when
 * that mesh is too large for a dense solve every rank runs the remaining V-cycle redundantly on
 * the all-reduced right-hand side, so the small levels cost no halo exchange */
int nsfem_mg_add_global_level(nsfem_ctx* ctx, const nsfem_mg_level_desc* level);
/* nsfem_mg_set_global_coarse for constrained (periodic) spaces: dofmap [n_cells * (dim + 1)] with
 * n_dofs distinct P1 dofs; on periodic partitions the local coarsest level may wrap around the end
 * of the global numbering (offset + local size > n_dofs) */
int nsfem_mg_set_global_coarse_constrained(nsfem_ctx* ctx, int32_t n_vertices, int32_t n_cells,
                                           const double* coords, const int32_t* cells,
                                           const int32_t* dofmap, int32_t n_dofs, int64_t offset);
int nsfem_mg_finalize(nsfem_ctx* ctx, const nsfem_mg_opts* opts /* may be NULL */);
/* truncated velocity cycle for mass-dominated operators (small time steps): the first P1 level
 * on which  c_v K_ii <= max_ratio * (alpha0/k) M_ii  for every node is solved by Chebyshev
 * iteration to the relative accuracy coarse_tol (a-priori spectral bounds), and the levels below
 * it -- including the dense / global coarse solve and its all-reduce -- leave the cycle.
 * Defaults: max_ratio = 4, coarse_tol = 0.1; max_ratio = 0 disables the truncation.  Re-evaluated whenever the step size or the
 * coefficients change. */
int nsfem_mg_set_truncation(nsfem_ctx* ctx, double max_ratio, double coarse_tol);
/* partitioned meshes: relaxed = 0 (default) exchanges the ghost values before every SpMV of the
 * multigrid preconditioners, so the partitioned cycle IS the serial one; relaxed = 1 exchanges
 * once per smoothing sequence and smooths with frozen ghost values in between (Chebyshev on the
 * rank-local operator around the true residual: block-Jacobi across ranks, still a symmetric
 * preconditioner) -- about a third fewer halo exchanges per step, iteration counts may differ
 * slightly from the serial run.  Krylov operators, residuals and the mass solve stay exact. */
int nsfem_mg_set_halo_mode(nsfem_ctx* ctx, int relaxed);
/* partitioned meshes: enable != 0 runs the halo exchange of every Krylov operator application and
 * smoothing step on the communicator's own HIP stream, concurrently with the row blocks of the
 * product that reference no ghost column; the halo-adjacent row blocks follow after an event wait
 * (SURVEY.md section 8e "overlapped with interior-row SpMV").  Results are bitwise those of the
 * non-overlapped run.  Default: off. */
int nsfem_set_overlap(nsfem_ctx* ctx, int enable);
/* number of overlapped halo exchanges since the last reset */
int nsfem_comm_overlapped(nsfem_ctx* ctx, int64_t* out, int reset);

/* ---- multi-GPU: one process per GPU, each owning a strip of the mesh (new; the reference
 * is serial).  The context is created on the LOCAL mesh (own cell rows + one ghost row);
 * nsfem_set_partition marks the ghost nodes and the contiguous halo ranges; a communicator is
 * attached either over RCCL (production) or in-process (several contexts on one device, one
 * host thread per rank: single-GPU testing of the partitioned algorithm). */
typedef struct {
  int32_t rank, size;
  const uint8_t* p2_ghost;   /* [n_p2] nonzero = ghost (owned by a neighbour) */
  const uint8_t* p1_ghost;   /* [n_p1] */
  nsfem_halo p2_halo, p1_halo;
  int64_t n_p2_global, n_p1_global;
  int32_t periodic;          /* nonzero: the strips / slabs close periodically -- rank size-1 is
                                the lower neighbour of rank 0, halo exchanges wrap around */
} nsfem_partition_desc;
int nsfem_set_partition(nsfem_ctx* ctx, const nsfem_partition_desc* part);
/* Unstructured partitions (recursive coordinate bisection, partition.GraphPartition): a rank has
 * any number of neighbours and its ghost / send nodes are index lists.  For neighbour k (rank
 * neighbour[k]) this rank sends the nodes send_idx[send_ptr[k] .. send_ptr[k+1]) -- owned here,
 * ghost there -- and receives into recv_idx[recv_ptr[k] .. recv_ptr[k+1]); both sides order a
 * pair's list by global node id, so the k-th value sent is the k-th value received.
 * target 0: P2 nodes, 1: P1 nodes of the context (call nsfem_set_partition first, with zeroed
 * nsfem_halo ranges), 2 + l: P1 nodes of multigrid level l (after its nsfem_mg_add_level, ghost
 * flags given there).  Before nsfem_mg_finalize. */
typedef struct {
  int32_t n_neighbours;
  const int32_t* neighbour;  /* [n_neighbours] */
  const int64_t* send_ptr;   /* [n_neighbours + 1] */
  const int32_t* send_idx;
  const int64_t* recv_ptr;   /* [n_neighbours + 1] */
  const int32_t* recv_idx;
} nsfem_halo_lists;
int nsfem_set_halo_lists(nsfem_ctx* ctx, int target, const nsfem_halo_lists* lists);
/* unstructured partitions: global id (in the mesh of nsfem_mg_set_global_coarse, offset 0) of every
 * node of the local coarsest level */
int nsfem_mg_set_global_index(nsfem_ctx* ctx, int32_t n_local, const int32_t* local_to_global);
int nsfem_comm_unique_id(char* id128 /* 128 bytes out */);
int nsfem_comm_attach_rccl(nsfem_ctx* ctx, const char* id128, int rank, int size);
int nsfem_comm_local_create(int size, void** group);
void nsfem_comm_local_destroy(void* group);
int nsfem_comm_attach_local(nsfem_ctx* ctx, void* group, int rank);
/* one PROCESS per rank on ranks that SHARE a device (RCCL refuses two ranks on one GPU): host-staged
 * through the POSIX shared-memory segment `name` ("/..."; rank 0 creates it, slot_bytes per rank,
 * <= 0: 64 MiB).  Functional rehearsal of the process-per-rank launch on a one-GPU box, not a
 * performance transport.  Collective over the ranks (returns when all have attached). */
int nsfem_comm_attach_shm(nsfem_ctx* ctx, const char* name, int rank, int size, int64_t slot_bytes);
/* communication of this rank since the last reset: out = {all-reduce calls, all-reduce payload
 * bytes, halo exchanges, halo bytes sent}; zeros without a communicator */
int nsfem_comm_stats(nsfem_ctx* ctx, int64_t out[4], int reset);

/* ---- fused per-step drivers: replace _solve_time_step
 * (ns_ipcs_solver.py:198-208, ns_bdf_solver.py:102-106) ------------------------ */
int nsfem_default_step_opts(nsfem_step_opts* opts);
int nsfem_step_ipcs(nsfem_ctx* ctx, const nsfem_step_opts* opts, nsfem_step_info* info);
int nsfem_step_bdf(nsfem_ctx* ctx, const nsfem_step_opts* opts, nsfem_step_info* info);
/* replaces _advance_solution (ns_solver_base.py:1012-1016, ns_ipcs_solver.py:35-43) */
int nsfem_advance(nsfem_ctx* ctx, int scheme /* 0 ipcs, 1 bdf */);
/* L2 projection solve  M x = b  (no Dirichlet rows) on the velocity (field 0, both
 * components, b/x node-interleaved) or pressure (field 1) space; replaces the
 * mass-matrix LU inside dlfn.project (ns_solver_base.py:1151,1168).  b = int f phi_i
 * is supplied by the caller. */
int nsfem_mass_solve(nsfem_ctx* ctx, int field, const double* b, double* x,
                     const nsfem_krylov_opts* opts, nsfem_solve_info* info);
/* mean-pressure shift (ns_solver_base.py:1190-1203): p -= (int p / |Omega| - target) */
/* post-processing: (grad phi, grad psi) = rhs on the P1 space, phi = 0 on `dofs` (velocity
 * potential, reference source/ns_problem.py:105-176); pure Neumann data are mean-projected */
int nsfem_poisson_solve(nsfem_ctx* ctx, const double* rhs, int64_t n_dirichlet, const int32_t* dofs,
                        double* x, const nsfem_krylov_opts* opts, nsfem_solve_info* info);
int nsfem_shift_mean_pressure(nsfem_ctx* ctx, double target, double* mean_before);
/* stationary / very large time steps at high cell Peclet numbers: the multigrid V-cycle of the
 * velocity block and the Schur-complement approximation are built for (J + shift M) instead of J
 * ("time-step preconditioner", shift = 1/tau with |u| tau / h = O(1)).  Changes only the
 * preconditioner, never the equations.  shift = 0 (default): off. */
int nsfem_set_preconditioner_shift(nsfem_ctx* ctx, double shift);
/* rotating frame of reference (2D): adds  2 c_coriolis omega (e_z x u, w)  to the momentum
 * residual/Jacobian and  c_euler omega_dot (e_z x x, w)  to its right-hand side
 * (reference source/ns_solver_base.py:173-211); call again when omega changes in time */
int nsfem_set_angular_velocity(nsfem_ctx* ctx, double omega, double omega_dot);
/* 3D meshes: angular velocity VECTOR and its time derivative, cross(Omega, u) / cross(dOmega/dt, x)
 * (ns_solver_base.py:186-190, 207-209) */
int nsfem_set_angular_velocity_3d(nsfem_ctx* ctx, const double omega[3], const double omega_dot[3]);
/* CFL diagnostic of velocity slot `slot` for the step size k: max-norm of the cell-local P2
 * projection of  2 |u| k / h_circumdiameter  (reference source/ns_problem.py:554-587) */
int nsfem_cfl_number(nsfem_ctx* ctx, int slot, double step_size, double* cfl);

/* ---- boundary functionals of the solution: replaces dolfin.assemble(... * ds(subdomain_id)) in the
 * reference's post-processing hooks (demo/dfg_benchmark.py:44-66: drag / lift from the traction
 * -p n + 1/Re sym(grad u) n on the cylinder; demo/gravity_driven_flow.py:66-70: total mass flux).
 * facet_cell[k] = cell adjacent to boundary facet k, facet_local[k] = local index of the vertex
 * opposite to it in that cell (UFC local facet number).  out [dim + 2]:
 *   out[0..dim-1] = int ( -p n + nu (grad u + sym grad u^T) n ) dS   (n = outward unit normal)
 *   out[dim]      = int u . n dS ,   out[dim + 1] = int dS
 * One thread per facet, exact facet quadrature, per-facet values summed in facet order. */
int nsfem_boundary_force(nsfem_ctx* ctx, int velocity_slot, int pressure_slot, int32_t n_facets,
                         const int32_t* facet_cell, const int32_t* facet_local, double nu,
                         double sym, double* out);

/* ---- measurement hooks (bench.py): time `reps` launches of the dominant SpMV
 * with HIP events on the context's stream; ms per launch returned ------------- */
/* in-situ HIP-event timing of the finest-level smoothing launches of the velocity multigrid
 * (the dominant kernel of a time step), one event pair around each run of consecutive launches
 * of a smoothing sequence: enable != 0 starts sampling, enable == 0 stops and
 * reports average launch duration [ms], number of launches and algorithmic bytes per launch */
int nsfem_profile_smoother(nsfem_ctx* ctx, int enable, double* avg_ms, int64_t* launches,
                           int64_t* algorithmic_bytes);
/* detail of the window nsfem_profile_smoother just closed: out = {launches, smoothing steps they
 * ran (the multi-step lattice kernel runs up to 4 per launch), algorithmic bytes they moved in total,
 * 1 when the lattice kernel ran them} */
int nsfem_profile_smoother_detail(nsfem_ctx* ctx, int64_t out[4]);
int nsfem_time_spmv(nsfem_ctx* ctx, int op, int reps, double* ms_per_launch,
                    int64_t* algorithmic_bytes);
/* which finest-level smoothing kernel of the velocity multigrid runs: out = {kind (0 CSR-stream,
 * 1 SELL-64, 2 stencil dictionary), dictionary entries, longest row (negative: the dictionary
 * reproduces the matrix bit for bit and every product uses it), algorithmic bytes a CSR
 * stream of the same operator moves per smoothing launch}.  The stencil dictionary (lattice
 * meshes: rows with equal column offsets and values to 2^-40 of the largest entry share one entry;
 * a launch reads one byte per row instead of 12 bytes per nonzero) is used by smoothing steps and
 * Newton-Jacobian products only -- never by a residual or a linear operator whose solution is
 * returned; NSFEM_DICT=0 disables it. */
int nsfem_smoother_info(nsfem_ctx* ctx, int64_t out[4]);
/* Test hook: z = M^-1 r, one cycle of a multigrid preconditioner on host vectors (which: 0 = pressure Poisson
   hierarchy, 1 = velocity hierarchy); nsfem_mg_info: out = {fused-leg kind (0 separate launches, 1 one launch below
   the finest level of a truncated cycle, 2 down-legs + tail + up-legs), fused launches per cycle, levels in use,
   fused launches so far}; which = 2 / 3: the multi-step lattice kernel on the Poisson / velocity hierarchy: out =
   {levels whose smoothing sequences run in it (partitioned strips: relaxed halo mode only), its launches so far,
   levels in use, ghost lattice lines of the finest level as bottom * 256 + top}.  New functionality (the reference has no preconditioner: sparse LU,
   source/ns_ipcs_solver.py:171,205). */
int nsfem_mg_apply(nsfem_ctx* ctx, int which, const double* r, double* z);   /* which = 2: fast diagonalisation */
/* Direct solver of the projection step on tensor-product lattices (replaces the sparse LU of
   source/ns_ipcs_solver.py:160-171 where it applies): Vx [W x W], Vy [H x H] generalised eigenvectors of the 1D
   stiffness / lumped-mass pairs of the two directions (row-major, V^T W V = I, zero rows on Dirichlet sides),
   inv [H x W] = 1 / (lambda_y,j + lambda_x,i) (0: singular mode, Dirichlet slots).  The P1 space must be the W x H
   lattice in lexicographic numbering.  nsfem_krylov_opts.precond = 3 of the projection step then runs
   x += A^+ (b - A x) (four dense products on the matrix cores) and checks the residual.  Host side:
   poisson_fd.factors(). */
int nsfem_poisson_set_fast_diag(nsfem_ctx* ctx, int32_t W, int32_t H, const double* Vx, const double* Vy,
                                const double* inv);
/* The same on a partitioned strip (nsfem_set_partition): the factors of the GLOBAL W x H lattice; the context's P1
   space is the lattice lines first_line ... first_line + n_p1 / W - 1, ghost lines included.  The projection step of
   nsfem_step_ipcs (precond = 3, no pressure Dirichlet dofs) then costs ONE all-reduce of W x H doubles instead of the
   halo exchanges and dot-product reductions of a multigrid-CG solve, and returns the pressure with valid ghost rows. */
int nsfem_poisson_set_fast_diag_rows(nsfem_ctx* ctx, int32_t W, int32_t H, int32_t first_line, const double* Vx,
                                     const double* Vy, const double* inv);
int nsfem_mg_info(nsfem_ctx* ctx, int which, int64_t out[4]);
/* in-situ HIP-event timing of the matrix-free convection action of the velocity Jacobian inside
 * the Newton-Krylov solves (element kernel k_conv_cell / k3_conv_cell + node gather = the
 * per-iteration "assembly" of the fused step drivers; replaces the dolfin assemble(J) call of
 * ns_ipcs_solver.py:136-147 / ns_bdf_solver.py:88-100): enable != 0 starts sampling, enable == 0
 * stops and reports the average duration [ms] of one application, the number of applications and
 * the algorithmic bytes of one application (SURVEY.md section 8d formula).  One GPU, triangles, lattice
 * mesh: the per-node sums run inside the L-product launch, the event pair then brackets the element
 * kernel alone -- flagged by a NEGATIVE byte count (minus the element kernel's own bytes).  Lattice meshes in
 * rectangle_mesh numbering (nsfem_jacobian_info path 2): the pair brackets k_jac_lattice, i.e. the WHOLE Jacobian
 * action  L x + c_c [d conv(u)/du] x  in one launch; the byte count is nsfem_jacobian_info's out[2] */
int nsfem_profile_convection(nsfem_ctx* ctx, int enable, double* avg_ms, int64_t* applications,
                             int64_t* algorithmic_bytes);
/* How the matrix-free action of the velocity Jacobian (NSFEM_OP_MOMENTUM_JAC_MF; the operator of the Newton-Krylov
 * solves in nsfem_step_ipcs / nsfem_step_bdf, replacing the assembled J of ns_ipcs_solver.py:136-147) runs on this
 * context: out[0] = 0  L product, element kernel, node gather (three launches: partitioned meshes, tetrahedra,
 * unstructured meshes); 1  element kernel, then the dictionary product of L sums its node-sorted element vectors;
 * 2  k_jac_lattice -- one launch (2D lattice meshes in rectangle_mesh numbering on one GPU, gradient-form viscosity,
 * no rotating frame; NSFEM_JAC_LATTICE=0 disables it).  out[1] = applications through k_jac_lattice so far,
 * out[2] = its algorithmic bytes per application (u, x read, y written: 48 B per P2 node; 3 B per node of
 * dictionary ids and masks; 48 B per cell of vertex coordinates), out[3] = 0. */
int nsfem_jacobian_info(nsfem_ctx* ctx, int64_t out[4]);
/* extreme eigenvalues of diag(M_e)^-1 M_e of the P2 element mass matrix (host arithmetic only) */
int nsfem_p2_mass_bounds(int dim, double* lmin, double* lmax);
int nsfem_synchronize(nsfem_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* NSFEM_H */
