// C ABI of libnsfem_hip.so (declared in include/nsfem.h) and the per-step drivers.
//
// Step drivers restate, on device-resident state, what the reference does in
//   IPCSSolver._solve_time_step            source/ns_ipcs_solver.py:198-208
//   (diffusion Newton / projection / correction forms :106-196)
//   InstationarySolverBase._advance_solution   source/ns_solver_base.py:1012-1016
//   IPCSSolver._advance_solution               source/ns_ipcs_solver.py:35-43
// with dolfin's NewtonSolver control (residual criterion, relaxation 1; parameters
// from source/ns_ipcs_solver.py:143-147) and Krylov solves instead of sparse LU.
#include "nsfem_internal.hpp"
#include <thread>
#include <chrono>
#include <algorithm>

using namespace nsfem;

static std::string g_create_error;

#define API_BEGIN try {
#define API_END(ctx)                                  \
  }                                                   \
  catch (const nsfem::Error& e) {                     \
    if (ctx) (ctx)->err = e.what();                   \
    else g_create_error = e.what();                   \
    return e.code;                                    \
  }                                                   \
  catch (const std::exception& e) {                   \
    if (ctx) (ctx)->err = e.what();                   \
    else g_create_error = e.what();                   \
    return NSFEM_ERR_ARG;                             \
  }                                                   \
  return NSFEM_OK;

static inline int64_t nvel(const nsfem_ctx* c) { return (int64_t)c->mesh.dim * c->mesh.n_p2; }
static inline int64_t npre(const nsfem_ctx* c) { return (int64_t)c->mesh.n_p1; }

static int64_t slot_size(const nsfem_ctx* c, int slot) {
  switch (slot) {
    case NSFEM_U0: case NSFEM_U1: case NSFEM_U2: case NSFEM_USTAR:
    case NSFEM_BODY_FORCE: case NSFEM_TRACTION:
      return nvel(c);
    case NSFEM_P: case NSFEM_P_OLD: case NSFEM_P2_OLD:
      return npre(c);
    default:
      return -1;
  }
}

static HaloRange to_halo(const nsfem_halo& h) {
  HaloRange r;
  r.send_up_off = h.send_up_off; r.send_up_cnt = h.send_up_cnt;
  r.recv_above_off = h.recv_above_off; r.recv_above_cnt = h.recv_above_cnt;
  r.send_down_off = h.send_down_off; r.send_down_cnt = h.send_down_cnt;
  r.recv_below_off = h.recv_below_off; r.recv_below_cnt = h.recv_below_cnt;
  return r;
}

static void upload_pattern(hipStream_t s, HostPattern& h, Pattern& d, bool want_contrib = false) {
  d.n_rows = h.n_rows;
  d.n_cols = h.n_cols;
  d.nr = h.nr;
  d.nc = h.nc;
  d.nnz = (int)h.col.size();
  d.rowptr.upload(h.rowptr, s);
  d.col.upload(h.col, s);
  if (!h.diag.empty()) d.diag.upload(h.diag, s);
  d.slot.upload(h.slot, s);
  if (want_contrib) {
    if (h.cptr.empty()) {          // (patterns built without the index: serial fallback)
      const int64_t nc = (int64_t)h.slot.size() / (h.nr * h.nc), loc = h.nr * h.nc;
      build_inverse_index(d.nnz, nc * loc,
                          [&](int64_t src) { return h.slot[(size_t)(src % loc) * nc + src / loc]; },
                          h.cptr, h.cidx);
    }
    d.cptr.upload(h.cptr, s);
    d.cidx.upload(h.cidx, s);
    std::vector<int32_t>().swap(h.cptr);
    std::vector<int32_t>().swap(h.cidx);
  }
  d.h_rowptr.swap(h.rowptr);
  d.h_col.swap(h.col);
  build_rowblocks(d, s);
  std::vector<int32_t>().swap(h.slot);
}

extern "C" int nsfem_version(void) { return 1; }

extern "C" const char* nsfem_last_error(const nsfem_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

extern "C" int nsfem_create(const nsfem_mesh_desc* m, int device, nsfem_ctx** out) {
  nsfem_ctx* ctx = nullptr;
  nsfem_ctx* fresh = nullptr;
  API_BEGIN
  NSFEM_REQUIRE(m && out, "null argument");
  refresh_env_switches();
  refresh_assembly_switches();
  refresh_leg_switches();
  *out = nullptr;
  NSFEM_REQUIRE(m->dim == 2 || m->dim == 3, "dim must be 2 (triangles) or 3 (tetrahedra)");
  NSFEM_REQUIRE(m->n_cells > 0 && m->n_vertices > 0 && m->n_p2 > 0 && m->n_p1 > 0, "empty mesh");
  NSFEM_REQUIRE(m->coords && m->cells && m->p2_dofmap && m->p1_dofmap, "null mesh array");
  int ndev = 0;
  NSFEM_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) throw Error(NSFEM_ERR_HIP, "no such HIP device");
  NSFEM_HIP(hipSetDevice(device));
  fresh = new nsfem_ctx();
  fresh->device = device;
  NSFEM_HIP(hipStreamCreate(&fresh->stream));
  hipStream_t s = fresh->stream;
  const int nc = m->n_cells;
  // NSFEM_DEBUG_SETUP=1: wall-clock of the phases of this function on stderr
  const bool dbg_setup = std::getenv("NSFEM_DEBUG_SETUP") != nullptr;
  auto t_setup = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!dbg_setup) return;
    (void)hipStreamSynchronize(s);
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[nsfem_create] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_setup).count());
    t_setup = now;
  };
  // ---- mesh arrays, SoA
  const int dim = m->dim;
  NSFEM_REQUIRE(dim == 2 || dim == 3, "dim must be 2 (triangles) or 3 (tetrahedra)");
  const int nl1 = dim + 1, nl2 = dim == 2 ? 6 : 10;
  {
    std::vector<double> vx((size_t)nl1 * dim * nc);
    std::vector<int32_t> p2((size_t)nl2 * nc), p1((size_t)nl1 * nc);
    // (cell ranges on host threads; the cell volumes are summed afterwards in cell order: the same total on any
    // number of threads)
    std::vector<double> vol((size_t)nc);
    std::vector<int> bad((size_t)std::max(1, host_threads()), 0);
    auto fill_cells = [&](int c0, int c1, int tix) {
    for (int c = c0; c < c1; ++c) {
      double x[4][3] = {{0}};
      for (int v = 0; v < nl1; ++v) {
        const int vid = m->cells[(size_t)c * nl1 + v];
        if (vid < 0 || vid >= m->n_vertices) { bad[tix] = 1; return; }
        for (int d = 0; d < dim; ++d) {
          x[v][d] = m->coords[(size_t)vid * dim + d];
          vx[(size_t)(dim * v + d) * nc + c] = x[v][d];
        }
      }
      double det;
      if (dim == 2) {
        det = (x[1][0] - x[0][0]) * (x[2][1] - x[0][1]) - (x[2][0] - x[0][0]) * (x[1][1] - x[0][1]);
        vol[c] = 0.5 * std::fabs(det);
      } else {
        double a[3], b[3], e[3];
        for (int d = 0; d < 3; ++d) { a[d] = x[1][d] - x[0][d]; b[d] = x[2][d] - x[0][d]; e[d] = x[3][d] - x[0][d]; }
        det = a[0] * (b[1] * e[2] - b[2] * e[1]) - a[1] * (b[0] * e[2] - b[2] * e[0]) +
              a[2] * (b[0] * e[1] - b[1] * e[0]);
        vol[c] = std::fabs(det) / 6.0;
      }
      if (det == 0.0) { bad[tix] = 2; return; }
      for (int k = 0; k < nl2; ++k) p2[(size_t)k * nc + c] = m->p2_dofmap[(size_t)c * nl2 + k];
      for (int k = 0; k < nl1; ++k) p1[(size_t)k * nc + c] = m->p1_dofmap[(size_t)c * nl1 + k];
    }
    };
    {
      const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), nc / 65536));
      std::vector<std::thread> th;
      for (int t = 1; t < nt; ++t)
        th.emplace_back(fill_cells, (int)((int64_t)nc * t / nt), (int)((int64_t)nc * (t + 1) / nt), t);
      fill_cells(0, (int)((int64_t)nc / nt), 0);
      for (auto& x : th) x.join();
      for (int t = 0; t < nt; ++t) {
        NSFEM_REQUIRE(bad[t] != 1, "cell vertex id out of range");
        NSFEM_REQUIRE(bad[t] != 2, "degenerate cell");
      }
    }
    double area = 0.0;
    for (int c = 0; c < nc; ++c) area += vol[c];
    fresh->area = area;
    fresh->h_p2map.assign(m->p2_dofmap, m->p2_dofmap + (size_t)nl2 * nc);
    fresh->h_p1map.assign(m->p1_dofmap, m->p1_dofmap + (size_t)nl1 * nc);
    fresh->mom_prec.c = fresh;
    fresh->mesh.dim = dim;
    fresh->mesh.n_cells = nc;
    fresh->mesh.n_p2 = m->n_p2;
    fresh->mesh.n_p1 = m->n_p1;
    fresh->mesh.n_vertices = m->n_vertices;
    fresh->mesh.vx.upload(vx, s);
    fresh->mesh.p2.upload(p2, s);
    fresh->mesh.p1.upload(p1, s);
  }
  lap("mesh arrays");
  // ---- sparsity patterns + slot maps (host), then device copies
  // (round 4: built on the device by one radix sort per pattern, pattern_device.hip; NSFEM_PATTERN_HOST=1 keeps the
  // threaded host construction of pattern.cpp -- the two produce the same arrays)
  const bool pattern_on_host = std::getenv("NSFEM_PATTERN_HOST") != nullptr;
  if (!pattern_on_host) {
    for (int64_t q = 0; q < (int64_t)nc * nl2; ++q)
      NSFEM_REQUIRE(m->p2_dofmap[q] >= 0 && m->p2_dofmap[q] < m->n_p2, "dof map entry out of range (P2)");
    for (int64_t q = 0; q < (int64_t)nc * nl1; ++q)
      NSFEM_REQUIRE(m->p1_dofmap[q] >= 0 && m->p1_dofmap[q] < m->n_p1, "dof map entry out of range (P1)");
  }
  {
    HostPattern h;
    if (pattern_on_host) {
      build_pattern(m->n_p2, m->n_p2, nc, m->p2_dofmap, nl2, m->p2_dofmap, nl2, true, h, true);
      lap("pattern p22 (host)");
      upload_pattern(s, h, fresh->p22, true);
      lap("pattern p22 upload");
    } else {
      build_pattern_device(s, m->n_p2, m->n_p2, nc, fresh->mesh.p2.p, nl2, fresh->mesh.p2.p, nl2, true, fresh->p22);
      lap("pattern p22 (device)");
    }
    if (pattern_on_host) {
      std::vector<int32_t> ptr, idx;
      build_inverse_index(m->n_p2, (int64_t)nc * nl2,
                          [&](int64_t src) { return m->p2_dofmap[src]; }, ptr, idx);
      fresh->mesh.nptr.upload(ptr, s);
      std::vector<int32_t> dst(idx.size());
      for (size_t pos = 0; pos < idx.size(); ++pos) {
        const int64_t src = idx[pos];
        dst[(size_t)(src % nl2) * nc + (size_t)(src / nl2)] = (int32_t)pos;
      }
      fresh->mesh.ndst.upload(dst, s);
    } else {
      build_node_index_device(s, m->n_p2, nc, nl2, fresh->mesh.p2.p, fresh->mesh.nptr, fresh->mesh.ndst);
    }
    fresh->mesh.ebuf.alloc((size_t)nc * nl2 * nl2 * dim * dim);
    fresh->mesh.rbuf.alloc((size_t)nc * nl2 * dim);
    lap("node-sorted index + buffers");
    if (pattern_on_host) {
      build_pattern(m->n_p1, m->n_p1, nc, m->p1_dofmap, nl1, m->p1_dofmap, nl1, true, h, true);
      upload_pattern(s, h, fresh->p11, true);
      lap("pattern p11");
      build_pattern(m->n_p1, m->n_p2, nc, m->p1_dofmap, nl1, m->p2_dofmap, nl2, false, h, true);
      upload_pattern(s, h, fresh->p12, true);
      lap("pattern p12");
      build_pattern(m->n_p2, m->n_p1, nc, m->p2_dofmap, nl2, m->p1_dofmap, nl1, false, h, true);
      upload_pattern(s, h, fresh->p21, true);
      lap("pattern p21");
    } else {
      build_pattern_device(s, m->n_p1, m->n_p1, nc, fresh->mesh.p1.p, nl1, fresh->mesh.p1.p, nl1, true, fresh->p11);
      lap("pattern p11");
      build_pattern_device(s, m->n_p1, m->n_p2, nc, fresh->mesh.p1.p, nl1, fresh->mesh.p2.p, nl2, false, fresh->p12);
      lap("pattern p12");
      build_pattern_device(s, m->n_p2, m->n_p1, nc, fresh->mesh.p2.p, nl2, fresh->mesh.p1.p, nl1, false, fresh->p21);
      lap("pattern p21");
    }
  }
  // ---- constant operators, integrated on the device
  QuadTables qt;
  fill_quad_tables(qt);
  upload_quad_tables(qt);
  if (dim == 3) upload_quad_tables_3d();
  fresh->M2.init(&fresh->p22, 1, 1, s);
  fresh->K2.init(&fresh->p22, 1, 1, s);
  fresh->L.init(&fresh->p22, 1, 1, s);
  fresh->J.init(&fresh->p22, dim, dim, s);
  fresh->Ap.init(&fresh->p11, 1, 1, s);
  fresh->Mp.init(&fresh->p11, 1, 1, s);
  fresh->Dv.init(&fresh->p12, 1, dim, s);
  fresh->Gr.init(&fresh->p21, dim, 1, s);
  fresh->DT.init(&fresh->p21, dim, 1, s);
  launch_assemble_p2_scalar(s, fresh->mesh, fresh->p22, fresh->M2.vals.p, fresh->K2.vals.p);
  launch_assemble_p1_scalar(s, fresh->mesh, fresh->p11, fresh->Ap.vals.p, fresh->Mp.vals.p);
  launch_assemble_div_grad(s, fresh->mesh, fresh->p12, fresh->p21, fresh->Dv.vals.p,
                           fresh->Gr.vals.p, fresh->DT.vals.p);
  lap("operator assembly (device)");
  for (BlockMat* A : {&fresh->M2, &fresh->K2, &fresh->Ap, &fresh->Mp}) A->sell_update(s);
  lap("SELL / dictionary copies");
  // ---- state + work vectors
  for (int i = 0; i < NSFEM_N_SLOTS; ++i) {
    fresh->state[i].alloc((size_t)slot_size(fresh, i));
    fresh->state[i].zero(s);
  }
  const size_t nv = (size_t)nvel(fresh), np = (size_t)npre(fresh);
  for (DevBuf<double>* b : {&fresh->rhs_v, &fresh->dx_v, &fresh->gconst, &fresh->tmp_v,
                            &fresh->dinv_v, &fresh->dinv_m}) {
    b->alloc(nv);
    b->zero(s);
  }
  for (DevBuf<double>* b : {&fresh->rhs_p, &fresh->tmp_p, &fresh->dinv_p}) {
    b->alloc(np);
    b->zero(s);
  }
  fresh->mask_v.alloc(nv);
  fresh->mask_v.zero(s);
  fresh->mask_p.alloc(np);
  fresh->mask_p.zero(s);
  fresh->kw.ensure((int64_t)std::max(nv, np));
  NSFEM_HIP(hipStreamSynchronize(s));
  lap("state + work vectors");
  *out = fresh;
  fresh = nullptr;
  }
  catch (const nsfem::Error& e) {
    g_create_error = e.what();
    delete fresh;
    return e.code;
  }
  catch (const std::exception& e) {
    g_create_error = e.what();
    delete fresh;
    return NSFEM_ERR_ARG;
  }
  (void)ctx;
  return NSFEM_OK;
}

extern "C" void nsfem_destroy(nsfem_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamDestroy(ctx->stream);
  }
  delete ctx;
}

// element bounds of the Jacobi-scaled P2 mass matrix (host only; exposed for the CPU tests)
extern "C" int nsfem_p2_mass_bounds(int dim, double* lmin, double* lmax) {
  if ((dim != 2 && dim != 3) || !lmin || !lmax) return NSFEM_ERR_ARG;
  p2_mass_jacobi_bounds(dim, *lmin, *lmax);
  return NSFEM_OK;
}

extern "C" int nsfem_synchronize(nsfem_ctx* ctx) {
  API_BEGIN
  NSFEM_REQUIRE(ctx, "null context");
  NSFEM_HIP(hipStreamSynchronize(ctx->stream));
  API_END(ctx)
}

extern "C" int nsfem_set_coeffs(nsfem_ctx* ctx, const double c[6]) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && c, "null argument");
  NSFEM_REQUIRE(std::isfinite(c[1]) && std::isfinite(c[2]), "pressure and viscous coefficients are required");
  for (int i = 0; i < 6; ++i) ctx->coef[i] = c[i];
  ctx->L_dirty = true;
  ctx->graph_epoch++;                  // (coefficients are baked into captured kernel arguments)
  API_END(ctx)
}

extern "C" int nsfem_set_bdf(nsfem_ctx* ctx, const double alpha[3], double k) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && alpha, "null argument");
  // alpha = (0, 0, 0) selects the stationary equations (monolithic step only)
  NSFEM_REQUIRE(k > 0.0 && std::isfinite(k) && std::isfinite(alpha[0]), "bad BDF coefficients");
  if (alpha[0] != ctx->alpha[0] || k != ctx->k) ctx->L_dirty = true;
  if (alpha[0] != ctx->alpha[0] || alpha[1] != ctx->alpha[1] || alpha[2] != ctx->alpha[2] || k != ctx->k) ctx->graph_epoch++;
  for (int i = 0; i < 3; ++i) ctx->alpha[i] = alpha[i];
  ctx->k = k;
  API_END(ctx)
}

extern "C" int nsfem_set_convective_form(nsfem_ctx* ctx, int form, int picard) {
  API_BEGIN
  NSFEM_REQUIRE(ctx, "null context");
  NSFEM_REQUIRE(form >= 0 && form <= 3, "unknown convective form");
  if (ctx->conv_form != form || ctx->picard != (picard != 0)) ctx->graph_epoch++;
  ctx->conv_form = form;
  ctx->picard = picard != 0;
  API_END(ctx)
}

extern "C" int nsfem_set_viscous_form(nsfem_ctx* ctx, int traction_form) {
  API_BEGIN
  NSFEM_REQUIRE(ctx, "null context");
  ctx->traction_form = traction_form ? 1 : 0;
  ctx->graph_epoch++;
  if (ctx->traction_form && !ctx->have_E) {
    ctx->E.init(&ctx->p22, ctx->mesh.dim, ctx->mesh.dim, ctx->stream);
    launch_assemble_viscous_extra(ctx->stream, ctx->mesh, ctx->p22, ctx->E.vals.p);
    ctx->have_E = true;
  }
  API_END(ctx)
}

extern "C" int nsfem_set_dirichlet(nsfem_ctx* ctx, int field, int32_t n, const int32_t* dofs,
                                   const double* vals) {
  API_BEGIN
  NSFEM_REQUIRE(ctx, "null context");
  NSFEM_REQUIRE(field == NSFEM_VELOCITY || field == NSFEM_PRESSURE ||
                    field == NSFEM_PRESSURE_PRECOND, "bad field");
  NSFEM_REQUIRE(n >= 0 && (n == 0 || dofs), "bad Dirichlet arrays");
  if (field == NSFEM_PRESSURE_PRECOND) {
    // Dirichlet set of the pressure Laplacian inside the Schur-complement preconditioner of
    // the monolithic scheme (open boundaries + true pressure conditions); values unused
    std::vector<int32_t> d(dofs, dofs + n);
    std::sort(d.begin(), d.end());
    d.erase(std::unique(d.begin(), d.end()), d.end());
    for (int32_t k : d) NSFEM_REQUIRE(k >= 0 && k < npre(ctx), "Dirichlet dof out of range");
    if (d != ctx->h_bc_s || !ctx->mask_s.p) {
      DevBuf<int32_t> dd;
      dd.upload(d, ctx->stream);
      if (!ctx->mask_s.p) ctx->mask_s.alloc((size_t)npre(ctx));
      ctx->mask_s.zero(ctx->stream);
      launch_fill_mask(ctx->stream, (int)d.size(), dd.p, ctx->mask_s.p);
      NSFEM_HIP(hipStreamSynchronize(ctx->stream));
      ctx->h_bc_s.swap(d);
      ctx->mg_s_dirty = true;
    }
    return NSFEM_OK;
  }
  NSFEM_REQUIRE(n == 0 || vals, "bad Dirichlet arrays");
  const int64_t size = field == NSFEM_VELOCITY ? nvel(ctx) : npre(ctx);
  // later entries win on duplicates (list order of dolfin bc.apply): dedupe on host
  std::vector<int32_t> d;
  std::vector<double> v;
  {
    std::vector<int32_t> last((size_t)size, -1);
    for (int32_t i = 0; i < n; ++i) {
      NSFEM_REQUIRE(dofs[i] >= 0 && dofs[i] < size, "Dirichlet dof out of range");
      last[dofs[i]] = i;
    }
    for (int64_t k = 0; k < size; ++k)
      if (last[k] >= 0) {
        d.push_back((int32_t)k);
        v.push_back(vals[last[k]]);
      }
  }
  hipStream_t s = ctx->stream;
  DevBuf<int32_t>& dd = field == NSFEM_VELOCITY ? ctx->bc_v_dofs : ctx->bc_p_dofs;
  DevBuf<double>& dv = field == NSFEM_VELOCITY ? ctx->bc_v_vals : ctx->bc_p_vals;
  DevBuf<uint8_t>& mask = field == NSFEM_VELOCITY ? ctx->mask_v : ctx->mask_p;
  dd.upload(d, s);
  dv.upload(v, s);
  mask.zero(s);
  launch_fill_mask(s, (int)d.size(), dd.p, mask.p);
  launch_overlay_ghost(s, size, field == NSFEM_VELOCITY ? ctx->ghost_v.p : ctx->ghost_p.p, mask.p);
  std::vector<int32_t>& prev = field == NSFEM_VELOCITY ? ctx->h_bc_v : ctx->h_bc_p;
  const bool changed = prev != d;
  if (changed) ctx->graph_epoch++;  // the device dof arrays were re-allocated (values-only
                                    // updates keep the pointers: captured graphs stay valid)
  if (field == NSFEM_VELOCITY) {
    ctx->nbc_v = (int)d.size();
    if (changed) {
      ctx->dinv_m_ready = false;
      ctx->mg_v_dirty = true;
      for (nsfem::MGLevel& L : ctx->mg_mv.lv) L.sidm_for = nullptr;   // (the mass smoother shares mask_v)
    }
  } else {
    ctx->nbc_p = (int)d.size();
    ctx->bc_p_any = -1;            // partitioned meshes: agreed on by all ranks at the next solve
    if (changed) { ctx->dinv_p_ready = false; ctx->mg_p_dirty = true; }
  }
  prev.swap(d);
  NSFEM_HIP(hipStreamSynchronize(s));
  API_END(ctx)
}

extern "C" int64_t nsfem_state_size(const nsfem_ctx* ctx, int slot) {
  if (!ctx || slot < 0 || slot >= NSFEM_N_SLOTS) return -1;
  return slot_size(ctx, slot);
}

extern "C" void* nsfem_state_devptr(nsfem_ctx* ctx, int slot) {
  if (!ctx || slot < 0 || slot >= NSFEM_N_SLOTS) return nullptr;
  return ctx->state[slot].p;
}

extern "C" int nsfem_set_state(nsfem_ctx* ctx, int slot, const double* host, int64_t n) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && host, "null argument");
  NSFEM_REQUIRE(slot >= 0 && slot < NSFEM_N_SLOTS && n == slot_size(ctx, slot), "bad slot / size");
  NSFEM_HIP(hipMemcpyAsync(ctx->state[slot].p, host, sizeof(double) * n, hipMemcpyHostToDevice,
                           ctx->stream));
  NSFEM_HIP(hipStreamSynchronize(ctx->stream));
  if (slot == NSFEM_BODY_FORCE) ctx->have_body_force = true;
  if (slot == NSFEM_TRACTION) ctx->have_traction = true;
  if (slot == NSFEM_P || slot == NSFEM_P_OLD || slot == NSFEM_P2_OLD) ctx->pressure_history = 0;
  API_END(ctx)
}

extern "C" int nsfem_get_state(nsfem_ctx* ctx, int slot, double* host, int64_t n) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && host, "null argument");
  NSFEM_REQUIRE(slot >= 0 && slot < NSFEM_N_SLOTS && n == slot_size(ctx, slot), "bad slot / size");
  NSFEM_HIP(hipMemcpyAsync(host, ctx->state[slot].p, sizeof(double) * n, hipMemcpyDeviceToHost,
                           ctx->stream));
  NSFEM_HIP(hipStreamSynchronize(ctx->stream));
  API_END(ctx)
}

// ------------------------------------------------------------------ internals
// Mass-dominated velocity operators (small time steps): on a level where the stiffness part of
// the diagonal is at most `mg_trunc_ratio` times the mass part, a = ap M + b K is spectrally close
// to the (well conditioned) mass matrix, lambda_min(D^-1 a) >= lambda_min(D_M^-1 M) / (1 + r) with
// r = max_i b K_ii / (ap M_ii) and lambda_min(D_M^-1 M) = 1/2 for P1 elements (element bound) --
// a few Chebyshev steps SOLVE that level, and every coarser level, the dense coarse solve and (on
// partitioned meshes) the global coarse all-reduce drop out of the cycle.
static void select_velocity_cycle_depth(nsfem_ctx* c, double ap, double b) {
  Multigrid& mg = c->mg_v;
  mg.active = 0;
  if (!(c->mg_trunc_ratio > 0.0) || !(ap > 0.0) || mg.lv.size() < 3) return;
  hipStream_t s = c->stream;
  c->kw.ensure(nvel(c));
  double* parts = c->kw.parts.p;
  // level 1 = P1 on the fine mesh, level l + 2 = coarse[l]
  for (size_t l = 1; l + 1 < mg.lv.size(); ++l) {
    double r;
    if (l == 1) r = diag_ratio_max(s, c->p11, c->Mp.vals.p, c->Ap.vals.p, parts);
    else if (l - 2 < c->coarse.size())
      r = diag_ratio_max(s, c->coarse[l - 2]->pat, c->coarse[l - 2]->M.vals.p, c->coarse[l - 2]->K.vals.p, parts);
    else break;
    r *= b / ap;
    if (c->distributed()) {           // every rank must run the same cycle
      NSFEM_HIP(hipMemcpyAsync(parts, &r, sizeof(double), hipMemcpyHostToDevice, s));
      c->comm->allreduce_max(s, parts, 1);
      NSFEM_HIP(hipMemcpyAsync(&r, parts, sizeof(double), hipMemcpyDeviceToHost, s));
      NSFEM_HIP(hipStreamSynchronize(s));
    }
    if (r <= c->mg_trunc_ratio) {
      mg.active = l + 1;
      mg.trunc_lmin = 0.5 / (1.0 + r);
      mg.trunc_tol = c->mg_trunc_tol;
      return;
    }
  }
}

static void ensure_L(nsfem_ctx* c) {
  if (!c->L_dirty) return;
  // L = alpha0/k M + c_viscous K   (scalar P2; acts on all velocity components)
  const double a = c->alpha[0] / c->k, b = c->coef[2];
  launch_scale_combine(c->stream, c->p22.nnz, a, c->M2.vals.p, b, c->K2.vals.p, c->L.vals.p);
  if (!c->dict22_tried) {
    // stencil dictionary of the scalar P2 operators (lattice meshes only): shared by every
    // a M + b K; used by the smoothing steps of the velocity multigrid
    c->dict22_tried = true;
    if (build_stencil_dict(c->stream, c->p22, c->M2.vals.p, c->K2.vals.p, c->dict22)) {
      c->L.dict = &c->dict22;
      c->Lprec.dict = &c->dict22;
      if (c->dict22.exact) {       // the Chebyshev / CG mass solves may use it too
        c->M2.dict = &c->dict22;
        c->M2.sell_update(c->stream);
      }
    }
  }
  c->L.sell_update(c->stream);
  // preconditioner version: (alpha0/k + shift) M + c_viscous K -- the multigrid hierarchy of the
  // velocity block is built on it.  shift = 0 (all transient problems): the operator itself.
  // shift = 1/tau > 0 (stationary problems at high cell Peclet numbers): the V-cycle then
  // approximates (J + M/tau)^{-1}, the "time-step preconditioner" -- convection-dominated modes
  // are clustered at 1, only the slow diffusive modes are left to the Krylov method.
  const double ap = a + c->prec_shift;
  if (c->prec_shift != 0.0) {
    if (!c->Lprec.vals.p) c->Lprec.init(&c->p22, 1, 1, c->stream);
    launch_scale_combine(c->stream, c->p22.nnz, ap, c->M2.vals.p, b, c->K2.vals.p, c->Lprec.vals.p);
    c->Lprec.sell_update(c->stream);
  }
  if (c->mg_built) {
    c->mg_v.lv[0].A = c->prec_shift != 0.0 ? &c->Lprec : &c->L;
    launch_scale_combine(c->stream, c->p11.nnz, ap, c->Mp.vals.p, b, c->Ap.vals.p, c->Lc0.vals.p);
    if (!c->dict11_tried) {        // the fine P1 level of both hierarchies (smoothing steps)
      c->dict11_tried = true;
      if (build_stencil_dict(c->stream, c->p11, c->Mp.vals.p, c->Ap.vals.p, c->dict11)) {
        c->Lc0.dict = &c->dict11;
        c->Ap.dict = &c->dict11;
        c->Ap.sell_update(c->stream);
        c->Mp.dict = &c->dict11;
        c->Mp.sell_update(c->stream);
      }
    }
    c->Lc0.sell_update(c->stream);
    for (nsfem_ctx::P1Level* lv : c->coarse) {
      launch_scale_combine(c->stream, lv->pat.nnz, ap, lv->M.vals.p, b, lv->K.vals.p, lv->Lc.vals.p);
      lv->Lc.sell_update(c->stream);
    }
    if (c->global_coarse) {
      nsfem_ctx::P1Level* g = c->global_coarse;
      launch_scale_combine(c->stream, g->pat.nnz, ap, g->M.vals.p, b, g->K.vals.p, g->Lc.vals.p);
      g->Lc.sell_update(c->stream);
      for (nsfem_ctx::P1Level* t : c->global_tail) {
        launch_scale_combine(c->stream, t->pat.nnz, ap, t->M.vals.p, b, t->K.vals.p, t->Lc.vals.p);
        t->Lc.sell_update(c->stream);
      }
    }
    select_velocity_cycle_depth(c, ap, b);
    c->mg_v_dirty = true;
  }
  c->L_dirty = false;
}

// (re)build masks / smoother data / coarse inverse of a hierarchy when its inputs changed
static void mg_refresh_schur(nsfem_ctx* c) {
  auto ghost_flags = [&](std::vector<uint8_t>& m) {
    for (size_t i = 0; i < c->h_ghost_p1.size(); ++i)
      if (c->h_ghost_p1[i]) m[i] = 2;
  };
  if (c->mg_s_dirty) {
    c->graph_epoch++;
    std::vector<uint8_t> m((size_t)npre(c), 0);
    for (int32_t d : c->h_bc_s) m[d] = 1;
    ghost_flags(m);
    const bool singular = c->schur_singular >= 0 ? c->schur_singular != 0
                                                 : (c->distributed() ? true : c->h_bc_s.empty());
    c->mg_s.refresh(c->stream, m, singular);
    c->mg_s_dirty = false;
  }
  if (!c->mg_m.ready) {
    std::vector<uint8_t> m((size_t)npre(c), 0);
    ghost_flags(m);
    if (c->distributed()) {       // level 0 of the mass smoother needs its own device mask
      if (!c->mask_m.p) c->mask_m.alloc((size_t)npre(c));
      c->mask_m.upload(m, c->stream);
      c->mg_m.lv[0].mask = c->mask_m.p;
    }
    c->mg_m.refresh(c->stream, m, false);
  }
}

static void mg_refresh(nsfem_ctx* c, bool momentum) {
  if (momentum) {
    ensure_L(c);
    if (!c->mg_v_dirty) return;
    c->graph_epoch++;
    std::vector<uint8_t> m((size_t)nvel(c), 0);
    for (int32_t d : c->h_bc_v) m[d] = 1;
    const size_t dim = (size_t)c->mesh.dim;
    for (size_t i = 0; i < c->h_ghost_p2.size(); ++i)
      if (c->h_ghost_p2[i])
        for (size_t a = 0; a < dim; ++a) m[dim * i + a] = 2;
    c->mg_v.refresh(c->stream, m, false);
    c->mg_v_dirty = false;
  } else {
    if (!c->mg_p_dirty) return;
    c->graph_epoch++;
    std::vector<uint8_t> m((size_t)npre(c), 0);
    for (int32_t d : c->h_bc_p) m[d] = 1;
    for (size_t i = 0; i < c->h_ghost_p1.size(); ++i)
      if (c->h_ghost_p1[i]) m[i] = 2;
    // partitioned: whether the problem is singular is decided on the all-reduced mask
    c->mg_p.refresh(c->stream, m, c->distributed() ? true : c->h_bc_p.empty());
    c->mg_p_dirty = false;
  }
}

void nsfem_ctx::MomentumPrec::apply(hipStream_t s, const double* r, double* z) {
  // Newton rows of Dirichlet dofs are identity rows: the preconditioner must be too -- the last
  // finest-level smoothing step of the cycle writes z = r on them (Multigrid::identity_rows)
  c->mg_v.apply(s, r, z);
}

constexpr int P10 = 10;      // partial-sum slots 10, 11 of the Krylov work space belong to the drivers
static double cc_of(const nsfem_ctx* c) { return std::isfinite(c->coef[0]) ? c->coef[0] : 0.0; }
// Coriolis factor 2 c_cor omega (source/ns_solver_base.py:173-191): a scalar about e_z in 2D, the
// vector 2 c_cor Omega in 3D (coriolis_gamma3)
static bool coriolis_active(const nsfem_ctx* c) {
  if (c->mesh.dim == 2) return c->omega != 0.0;
  return c->omega3[0] != 0.0 || c->omega3[1] != 0.0 || c->omega3[2] != 0.0;
}
static double coriolis_gamma(const nsfem_ctx* c) {
  if (!coriolis_active(c)) return 0.0;
  if (!std::isfinite(c->coef[4])) throw Error(NSFEM_ERR_ARG, "angular velocity set but coriolis_term coefficient is None");
  return 2.0 * c->coef[4] * (c->mesh.dim == 2 ? c->omega : 1.0);
}
// y += (2 c_cor Omega x u, w): the mass matrix applied to the rotated field
static void coriolis_apply(nsfem_ctx* c, double g, const double* u, double* y,
                           const uint8_t* skipmask = nullptr) {
  hipStream_t s = c->stream;
  if (!c->rot_tmp.p) c->rot_tmp.alloc((size_t)nvel(c));
  if (c->mesh.dim == 2) {
    launch_rot90(s, c->mesh.n_p2, g, u, c->rot_tmp.p);
  } else {
    const double gv[3] = {g * c->omega3[0], g * c->omega3[1], g * c->omega3[2]};
    launch_cross3(s, c->mesh.n_p2, gv, u, c->rot_tmp.p);
  }
  launch_spmv_axpy(s, c->M2, c->mesh.dim, 1.0, c->rot_tmp.p, y, skipmask);
}

// lattice meshes: dictionary copies of the rectangular divergence / gradient blocks (the monolithic operator uses
// them in every Krylov iteration; the pressure-correction scheme once per step each -- there only dictionaries that
// equal the assembled blocks bit for bit serve, see spmv_dispatch)
static void ensure_div_dicts(nsfem_ctx* ctx) {
  if (ctx->dictD_tried) return;
  ctx->dictD_tried = true;
  hipStream_t s = ctx->stream;
  const int dim = ctx->mesh.dim;
  if (build_stencil_dict(s, ctx->p21, ctx->DT.vals.p, nullptr, ctx->dict21, dim, true)) {
    ctx->DT.dict = &ctx->dict21;
    ctx->DT.sell_update(s);
  }
  if (build_stencil_dict(s, ctx->p12, ctx->Dv.vals.p, nullptr, ctx->dict12, dim, true)) {
    ctx->Dv.dict = &ctx->dict12;
    ctx->Dv.sell_update(s);
  }
  if (build_stencil_dict(s, ctx->p21, ctx->Gr.vals.p, nullptr, ctx->dict21g, dim, true)) {
    ctx->Gr.dict = &ctx->dict21g;
    ctx->Gr.sell_update(s);
  }
}

// time-step constant part of the momentum residual:
//   g = M (a1 u1 + a2 u2) / k - c_p (p_old, div w) - c_b M f + traction
static void momentum_begin_step(nsfem_ctx* c, bool with_old_pressure = true) {
  hipStream_t s = c->stream;
  const int64_t nv = nvel(c);
  ensure_L(c);
  ensure_div_dicts(c);
  const double a1 = c->alpha[1] / c->k, a2 = c->alpha[2] / c->k;
  if (c->have_body_force) {
    NSFEM_REQUIRE(std::isfinite(c->coef[3]), "body force set but body_force_term coefficient is None");
    launch_lincomb3(s, nv, a1, c->state[NSFEM_U1].p, a2, c->state[NSFEM_U2].p, -c->coef[3],
                    c->state[NSFEM_BODY_FORCE].p, c->tmp_v.p);
  } else {
    launch_axpby(s, nv, a1, c->state[NSFEM_U1].p, a2, c->state[NSFEM_U2].p, c->tmp_v.p);
  }
  const bool euler = c->mesh.dim == 2 ? c->omega_dot != 0.0
                                      : (c->omega_dot3[0] != 0.0 || c->omega_dot3[1] != 0.0 || c->omega_dot3[2] != 0.0);
  if (euler) {      // Euler acceleration  c_e (d Omega/dt) x x  (ns_solver_base.py:193-211)
    NSFEM_REQUIRE(std::isfinite(c->coef[5]), "angular acceleration set but euler_term coefficient is None");
    if (!c->rot_field.p) {
      c->rot_field.alloc((size_t)nv);
      if (c->mesh.dim == 2) launch_rot_field(s, c->mesh, c->rot_field.p);       // e_z x x
      else launch_coord_field_3d(s, c->mesh, c->rot_field.p);                    // x
    }
    if (c->mesh.dim == 2) {
      launch_axpby(s, nv, 1.0, c->tmp_v.p, c->coef[5] * c->omega_dot, c->rot_field.p, c->tmp_v.p);
    } else {
      if (!c->rot_tmp.p) c->rot_tmp.alloc((size_t)nv);
      const double a[3] = {c->coef[5] * c->omega_dot3[0], c->coef[5] * c->omega_dot3[1],
                           c->coef[5] * c->omega_dot3[2]};
      launch_cross3(s, c->mesh.n_p2, a, c->rot_field.p, c->rot_tmp.p);
      launch_axpby(s, nv, 1.0, c->tmp_v.p, 1.0, c->rot_tmp.p, c->tmp_v.p);
    }
  }
  launch_spmv(s, c->M2, c->mesh.dim, c->tmp_v.p, c->gconst.p, nullptr, MASK_NONE);
  if (with_old_pressure)   // IPCS: - c_p (p_old, div w); the monolithic scheme keeps p unknown
    launch_spmv_axpy(s, c->DT, 1, -c->coef[1], c->state[NSFEM_P_OLD].p, c->gconst.p, nullptr);
  if (c->have_traction)
    launch_axpby(s, nv, 1.0, c->gconst.p, 1.0, c->state[NSFEM_TRACTION].p, c->gconst.p);
}

// b = L u* [+ c_v E u*] + g + c_c conv(u*) ;  Dirichlet rows: u*_i - g_i ;  returns |b|
// |x|_2 over owned entries (ghost entries are zeroed first), all-reduced over the ranks
static double global_norm(nsfem_ctx* c, int64_t n, double* x, const uint8_t* ghostmask) {
  hipStream_t s = c->stream;
  if (ghostmask) launch_zero_ghost(s, n, ghostmask, x);
  double* parts = c->kw.parts.p + (size_t)10 * kParts;
  launch_dot(s, n, x, x, parts);
  if (c->distributed()) c->comm->allreduce_sum(s, parts, kParts);
  return std::sqrt(host_sum_parts(s, c->kw, 10));
}

static void fill_linop(nsfem_ctx* c, LinOp& op, bool velocity) {
  op.graph_epoch = c->graph_epoch;
  if (!c->distributed()) return;
  op.comm = c->comm;
  op.halo = velocity ? &c->halo_p2 : &c->halo_p1;
  op.halo_width = velocity ? c->mesh.dim : 1;
  op.ghostmask = velocity ? c->mask_v.p : c->mask_p.p;
  op.n_global = velocity ? 2 * c->n_p2_global : c->n_p1_global;
}

static int jacobian_path(nsfem_ctx* c);
static void momentum_residual_raw(nsfem_ctx* c, const double* u, double* out) {
  hipStream_t s = c->stream;
  const int64_t nv = nvel(c);
  // lattice meshes (same conditions as the one-launch Jacobian action): product, + g, element kernel and node sums
  // in one launch of k_jac_lattice<FORM, 0>
  if (jacobian_path(c) == 2 &&
      launch_residual_lattice(s, c->mesh, c->L, u, c->gconst.p, cc_of(c), c->conv_form, out)) {
    ++c->jac_lattice_launches;
    return;
  }
  launch_spmv(s, c->L, c->mesh.dim, u, out, nullptr, MASK_NONE);
  launch_axpby(s, nv, 1.0, out, 1.0, c->gconst.p, out);
  if (c->traction_form) launch_spmv_axpy(s, c->E, 1, c->coef[2], u, out, nullptr);
  const double cc = cc_of(c);
  if (cc != 0.0) launch_convection_residual(s, c->mesh, u, cc, out, c->conv_form);
  const double g = coriolis_gamma(c);
  if (g != 0.0) coriolis_apply(c, g, u, out);
}

static double momentum_residual(nsfem_ctx* c) {
  hipStream_t s = c->stream;
  const int64_t nv = nvel(c);
  double* u = c->state[NSFEM_USTAR].p;
  momentum_residual_raw(c, u, c->rhs_v.p);
  if (!c->distributed() && c->mask_v.p) {      // Dirichlet rows and the norm in one launch (mask != 0 <=> bc dof)
    double* parts = c->kw.parts.p + (size_t)10 * kParts;
    launch_bc_residual_norm(s, nv, c->rhs_v.p, c->mask_v.p, c->nbc_v, c->bc_v_dofs.p, c->bc_v_vals.p, u, parts);
    return std::sqrt(host_sum_parts(s, c->kw, 10));
  }
  launch_set_bc_residual(s, c->nbc_v, c->bc_v_dofs.p, c->bc_v_vals.p, u, c->rhs_v.p);
  return global_norm(c, nv, c->rhs_v.p, c->ghost_v.p ? c->mask_v.p : nullptr);
}

static void momentum_jacobian(nsfem_ctx* c, int vel_slot = NSFEM_USTAR) {
  hipStream_t s = c->stream;
  const double* E = c->traction_form ? c->E.vals.p : nullptr;
  const double cc = cc_of(c);
  if (cc != 0.0)
    launch_convection_jacobian(s, c->mesh, c->p22, c->state[vel_slot].p, cc, c->L.vals.p, E,
                               c->coef[2], c->J.vals.p, c->conv_form, c->picard);
  else
    if (c->mesh.dim == 3) jacobian_init_3d(s, c->p22.nnz, c->L.vals.p, E, c->coef[2], c->J.vals.p);
    else launch_jacobian_init(s, c->p22.nnz, c->L.vals.p, E, c->coef[2], c->J.vals.p);
  const double g = coriolis_gamma(c);
  if (g != 0.0) {
    if (c->mesh.dim == 2) {
      launch_jac_add_skew(s, c->p22.nnz, g, c->M2.vals.p, c->J.vals.p);
    } else {
      const double gv[3] = {g * c->omega3[0], g * c->omega3[1], g * c->omega3[2]};
      launch_jac_add_skew3(s, c->p22.nnz, gv, c->M2.vals.p, c->J.vals.p);
    }
  }
  launch_inv_diag(s, c->J, 1, c->mask_v.p, c->dinv_v.p);
}

// how the matrix-free Jacobian action runs: 0 = L product, element kernel, node gather; 1 = element kernel, then
// the L product sums its node-sorted vectors (one GPU, triangles, stencil dictionary); 2 = k_jac_lattice, everything
// in one launch (lattice mesh in rectangle_mesh numbering, gradient-form viscosity, no rotating frame)
// Partitioned strips (round 4): the local mesh of a rank is a lattice of its own (own cell rows + the ghost row), so
// path 2 runs there as well -- after ONE halo exchange of the input; rows of ghost nodes come out as zeros (mask
// value 2).  Path 1 stays single-context (its node-sorted buffer has no interior / halo split).
static int jacobian_path(nsfem_ctx* c) {
  if (c->mesh.dim != 2 || cc_of(c) == 0.0 || !c->L.dict_ready) return 0;
  if (c->distributed() && !partitioned_lattice_kernels()) return 0;
  if (!c->mesh.cl.tried && c->L.dict && c->L.dict->lat_w > 0) {
    build_cell_lattice(c->h_p2map.data(), c->mesh.n_cells, c->L.dict->lat_w, c->L.dict->lat_h, c->mesh.cl);
    check_uniform_geometry(c->stream, c->mesh);
  }
  if (!c->traction_form && coriolis_gamma(c) == 0.0 && c->mask_v.p && jacobian_lattice_available(c->mesh, c->L))
    return 2;
  return c->distributed() ? 0 : 1;
}

void nsfem_ctx::MomentumMF::apply(hipStream_t s, const double* x, double* y) {
  const int64_t nv = nvel(c);
  const int dim = c->mesh.dim;
  // (partitioned: the halo exchange of x runs under the interior rows of the L product; the
  // element kernel of the convection action below reads the ghost nodes and comes after it)
  // identity rows on the Dirichlet dofs and zero ghost rows come out of the L product itself
  // (row mask, MASK_IDENTITY); every later contribution leaves the flagged rows untouched
  (void)nv;
  const double cc = cc_of(c);
  // one GPU, triangles, lattice mesh: the element kernel of the convection action first, then ONE launch of the
  // dictionary kernel forms L x AND sums the node-sorted element vectors (no separate node gather, y is written
  // once instead of written, read and written again)
  const int path = (dim == 2 && cc != 0.0 && c->L.dict_ready) ? jacobian_path(c) : 0;
  if (path != 0) {
    nsfem_ctx::Probe& pr = c->conv_probe;
    const bool timed = pr.on && pr.n + 2 <= pr.ev.size();
    // lattice meshes: element kernel, node sums and L product in one launch (k_jac_lattice)
    if (path == 2) {
      // strips: the ghost lines of the input first -- under the tile rows that read none of them when the overlap
      // mode is on and the strip is high enough; the kernel writes zeros to the ghost rows
      if (c->distributed()) {
        if (c->p2_gh_lo == -2) {
          int lo, hi;
          const bool lines = c->L.dict && ghost_lattice_lines(c->h_ghost_p2, c->L.dict->lat_w, c->L.dict->lat_h, lo, hi);
          c->p2_gh_lo = lines ? lo : -1;
          c->p2_gh_hi = lines ? hi : -1;
        }
        if (c->comm->overlap && c->p2_gh_lo >= 0 && jacobian_lattice_split(c->mesh, c->p2_gh_lo, c->p2_gh_hi)) {
          c->comm->exchange_begin(s, c->halo_p2, const_cast<double*>(x), dim);
          bool ok = launch_jacobian_lattice(s, c->mesh, c->L, c->state[vel_slot].p, x, cc, c->conv_form, c->picard,
                                            c->mask_v.p, y, 1, c->p2_gh_lo, c->p2_gh_hi);
          c->comm->exchange_end(s);
          ok = ok && launch_jacobian_lattice(s, c->mesh, c->L, c->state[vel_slot].p, x, cc, c->conv_form, c->picard,
                                             c->mask_v.p, y, 2, c->p2_gh_lo, c->p2_gh_hi);
          NSFEM_REQUIRE(ok, "one-launch Jacobian action: tile-row split failed");
          ++c->jac_lattice_launches;
          return;
        }
        c->comm->exchange(s, c->halo_p2, const_cast<double*>(x), dim);
      }
      if (timed) NSFEM_HIP(hipEventRecord(pr.ev[pr.n], s));
      if (launch_jacobian_lattice(s, c->mesh, c->L, c->state[vel_slot].p, x, cc, c->conv_form, c->picard,
                                  c->mask_v.p, y)) {
        ++c->jac_lattice_launches;
        if (timed) {
          NSFEM_HIP(hipEventRecord(pr.ev[pr.n + 1], s));
          pr.n += 2;
        }
        return;
      }
    }
    if (!c->distributed()) {
      if (timed) NSFEM_HIP(hipEventRecord(pr.ev[pr.n], s));
      launch_convection_cells(s, c->mesh, c->state[vel_slot].p, x, cc, c->conv_form, c->picard);
      if (timed) {
        NSFEM_HIP(hipEventRecord(pr.ev[pr.n + 1], s));
        pr.n += 2;
      }
      if (launch_spmv_with_gather(s, c->L, dim, x, y, c->mask_v.p, MASK_IDENTITY, c->mesh.nptr.p, c->mesh.rbuf.p)) {
        if (c->traction_form) launch_spmv_axpy(s, c->E, 1, c->coef[2], x, y, c->mask_v.p);
        const double g = coriolis_gamma(c);
        if (g != 0.0) coriolis_apply(c, g, x, y, c->mask_v.p);
        return;
      }
      // (no dictionary kernel: product, then the gather below re-runs the cheap element kernel)
    }
  }
  product_with_halo(c->distributed() ? c->comm : nullptr, &c->halo_p2, dim, s, x, c->L.pat,
                    [&](int phase) { launch_spmv(s, c->L, dim, x, y, c->mask_v.p, MASK_IDENTITY, 0, phase, 1); });
  if (c->traction_form) launch_spmv_axpy(s, c->E, 1, c->coef[2], x, y, c->mask_v.p);
  if (cc != 0.0) {
    nsfem_ctx::Probe& pr = c->conv_probe;
    const bool timed = pr.on && pr.n + 2 <= pr.ev.size();
    if (timed) NSFEM_HIP(hipEventRecord(pr.ev[pr.n], s));
    launch_convection_action(s, c->mesh, c->state[vel_slot].p, x, cc, y, c->conv_form, c->picard,
                             c->mask_v.p);
    if (timed) {
      NSFEM_HIP(hipEventRecord(pr.ev[pr.n + 1], s));
      pr.n += 2;
    }
  }
  const double g = coriolis_gamma(c);
  if (g != 0.0) coriolis_apply(c, g, x, y, c->mask_v.p);
}

// 0 = auto = matrix-free: inside the fused step drivers the velocity Jacobian is applied as
// L x + c_c [d conv(u)/du] x by an element kernel (one thread per cell) instead of being
// assembled.  Measured: tetrahedra, n = 48: step 55 -> 29 ms (assembling the 3x3-block matrix cost
// more than the handful of products an inexact Newton step needs); triangles, n = 512:
// 9.3 -> 8.6 ms with inexact Newton, equal (16 ms) with rtol 1e-12 + exact Newton.
// 1 = always assemble (block CSR), 2 = always matrix-free.
static bool use_matrix_free(const nsfem_ctx* c, const nsfem_step_opts* o) {
  (void)c;
  return o->matrix_free != 1;
}

// J dx = b ; u* -= dx
// known_rhs_norm >= 0: |rhs_v|_2 as the caller has just evaluated it (the Newton residual norm): the solve starts
// without reading its start-up sums back
static int momentum_solve_update(nsfem_ctx* c, const nsfem_krylov_opts& o, nsfem_solve_info& info,
                                 double known_rhs_norm = -1.0) {
  hipStream_t s = c->stream;
  LinOp op;
  op.A = &c->J;
  op.nv = 1;
  op.rowmask = c->mask_v.p;
  op.maskmode = MASK_IDENTITY;
  op.dinv = c->dinv_v.p;
  fill_linop(c, op, true);
  if (c->mf_active) {               // matrix-free Jacobian (the step driver skipped the assembly)
    c->mom_mf.c = c;
    c->mom_mf.n = nvel(c);
    c->mom_mf.vel_slot = NSFEM_USTAR;
    op.custom = &c->mom_mf;
    if (o.precond != 1) launch_inv_diag(s, c->L, c->mesh.dim, c->mask_v.p, c->dinv_v.p);
  }
  if (o.precond == 1) {
    NSFEM_REQUIRE(c->mg_built, "multigrid requested but no hierarchy was set (nsfem_mg_finalize)");
    mg_refresh(c, true);
    op.prec = &c->mom_prec;
  }
  op.graph_epoch = c->graph_epoch;
  op.x_zero = true;                 // (dx_v is not read: the first update of the solve writes it)
  op.known_bnorm = known_rhs_norm;
  int rc = bicgstab(s, c->kw, op, c->rhs_v.p, c->dx_v.p, o, info);
  if (rc != NSFEM_OK) return rc;
  double* u = c->state[NSFEM_USTAR].p;
  launch_axpby(s, nvel(c), 1.0, u, -1.0, c->dx_v.p, u);
  return NSFEM_OK;
}

// rhs = A_p p_old - alpha0/k D u* ; start vector p = p_old with Dirichlet values
static void poisson_assemble(nsfem_ctx* c, bool extrapolate = false) {
  hipStream_t s = c->stream;
  const int64_t np = npre(c);
  launch_spmv(s, c->Ap, 1, c->state[NSFEM_P_OLD].p, c->rhs_p.p, nullptr, MASK_NONE);
  launch_spmv(s, c->Dv, 1, c->state[NSFEM_USTAR].p, c->tmp_p.p, nullptr, MASK_NONE);
  launch_axpby(s, np, 1.0, c->rhs_p.p, -c->alpha[0] / c->k, c->tmp_p.p, c->rhs_p.p);
  if (c->ghost_p.p) launch_zero_ghost(s, np, c->mask_p.p, c->rhs_p.p);
  if (extrapolate && c->pressure_history >= 2)       // start vector 2 p_n - p_(n-1)
    launch_axpby(s, np, 2.0, c->state[NSFEM_P_OLD].p, -1.0, c->state[NSFEM_P2_OLD].p, c->state[NSFEM_P].p);
  else
    NSFEM_HIP(hipMemcpyAsync(c->state[NSFEM_P].p, c->state[NSFEM_P_OLD].p, sizeof(double) * np,
                             hipMemcpyDeviceToDevice, s));
  launch_set_values(s, c->nbc_p, c->bc_p_dofs.p, c->bc_p_vals.p, c->state[NSFEM_P].p);
  if (!c->dinv_p_ready) {
    launch_inv_diag(s, c->Ap, 1, c->mask_p.p, c->dinv_p.p);
    c->dinv_p_ready = true;
  }
}

// is the pressure pinned by a Dirichlet value ANYWHERE?  On partitioned meshes a rank may hold
// none of the outlet nodes (unstructured partitions): all ranks must take the same branch in the
// solver (mean projection of the singular problem = an all-reduce), so the flag is all-reduced
// once after every nsfem_set_dirichlet(PRESSURE) -- which every rank calls, if only with n = 0.
static bool pressure_pinned_anywhere(nsfem_ctx* c) {
  if (!c->distributed()) return c->nbc_p > 0;
  if (c->bc_p_any < 0) {
    hipStream_t s = c->stream;
    c->kw.ensure(nvel(c));
    double* parts = c->kw.parts.p;
    double v = c->nbc_p > 0 ? 1.0 : 0.0;
    NSFEM_HIP(hipMemcpyAsync(parts, &v, sizeof(double), hipMemcpyHostToDevice, s));
    c->comm->allreduce_max(s, parts, 1);
    NSFEM_HIP(hipMemcpyAsync(&v, parts, sizeof(double), hipMemcpyDeviceToHost, s));
    NSFEM_HIP(hipStreamSynchronize(s));
    c->bc_p_any = v > 0.5 ? 1 : 0;
  }
  return c->bc_p_any > 0;
}

// Projection step by fast diagonalisation (precond = 3; tensor-product lattices, Dirichlet conditions on whole sides):
// x += A^+ (b - A x), then the residual is checked -- the direct solve is exact up to round-off, so ONE pass and one
// device -> host round trip; further passes (iterative refinement) only if the check fails.
static int poisson_solve_fast_diag(nsfem_ctx* c, const nsfem_krylov_opts& o, nsfem_solve_info& info) {
  hipStream_t s = c->stream;
  const int64_t np = npre(c);
  NSFEM_REQUIRE(c->fd_p.ready() && (int64_t)c->fd_p.W * c->fd_p.H == np,
                "fast diagonalisation requested but no factors were set (nsfem_poisson_set_fast_diag)");
  NSFEM_REQUIRE(!c->distributed(), "fast diagonalisation: one rank only");
  KrylovWork& w = c->kw;
  w.ensure(std::max<int64_t>(np, nvel(c)));
  ++w.touch;
  double* parts = w.parts.p;
  double* x = c->state[NSFEM_P].p;
  const double* rhs = c->rhs_p.p;
  if (!pressure_pinned_anywhere(c)) {
    // singular (all-Neumann) operator: make the right-hand side compatible, as the CG path does
    NSFEM_HIP(hipMemcpyAsync(w.t.p, rhs, sizeof(double) * np, hipMemcpyDeviceToDevice, s));
    launch_sum_sub_mean(s, np, w.t.p, parts + 10 * kParts);
    rhs = w.t.p;
  }
  constexpr int PR = 10, PB0 = 13;
  launch_dot(s, np, rhs, rhs, parts + PB0 * kParts);
  info.iterations = 0;
  info.converged = 0;
  double bnorm = 0.0, target = 0.0;
  launch_residual(s, c->Ap, 1, x, rhs, w.r.p, c->mask_p.p, MASK_ZERO);
  for (int pass = 0; pass < std::max(1, std::min(o.max_iter, 8)); ++pass) {
    c->fd_p.apply(s, w.r.p, w.z.p);
    launch_axpby(s, np, 1.0, x, 1.0, w.z.p, x);
    ++info.iterations;
    // the residual of the corrected iterate (the next pass's right-hand side)
    launch_residual(s, c->Ap, 1, x, rhs, w.r.p, c->mask_p.p, MASK_ZERO);
    launch_dot(s, np, w.r.p, w.r.p, parts + PR * kParts);
    double rr, bb;
    host_sum_parts2(s, w, PR, PB0, rr, bb);
    bnorm = std::sqrt(bb);
    target = std::max(o.atol, o.rtol * (bnorm > 0.0 ? bnorm : 1.0));
    w.last_target = target;
    if (!std::isfinite(rr)) return NSFEM_ERR_BREAKDOWN;
    info.residual = std::sqrt(rr);
    if (pass == 0) info.residual0 = info.residual;
    if (info.residual <= target) { info.converged = 1; break; }
  }
  return info.converged ? NSFEM_OK : NSFEM_ERR_NOT_CONVERGED;
}

// The whole projection step of the fused driver by fast diagonalisation, for the pure Neumann problem (no pressure
// Dirichlet dofs, one rank): with the start vector p_old the residual of  A p = A p_old - (alpha0 / k) D u*  is
// -(alpha0 / k) D u*  itself -- no product with A to form a right-hand side, none for a start residual:
//   r = -(alpha0 / k) D u* - mean ;  z = A^+ r ;  p = p_old + z ;  check |r - A z| <= max(atol, rtol |r|)
// (the same p as the assembled system gives, up to the constant and round-off; |r| <= |rhs| makes the check the
// stricter one).  Further passes refine (never needed so far: the direct solve leaves ~1e-14 |r|).
static int poisson_direct_step(nsfem_ctx* c, const nsfem_krylov_opts& o, nsfem_solve_info& info) {
  hipStream_t s = c->stream;
  const int64_t np = npre(c);
  KrylovWork& w = c->kw;
  w.ensure(std::max<int64_t>(np, nvel(c)));
  ++w.touch;
  double* parts = w.parts.p;
  constexpr int PR = 10, PB0 = 13;
  // Partitioned strips: the same solve with ONE collective (FastDiag::apply_strip).  u* carries valid ghost values
  // (every Krylov solve fills the ghosts of its solution), so D u* is complete on the owned rows; ghost rows are
  // zeroed (every node counts once in the sums over the ranks); z comes back on every local row, ghost lines
  // included, so p = p_old + z needs no halo exchange and the check r - A z none either.
  const bool dist = c->distributed();
  const uint8_t* gm = dist ? c->mask_p.p : nullptr;                            // (flag 2 on ghost rows)
  NSFEM_REQUIRE(!dist || (c->fd_p.strip() && gm), "fast diagonalisation on a partitioned mesh: strip factors not set");
  launch_spmv_scaled(s, c->Dv, 1, -c->alpha[0] / c->k, c->state[NSFEM_USTAR].p, w.r.p);
  if (dist) {
    launch_zero_ghost(s, np, gm, w.r.p);
    launch_sum(s, np, w.r.p, parts + PR * kParts);
    c->comm->allreduce_sum(s, parts + PR * kParts, kParts);
    launch_sub_mean(s, np, c->n_p1_global, parts + PR * kParts, w.r.p);
    launch_zero_ghost(s, np, gm, w.r.p);
  } else {
    launch_sum_sub_mean(s, np, w.r.p, parts + PR * kParts);
  }
  launch_dot(s, np, w.r.p, w.r.p, parts + PB0 * kParts);
  const double* base = c->state[NSFEM_P_OLD].p;
  info.iterations = 0;
  info.converged = 0;
  for (int pass = 0; pass < std::max(1, std::min(o.max_iter, 8)); ++pass) {
    if (dist) c->fd_p.apply_strip(s, c->comm, w.r.p, w.z.p);
    else c->fd_p.apply(s, w.r.p, w.z.p);
    launch_axpby(s, np, 1.0, base, 1.0, w.z.p, c->state[NSFEM_P].p);          // p = p_old + z (later passes: p += z)
    base = c->state[NSFEM_P].p;
    ++info.iterations;
    launch_residual(s, c->Ap, 1, w.z.p, w.r.p, w.q.p, gm, gm ? MASK_ZERO : MASK_NONE);    // q = r - A z
    launch_dot(s, np, w.q.p, w.q.p, parts + PR * kParts);
    if (dist) c->comm->allreduce_sum(s, parts + PR * kParts, (PB0 - PR + 1) * kParts);    // (slots 10 ... 13)
    double qq, rr;
    host_sum_parts2(s, w, PR, PB0, qq, rr);
    if (!std::isfinite(qq)) return NSFEM_ERR_BREAKDOWN;
    const double rnorm = std::sqrt(rr);
    const double target = std::max(o.atol, o.rtol * (rnorm > 0.0 ? rnorm : 1.0));
    w.last_target = target;
    info.residual = std::sqrt(qq);
    if (pass == 0) info.residual0 = rnorm;
    if (info.residual <= target) { info.converged = 1; break; }
    std::swap(w.r.p, w.q.p);                                                   // refine: the residual is the next right-hand side
  }
  return info.converged ? NSFEM_OK : NSFEM_ERR_NOT_CONVERGED;
}

static int poisson_solve(nsfem_ctx* c, const nsfem_krylov_opts& o, nsfem_solve_info& info) {
  if (o.precond == 3) return poisson_solve_fast_diag(c, o, info);
  LinOp op;
  op.A = &c->Ap;
  op.nv = 1;
  op.rowmask = c->mask_p.p;
  op.maskmode = MASK_ZERO;
  op.dinv = c->dinv_p.p;
  fill_linop(c, op, false);
  if (o.precond == 1) {
    NSFEM_REQUIRE(c->mg_built, "multigrid requested but no hierarchy was set (nsfem_mg_finalize)");
    mg_refresh(c, false);
    op.prec = &c->mg_p;
  }
  op.graph_epoch = c->graph_epoch;
  return pcg(c->stream, c->kw, op, c->rhs_p.p, c->state[NSFEM_P].p, o, info, !pressure_pinned_anywhere(c));
}

// rhs = M u* - k/alpha0 G (p - p_old) ; start vector u0 = u* with Dirichlet values
// with_start_residual (fused step, one rank, Chebyshev mass solve): the start residual of the solve and the two sums
// its convergence check needs come out of the same pass as the right-hand side (k_correction_setup): r0 in kw.r,
// |r0|^2 and |rhs|^2 in the partial-sum slots 12, 13
static void correction_assemble(nsfem_ctx* c, bool with_start_residual = false) {
  hipStream_t s = c->stream;
  const int64_t nv = nvel(c), np = npre(c);
  launch_spmv(s, c->M2, c->mesh.dim, c->state[NSFEM_USTAR].p, c->rhs_v.p, nullptr, MASK_NONE);
  launch_axpby(s, np, 1.0, c->state[NSFEM_P].p, -1.0, c->state[NSFEM_P_OLD].p, c->tmp_p.p);
  launch_spmv(s, c->Gr, 1, c->tmp_p.p, c->tmp_v.p, nullptr, MASK_NONE);
  c->cor_start_ready = false;
  if (with_start_residual && !c->distributed() && c->mask_v.p) {
    c->kw.ensure(nv);
    launch_correction_setup(s, nv, -c->k / c->alpha[0], c->rhs_v.p, c->tmp_v.p, c->mask_v.p, c->kw.r.p,
                            c->kw.parts.p + 12 * kParts, c->kw.parts.p + 13 * kParts);
    c->cor_start_ready = true;
    c->cor_start_touch = ++c->kw.touch;
  } else {
    launch_axpby(s, nv, 1.0, c->rhs_v.p, -c->k / c->alpha[0], c->tmp_v.p, c->rhs_v.p);
  }
  if (c->ghost_v.p) launch_zero_ghost(s, nv, c->mask_v.p, c->rhs_v.p);
  NSFEM_HIP(hipMemcpyAsync(c->state[NSFEM_U0].p, c->state[NSFEM_USTAR].p, sizeof(double) * nv,
                           hipMemcpyDeviceToDevice, s));
  launch_set_values(s, c->nbc_v, c->bc_v_dofs.p, c->bc_v_vals.p, c->state[NSFEM_U0].p);
  if (!c->dinv_m_ready) {
    ++c->mass_dinv_epoch;
    launch_inv_diag(s, c->M2, c->mesh.dim, c->mask_v.p, c->dinv_m.p);
    c->dinv_m_ready = true;
  }
}

// Chebyshev iteration on the Jacobi-scaled velocity mass matrix with A-PRIORI spectral bounds
// (Wathen's element bounds, p2_mass_jacobi_bounds): no dot products, every iteration is ONE
// launch of the fused smoother kernel, and on partitioned meshes no all-reduce at all inside
// the iteration.  Solves the correction equation  M e = b - M x0  (e vanishes on the Dirichlet
// dofs), then x0 += e; the residual norm is checked after the predicted number of steps.
static int correction_solve_chebyshev(nsfem_ctx* c, const nsfem_krylov_opts& o, nsfem_solve_info& info) {
  hipStream_t s = c->stream;
  const int64_t nv = nvel(c);
  Multigrid& mg = c->mg_mv;
  if (mg.lv.empty()) {
    double lmin, lmax;
    p2_mass_jacobi_bounds(c->mesh.dim, lmin, lmax);
    mg.nv = c->mesh.dim;
    mg.coarse_dense_max = 0;
    mg.smoother_only = true;
    mg.lv.resize(1);
    mg.lv[0].A = &c->M2;
    mg.lv[0].n = c->mesh.n_p2;
    mg.lv[0].mask = c->mask_v.p;
    if (c->distributed()) {
      mg.comm = c->comm;
      mg.lv[0].halo = c->halo_p2;
      mg.lv[0].has_halo = true;
    }
    mg.setup_work(s);
    c->mass_dinv_copied_epoch = -1;            // (fresh work vectors)
    // interval [0.98 lmin, 1.02 lmax] in the parametrisation of Multigrid::cheb_coeffs
    c->mass_kappa = (1.02 * lmax) / (0.98 * lmin);
    mg.eig_ratio = c->mass_kappa;
    mg.lv[0].lmax = 1.02 * lmax / 1.05;
  }
  MGLevel& L = mg.lv[0];
  if (c->mass_dinv_copied_epoch != c->mass_dinv_epoch) {       // (once per change of the Dirichlet set, not per solve)
    NSFEM_HIP(hipMemcpyAsync(L.dinv.p, c->dinv_m.p, sizeof(double) * nv, hipMemcpyDeviceToDevice, s));
    c->mass_dinv_copied_epoch = c->mass_dinv_epoch;
  }
  KrylovWork& w = c->kw;
  w.ensure(nv);
  double* x = c->state[NSFEM_U0].p;
  double* parts = w.parts.p;
  auto residual_norm = [&](double* r, int slot = P10) {
    if (c->distributed()) c->comm->exchange(s, c->halo_p2, x, c->mesh.dim);
    launch_residual(s, c->M2, c->mesh.dim, x, c->rhs_v.p, r, c->mask_v.p, MASK_ZERO);
    launch_dot(s, nv, r, r, parts + slot * kParts);
  };
  // The a-priori step count is a worst-case bound; the same solve one time step ago tells what was actually needed
  // (first_check = its count, reduced by next_hint's linear-convergence estimate when it overshot): the first
  // sequence runs that many steps, the bound takes over if the residual check then fails.  With such a prediction
  // (one rank) the start-up sums |r0|^2, |b|^2 are not read back before the sequence: they wait in their own slots
  // and come with the residual check after it -- one device -> host round trip per solve instead of two.
  int k_hint = o.first_check >= 1 ? o.first_check : 0;
  // (correction_assemble left r0 in w.r and the sums in slots 12, 13 -- and no other solver has run since)
  const bool start_ready = c->cor_start_ready && c->cor_start_touch == w.touch;
  c->cor_start_ready = false;
  ++w.touch;
  const bool deferred = (k_hint > 0 || start_ready) && !c->distributed();
  constexpr int PR0 = 12, PB0 = 13;
  if (!start_ready) {
    residual_norm(w.r.p, deferred ? PR0 : P10);
    launch_dot(s, nv, c->rhs_v.p, c->rhs_v.p, parts + (deferred ? PB0 : P10 + 1) * kParts);
  }
  if (c->distributed()) c->comm->allreduce_sum(s, parts + P10 * kParts, 2 * kParts);
  double rr0 = 0.0, bb0 = 0.0, r0 = 0.0, bnorm = 0.0, target = 0.0;
  if (!deferred) {
    host_sum_parts2(s, w, P10, P10 + 1, rr0, bb0);
    r0 = std::sqrt(rr0);
    bnorm = std::sqrt(bb0);
    target = std::max(o.atol, o.rtol * (bnorm > 0.0 ? bnorm : 1.0));
    w.last_target = target;
  }
  info.residual0 = info.residual = r0;
  info.iterations = 0;
  info.converged = deferred ? false : r0 <= target;
  const double sk = std::sqrt(c->mass_kappa);
  const double rate = std::log((sk + 1.0) / (sk - 1.0));
  double res = r0;
  bool first_pass = true;
  while (!info.converged && info.iterations < o.max_iter) {
    // steps needed for the error bound 2 sqrt(kappa) ((sqrt(kappa)-1)/(sqrt(kappa)+1))^k <= target / res
    int k = (deferred && first_pass && k_hint > 0) ? k_hint : 0;
    if (k == 0 && deferred && first_pass) {
      // no prediction yet (first step): the start sums are read now after all
      double rr_, bb_;
      host_sum_parts2(s, w, PR0, PB0, rr_, bb_);
      r0 = res = std::sqrt(rr_);
      bnorm = std::sqrt(bb_);
      target = std::max(o.atol, o.rtol * (bnorm > 0.0 ? bnorm : 1.0));
      w.last_target = target;
      info.residual0 = info.residual = r0;
      if (r0 <= target) { info.converged = 1; break; }
    }
    if (k == 0) k = (int)std::ceil(std::log(2.0 * sk * res / target) / rate);
    if (k_hint > 0) k = std::min(k, k_hint);
    k_hint = 0;
    k = std::max(1, std::min(k, o.max_iter - info.iterations));
    if (!c->distributed() && mg.lattice_ok(L)) {
      // lattice kernel: the residual of the correction equation r - M e (= rhs - M (x + e)) rides along in the last
      // launch of the sequence -- no separate product with M for the check
      mg.smooth_lattice(s, L, w.r.p, nullptr, w.q.p, k, false, w.t.p);
      launch_axpby(s, nv, 1.0, x, 1.0, w.q.p, x);
      info.iterations += k;
      std::swap(w.r.p, w.t.p);
      launch_dot(s, nv, w.r.p, w.r.p, parts + P10 * kParts);
    } else {
      mg.smooth(s, L, w.r.p, nullptr, w.q.p, k);                       // e ~ M^{-1} r
      launch_axpby(s, nv, 1.0, x, 1.0, w.q.p, x);                      // x += e (e = 0 on Dirichlet dofs)
      info.iterations += k;
      residual_norm(w.r.p);
    }
    if (c->distributed()) c->comm->allreduce_sum(s, parts + P10 * kParts, kParts);
    if (deferred && first_pass) {
      double rr;
      host_sum_parts3(s, w, P10, PR0, PB0, rr, rr0, bb0);
      r0 = std::sqrt(rr0);
      bnorm = std::sqrt(bb0);
      target = std::max(o.atol, o.rtol * (bnorm > 0.0 ? bnorm : 1.0));
      w.last_target = target;
      info.residual0 = r0;
      res = std::sqrt(rr);
    } else {
      res = std::sqrt(host_sum_parts(s, w, P10));
    }
    first_pass = false;
    if (!std::isfinite(res)) return NSFEM_ERR_BREAKDOWN;
    info.residual = res;
    info.converged = res <= target;
  }
  return info.converged ? NSFEM_OK : NSFEM_ERR_NOT_CONVERGED;
}

static int correction_solve(nsfem_ctx* c, const nsfem_krylov_opts& o, nsfem_solve_info& info) {
  if (o.precond == 2) return correction_solve_chebyshev(c, o, info);
  LinOp op;
  op.A = &c->M2;
  op.nv = c->mesh.dim;
  op.rowmask = c->mask_v.p;
  op.maskmode = MASK_ZERO;
  op.dinv = c->dinv_m.p;
  fill_linop(c, op, true);
  return pcg(c->stream, c->kw, op, c->rhs_v.p, c->state[NSFEM_U0].p, o, info, false);
}

// ------------------------------------------------------- assembly seam + solve
extern "C" int nsfem_assemble(nsfem_ctx* ctx, int system, uint32_t flags) {
  API_BEGIN
  NSFEM_REQUIRE(ctx, "null context");
  switch (system) {
    case NSFEM_SYS_MOMENTUM:
      if (flags & 1u) momentum_begin_step(ctx);
      (void)momentum_residual(ctx);
      momentum_jacobian(ctx);
      break;
    case NSFEM_SYS_POISSON:
      poisson_assemble(ctx);
      break;
    case NSFEM_SYS_CORRECTION:
      correction_assemble(ctx, true);      // (the same pass as the fused driver: its Chebyshev solve may start from it)
      break;
    default:
      throw Error(NSFEM_ERR_ARG, "nsfem_assemble: system not available");
  }
  ctx->assembled_system = system;
  NSFEM_HIP(hipStreamSynchronize(ctx->stream));
  API_END(ctx)
}

extern "C" int nsfem_residual_norm(nsfem_ctx* ctx, int system, double* out) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && out, "null argument");
  NSFEM_REQUIRE(system == ctx->assembled_system, "system not assembled");
  const bool vel = (system != NSFEM_SYS_POISSON);
  const double* b = vel ? ctx->rhs_v.p : ctx->rhs_p.p;
  const int64_t n = vel ? nvel(ctx) : npre(ctx);
  *out = global_norm(ctx, n, const_cast<double*>(b), nullptr);
  API_END(ctx)
}

extern "C" int nsfem_get_rhs(nsfem_ctx* ctx, int system, double* host, int64_t n) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && host, "null argument");
  const bool vel = (system != NSFEM_SYS_POISSON);
  NSFEM_REQUIRE(n == (vel ? nvel(ctx) : npre(ctx)), "bad size");
  NSFEM_HIP(hipMemcpyAsync(host, vel ? ctx->rhs_v.p : ctx->rhs_p.p, sizeof(double) * n,
                           hipMemcpyDeviceToHost, ctx->stream));
  NSFEM_HIP(hipStreamSynchronize(ctx->stream));
  API_END(ctx)
}

static nsfem_krylov_opts hinted(nsfem_krylov_opts k, const nsfem_ctx::SolveHint& h, bool exact = false);
static void note_solve(nsfem_ctx::SolveHint& h, const nsfem_solve_info& si, const nsfem_krylov_opts& k, double target);
static int next_hint(const nsfem_solve_info& si, const nsfem_krylov_opts& k, double target);
extern "C" int nsfem_solve(nsfem_ctx* ctx, int system, const nsfem_krylov_opts* opts,
                           nsfem_solve_info* info) {
  nsfem_solve_info local;
  API_BEGIN
  NSFEM_REQUIRE(ctx && opts, "null argument");
  NSFEM_REQUIRE(system == ctx->assembled_system, "system not assembled");
  nsfem_solve_info& inf = info ? *info : local;
  int rc = NSFEM_OK;
  switch (system) {
    case NSFEM_SYS_MOMENTUM: rc = momentum_solve_update(ctx, *opts, inf); break;
    case NSFEM_SYS_POISSON: rc = poisson_solve(ctx, *opts, inf); break;
    case NSFEM_SYS_CORRECTION:
      // (the Chebyshev mass solve takes its step count from the previous solve: the same predictor as in the fused
      // step drivers, so that the explicit seam and the fused step stay bit for bit equal)
      rc = correction_solve(ctx, hinted(*opts, ctx->hint_cor, opts->precond == 2), inf);
      note_solve(ctx->hint_cor, inf, *opts, ctx->kw.last_target);
      break;
    default: throw Error(NSFEM_ERR_ARG, "nsfem_solve: system not available");
  }
  if (rc == NSFEM_ERR_BREAKDOWN) throw Error(rc, "Krylov breakdown");
  if (rc == NSFEM_ERR_NOT_CONVERGED) throw Error(rc, "Krylov solver did not converge");
  API_END(ctx)
}

// mesh arrays, pattern and the P1 stiffness / mass matrices of one coarse level (same space
// dimension as the fine mesh)
static void fill_p1_level(nsfem_ctx* ctx, nsfem_ctx::P1Level* lv, int n_vertices, int nc,
                          const double* coords, const int32_t* cells,
                          const int32_t* dofmap = nullptr, int n_dofs = 0) {
  hipStream_t s = ctx->stream;
  const int dim = ctx->mesh.dim, nl1 = dim + 1;
  // numbering of the P1 space: vertex ids, or (periodic levels) the given cell dof map -- the
  // geometry always comes from the vertex coordinates of the cells
  const int32_t* dm = dofmap ? dofmap : cells;
  const int n = dofmap ? n_dofs : n_vertices;
  NSFEM_REQUIRE(n > 0 && n <= n_vertices, "bad number of coarse dofs");
  lv->n = n;
  std::vector<double> vx((size_t)nl1 * dim * nc);
  std::vector<int32_t> p1((size_t)nl1 * nc);
  for (int c = 0; c < nc; ++c)
    for (int v = 0; v < nl1; ++v) {
      const int vid = cells[(size_t)c * nl1 + v];
      NSFEM_REQUIRE(vid >= 0 && vid < n_vertices, "coarse cell vertex id out of range");
      const int dof = dm[(size_t)c * nl1 + v];
      NSFEM_REQUIRE(dof >= 0 && dof < n, "coarse cell dof id out of range");
      p1[(size_t)v * nc + c] = dof;
      for (int k = 0; k < dim; ++k) vx[(size_t)(dim * v + k) * nc + c] = coords[(size_t)vid * dim + k];
    }
  lv->mesh.dim = dim;
  lv->mesh.n_cells = nc;
  lv->mesh.n_p1 = n;
  lv->mesh.n_vertices = n_vertices;
  lv->mesh.vx.upload(vx, s);
  lv->mesh.p1.upload(p1, s);
  HostPattern h;
  build_pattern(n, n, nc, dm, nl1, dm, nl1, true, h);
  upload_pattern(s, h, lv->pat, true);
  lv->K.init(&lv->pat, 1, 1, s);
  lv->M.init(&lv->pat, 1, 1, s);
  lv->Lc.init(&lv->pat, 1, 1, s);
  launch_assemble_p1_scalar(s, lv->mesh, lv->pat, lv->K.vals.p, lv->M.vals.p);
  // lattice levels: stencil dictionary of the level's operators (smoothing steps of both hierarchies;
  // the multi-step lattice kernel needs it)
  // (2D: down to 3 x 3 nodes -- the fused legs of mglegs.hip run the whole bottom of a cycle from these tables)
  if (build_stencil_dict(s, lv->pat, lv->M.vals.p, lv->K.vals.p, lv->dict, 1, false, dim == 2 ? 9 : 0))
    lv->K.dict = lv->M.dict = lv->Lc.dict = &lv->dict;
  lv->K.sell_update(s);
  lv->M.sell_update(s);
}

extern "C" int nsfem_mg_add_level(nsfem_ctx* ctx, const nsfem_mg_level_desc* d) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && d, "null argument");
  NSFEM_REQUIRE(!ctx->mg_built, "hierarchy already finalized");
  NSFEM_REQUIRE(d->n_vertices > 0 && d->n_cells > 0 && d->coords && d->cells && d->p_rowptr &&
                    d->p_col && d->p_val, "bad level description");
  const int n_fine = ctx->coarse.empty() ? ctx->mesh.n_p1 : ctx->coarse.back()->n;
  NSFEM_REQUIRE(d->n_fine == n_fine, "prolongation rows must match the previous level");
  hipStream_t s = ctx->stream;
  nsfem_ctx::P1Level* lv = new nsfem_ctx::P1Level();
  ctx->coarse.push_back(lv);
  fill_p1_level(ctx, lv, d->n_vertices, d->n_cells, d->coords, d->cells, d->dofmap, d->n_dofs);
  NSFEM_REQUIRE(d->transfer_kind >= 0 && d->transfer_kind <= 2, "transfer_kind: 0 (from the values), 1 nested, 2 interpolation");
  lv->to_finer.build(s, n_fine, lv->n, d->p_rowptr, d->p_col, d->p_val, d->transfer_kind);
  if (d->ghost) {                       // one flag per dof of the level (= per vertex without a dof map)
    lv->h_ghost.assign(d->ghost, d->ghost + lv->n);
    lv->halo = to_halo(d->halo);
    lv->has_halo = true;
    mark_interior_blocks(lv->pat, lv->h_ghost);
  }
  NSFEM_HIP(hipStreamSynchronize(s));
  API_END(ctx)
}

// Algebraic pressure Laplacian of the Schur-complement preconditioner, level by level (level 0 =
// fine P1 space): A_L = D_f diag(M_v)^{-1} D_f^T and its Galerkin coarsenings, computed by the
// host at set-up.  Replaces the geometric stiffness matrices in the mg_s hierarchy: the algebraic
// form carries the correct behaviour on open (natural-outflow) boundaries, where a strongly
// imposed Dirichlet condition leaves one badly preconditioned mode per outflow node.
extern "C" int nsfem_mg_set_schur_operator(nsfem_ctx* ctx, int level, int32_t n,
                                           const int32_t* rowptr, const int32_t* col,
                                           const double* val, int singular) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && rowptr && col && val, "null argument");
  ctx->schur_singular = singular ? 1 : 0;
  NSFEM_REQUIRE(ctx->mg_built, "build the hierarchy first (nsfem_mg_finalize)");
  NSFEM_REQUIRE(level >= 0 && level < (int)ctx->mg_s.lv.size(), "no such level");
  NSFEM_REQUIRE(n == ctx->mg_s.lv[level].n, "operator size does not match the level");
  hipStream_t s = ctx->stream;
  if (ctx->schur_ops.size() < ctx->mg_s.lv.size()) ctx->schur_ops.resize(ctx->mg_s.lv.size());
  nsfem_ctx::CsrOp*& op = ctx->schur_ops[level];
  delete op;
  op = new nsfem_ctx::CsrOp();
  Pattern& p = op->pat;
  p.n_rows = p.n_cols = n;
  p.nnz = rowptr[n];
  p.h_rowptr.assign(rowptr, rowptr + n + 1);
  p.h_col.assign(col, col + p.nnz);
  std::vector<int32_t> diag((size_t)n, -1);
  for (int r = 0; r < n; ++r)
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
      NSFEM_REQUIRE(col[k] >= 0 && col[k] < n, "column out of range");
      if (col[k] == r) diag[r] = k;
    }
  for (int r = 0; r < n; ++r) NSFEM_REQUIRE(diag[r] >= 0, "operator without a stored diagonal");
  p.rowptr.upload(p.h_rowptr, s);
  p.col.upload(p.h_col, s);
  p.diag.upload(diag, s);
  build_rowblocks(p, s);
  op->mat.pat = &p;
  op->mat.br = op->mat.bc = 1;
  op->mat.vals.upload(val, (size_t)p.nnz, s);
  NSFEM_HIP(hipStreamSynchronize(s));
  if (build_stencil_dict(s, p, op->mat.vals.p, nullptr, op->dict)) op->mat.dict = &op->dict;
  op->mat.sell_update(s);
  NSFEM_HIP(hipStreamSynchronize(s));
  ctx->mg_s.lv[level].A = &op->mat;
  ctx->mg_s.lv[level].additive = ctx->schur_additive;
  ctx->mg_s_dirty = true;
  API_END(ctx)
}

// Partitioned meshes: the operators handed to nsfem_mg_set_schur_operator AFTER this call are the
// rank's additive parts A_r = D_r W_r D_r^T (W_r: 1 / M_v,jj on the velocity dofs this rank owns,
// 0 elsewhere), ghost rows included; the hierarchy applies them with a forward halo exchange of
// the input and a reverse (add) exchange of the ghost rows of the product.
extern "C" int nsfem_mg_set_schur_mode(nsfem_ctx* ctx, int additive) {
  API_BEGIN
  NSFEM_REQUIRE(ctx, "null argument");
  ctx->schur_additive = additive != 0;
  API_END(ctx)
}

// max (op = 1) or sum (op = 0) over the ranks of `count` host doubles (set-up-time agreement on
// flags and bounds; single contexts: a no-op)
extern "C" int nsfem_comm_allreduce(nsfem_ctx* ctx, double* values, int count, int op) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && values && count > 0 && count <= 1024, "bad argument");
  if (ctx->distributed()) {
    hipStream_t s = ctx->stream;
    DevBuf<double> tmp;
    tmp.upload(values, (size_t)count, s);
    if (op) ctx->comm->allreduce_max(s, tmp.p, count);
    else ctx->comm->allreduce_sum(s, tmp.p, count);
    NSFEM_HIP(hipMemcpyAsync(values, tmp.p, sizeof(double) * count, hipMemcpyDeviceToHost, s));
    NSFEM_HIP(hipStreamSynchronize(s));
  }
  API_END(ctx)
}

// replicated global coarsest mesh of a partitioned hierarchy; `offset` = global id of this
// rank's local coarsest node 0
extern "C" int nsfem_mg_set_global_coarse(nsfem_ctx* ctx, int32_t n_vertices, int32_t n_cells,
                                          const double* coords, const int32_t* cells,
                                          int64_t offset) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && coords && cells && n_vertices > 0 && n_cells > 0, "bad argument");
  NSFEM_REQUIRE(!ctx->mg_built, "hierarchy already finalized");
  hipStream_t s = ctx->stream;
  delete ctx->global_coarse;
  nsfem_ctx::P1Level* lv = ctx->global_coarse = new nsfem_ctx::P1Level();
  fill_p1_level(ctx, lv, n_vertices, n_cells, coords, cells);
  ctx->glob_off = offset;
  NSFEM_HIP(hipStreamSynchronize(s));
  API_END(ctx)
}

// a coarser level of the REPLICATED global hierarchy below the mesh of nsfem_mg_set_global_coarse
// (finest first; prolongation rows = nodes of the previous global level)
extern "C" int nsfem_mg_add_global_level(nsfem_ctx* ctx, const nsfem_mg_level_desc* d) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && d, "null argument");
  NSFEM_REQUIRE(!ctx->mg_built, "hierarchy already finalized");
  NSFEM_REQUIRE(ctx->global_coarse, "set the global coarsest mesh first (nsfem_mg_set_global_coarse)");
  NSFEM_REQUIRE(d->n_vertices > 0 && d->n_cells > 0 && d->coords && d->cells && d->p_rowptr &&
                    d->p_col && d->p_val, "bad level description");
  const int n_fine = ctx->global_tail.empty() ? ctx->global_coarse->n : ctx->global_tail.back()->n;
  NSFEM_REQUIRE(d->n_fine == n_fine, "prolongation rows must match the previous global level");
  hipStream_t s = ctx->stream;
  nsfem_ctx::P1Level* lv = new nsfem_ctx::P1Level();
  ctx->global_tail.push_back(lv);
  fill_p1_level(ctx, lv, d->n_vertices, d->n_cells, d->coords, d->cells, d->dofmap, d->n_dofs);
  lv->to_finer.build(s, n_fine, lv->n, d->p_rowptr, d->p_col, d->p_val, d->transfer_kind);
  NSFEM_HIP(hipStreamSynchronize(s));
  API_END(ctx)
}

extern "C" int nsfem_mg_set_global_coarse_constrained(nsfem_ctx* ctx, int32_t n_vertices, int32_t n_cells,
                                                      const double* coords, const int32_t* cells,
                                                      const int32_t* dofmap, int32_t n_dofs,
                                                      int64_t offset) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && coords && cells && dofmap && n_vertices > 0 && n_cells > 0 && n_dofs > 0, "bad argument");
  NSFEM_REQUIRE(!ctx->mg_built, "hierarchy already finalized");
  NSFEM_REQUIRE(!ctx->global_coarse, "global coarsest mesh already set");
  NSFEM_REQUIRE(offset >= 0 && offset < n_dofs, "offset outside the global coarsest space");
  nsfem_ctx::P1Level* lv = ctx->global_coarse = new nsfem_ctx::P1Level();
  fill_p1_level(ctx, lv, n_vertices, n_cells, coords, cells, dofmap, n_dofs);
  ctx->glob_off = offset;
  NSFEM_HIP(hipStreamSynchronize(ctx->stream));
  API_END(ctx)
}

// unstructured partitions: the local coarsest level maps into the global coarsest mesh through
// an index list instead of an offset
extern "C" int nsfem_mg_set_global_index(nsfem_ctx* ctx, int32_t n_local, const int32_t* local_to_global) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && local_to_global && n_local > 0, "bad argument");
  NSFEM_REQUIRE(!ctx->mg_built, "hierarchy already finalized");
  NSFEM_REQUIRE(ctx->global_coarse, "set the global coarsest mesh first (nsfem_mg_set_global_coarse)");
  for (int i = 0; i < n_local; ++i)
    NSFEM_REQUIRE(local_to_global[i] >= 0 && local_to_global[i] < ctx->global_coarse->n, "global id out of range");
  ctx->h_glob_idx.assign(local_to_global, local_to_global + n_local);
  ctx->glob_idx.upload(ctx->h_glob_idx, ctx->stream);
  NSFEM_HIP(hipStreamSynchronize(ctx->stream));
  API_END(ctx)
}

// halo of an unstructured partition as index lists; target 0: P2 nodes, 1: P1 nodes of the
// context (after nsfem_set_partition), 2 + l: P1 nodes of multigrid level l (after its
// nsfem_mg_add_level); before nsfem_mg_finalize
extern "C" int nsfem_set_halo_lists(nsfem_ctx* ctx, int target, const nsfem_halo_lists* d) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && d && d->n_neighbours >= 0, "bad argument");
  NSFEM_REQUIRE(!ctx->mg_built, "set the halo lists before nsfem_mg_finalize");
  int64_t n_nodes = 0;
  HaloRange* h = nullptr;
  if (target == 0) { h = &ctx->halo_p2; n_nodes = ctx->mesh.n_p2; }
  else if (target == 1) { h = &ctx->halo_p1; n_nodes = ctx->mesh.n_p1; }
  else {
    NSFEM_REQUIRE(target >= 2 && target - 2 < (int)ctx->coarse.size(), "no such multigrid level");
    h = &ctx->coarse[target - 2]->halo;
    n_nodes = ctx->coarse[target - 2]->n;
    ctx->coarse[target - 2]->has_halo = true;
  }
  HaloLists* L = new HaloLists();
  ctx->halo_lists.push_back(L);
  const int nn = d->n_neighbours;
  NSFEM_REQUIRE(nn == 0 || (d->neighbour && d->send_ptr && d->recv_ptr), "null list");
  L->nbr.assign(d->neighbour, d->neighbour + nn);
  L->send_ptr.assign(1, 0);
  L->recv_ptr.assign(1, 0);
  if (nn > 0) {
    L->send_ptr.assign(d->send_ptr, d->send_ptr + nn + 1);
    L->recv_ptr.assign(d->recv_ptr, d->recv_ptr + nn + 1);
  }
  NSFEM_REQUIRE(L->send_ptr[0] == 0 && L->recv_ptr[0] == 0, "list pointers start at 0");
  for (int k = 0; k < nn; ++k) {
    NSFEM_REQUIRE(L->nbr[k] >= 0 && L->send_ptr[k + 1] >= L->send_ptr[k] && L->recv_ptr[k + 1] >= L->recv_ptr[k],
                  "malformed halo lists");
  }
  for (int64_t i = 0; i < L->n_send(); ++i)
    NSFEM_REQUIRE(d->send_idx[i] >= 0 && d->send_idx[i] < n_nodes, "send index out of range");
  for (int64_t i = 0; i < L->n_recv(); ++i)
    NSFEM_REQUIRE(d->recv_idx[i] >= 0 && d->recv_idx[i] < n_nodes, "receive index out of range");
  if (L->n_send() > 0) L->send_idx.upload(d->send_idx, (size_t)L->n_send(), ctx->stream);
  if (L->n_recv() > 0) L->recv_idx.upload(d->recv_idx, (size_t)L->n_recv(), ctx->stream);
  NSFEM_HIP(hipStreamSynchronize(ctx->stream));
  *h = HaloRange();
  h->lists = L;
  ctx->mg_mv.lv.clear();
  API_END(ctx)
}

extern "C" int nsfem_set_partition(nsfem_ctx* ctx, const nsfem_partition_desc* d) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && d && d->p2_ghost && d->p1_ghost, "null argument");
  NSFEM_REQUIRE(!ctx->mg_built, "set the partition before building the multigrid hierarchy");
  hipStream_t s = ctx->stream;
  const int n2 = ctx->mesh.n_p2, n1 = ctx->mesh.n_p1;
  ctx->h_ghost_p2.assign(d->p2_ghost, d->p2_ghost + n2);
  ctx->h_ghost_p1.assign(d->p1_ghost, d->p1_ghost + n1);
  std::vector<uint8_t> gv((size_t)ctx->mesh.dim * n2), gp((size_t)n1);
  for (int i = 0; i < n2; ++i)
    for (int a = 0; a < ctx->mesh.dim; ++a) gv[(size_t)ctx->mesh.dim * i + a] = d->p2_ghost[i] ? 2 : 0;
  for (int i = 0; i < n1; ++i) gp[i] = d->p1_ghost[i] ? 2 : 0;
  // (a single-rank "partition" has no ghosts: keep the ghost arrays unallocated, so that none of
  // the ghost-zeroing launches run)
  bool any_ghost = false;
  for (uint8_t g : gv) any_ghost |= g != 0;
  for (uint8_t g : gp) any_ghost |= g != 0;
  if (any_ghost) {
    ctx->ghost_v.upload(gv, s);
    ctx->ghost_p.upload(gp, s);
  }
  ctx->halo_p2 = to_halo(d->p2_halo);
  mark_interior_blocks(ctx->p22, ctx->h_ghost_p2);   // row blocks that can run under the halo exchange
  mark_interior_blocks(ctx->p11, ctx->h_ghost_p1);
  ctx->mg_mv.lv.clear();            // rebuilt with the halo on the next Chebyshev mass solve
  ctx->halo_p1 = to_halo(d->p1_halo);
  ctx->n_p2_global = d->n_p2_global;
  ctx->n_p1_global = d->n_p1_global;
  ctx->partition_periodic = d->periodic != 0;
  if (ctx->comm) ctx->comm->periodic = ctx->partition_periodic;
  // masks carry the ghost flag from now on
  launch_overlay_ghost(s, nvel(ctx), ctx->ghost_v.p, ctx->mask_v.p);
  launch_overlay_ghost(s, n1, ctx->ghost_p.p, ctx->mask_p.p);
  ctx->dinv_m_ready = ctx->dinv_p_ready = false;
  NSFEM_HIP(hipStreamSynchronize(s));
  API_END(ctx)
}

extern "C" int nsfem_comm_attach_local(nsfem_ctx* ctx, void* group, int rank) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && group, "null argument");
  delete ctx->comm;
  ctx->comm = nullptr;
  ctx->comm = make_local_comm(group, rank);
  ctx->comm->periodic = ctx->partition_periodic;
  ctx->comm->overlap = ctx->overlap;
  API_END(ctx)
}

extern "C" int nsfem_comm_attach_rccl(nsfem_ctx* ctx, const char* id128, int rank, int size) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && id128, "null argument");
  NSFEM_HIP(hipSetDevice(ctx->device));
  delete ctx->comm;
  ctx->comm = nullptr;
  ctx->comm = make_rccl_comm(id128, rank, size);
  ctx->comm->periodic = ctx->partition_periodic;
  ctx->comm->overlap = ctx->overlap;
  API_END(ctx)
}

extern "C" int nsfem_comm_attach_shm(nsfem_ctx* ctx, const char* name, int rank, int size, int64_t slot_bytes) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && name, "null argument");
  NSFEM_HIP(hipSetDevice(ctx->device));
  delete ctx->comm;
  ctx->comm = nullptr;
  ctx->comm = make_shm_comm(name, rank, size, slot_bytes);
  ctx->comm->periodic = ctx->partition_periodic;
  ctx->comm->overlap = ctx->overlap;
  API_END(ctx)
}

// partitioned meshes: communicator, halo ranges and ghost flags of every level, and the
// replicated global coarsest operator.  `first_p1` = index of the fine P1 level in mg.lv.
static void wire_partition(nsfem_ctx* ctx, Multigrid& mg, size_t first_p1, bool momentum) {
  if (!ctx->distributed()) return;
  NSFEM_REQUIRE(ctx->global_coarse, "partitioned multigrid needs nsfem_mg_set_global_coarse");
  mg.comm = ctx->comm;
  if (first_p1 == 1) {
    mg.lv[0].halo = ctx->halo_p2;
    mg.lv[0].has_halo = true;
    mg.lv[0].h_ghost = &ctx->h_ghost_p2;
  }
  mg.lv[first_p1].halo = ctx->halo_p1;
  mg.lv[first_p1].has_halo = true;
  mg.lv[first_p1].h_ghost = &ctx->h_ghost_p1;
  for (size_t l = 0; l < ctx->coarse.size(); ++l) {
    nsfem_ctx::P1Level* c = ctx->coarse[l];
    NSFEM_REQUIRE(c->has_halo, "coarse level without partition data");
    mg.lv[first_p1 + 1 + l].halo = c->halo;
    mg.lv[first_p1 + 1 + l].has_halo = true;
    mg.lv[first_p1 + 1 + l].h_ghost = &c->h_ghost;
  }
  mg.globA = momentum ? &ctx->global_coarse->Lc : &ctx->global_coarse->K;
  mg.n_glob = ctx->global_coarse->n;
  mg.glob_off = ctx->glob_off;
  if (!ctx->h_glob_idx.empty()) {
    NSFEM_REQUIRE((int)ctx->h_glob_idx.size() == mg.lv.back().n, "global index list does not match the coarsest level");
    mg.glob_idx = ctx->glob_idx.p;
    mg.h_glob_idx = &ctx->h_glob_idx;
  }
  if (!ctx->global_tail.empty()) {          // replicated hierarchy below the global coarse mesh
    Multigrid& t = momentum ? ctx->mg_v_tail : ctx->mg_p_tail;
    t.nv = mg.nv; t.degree = mg.degree; t.pre_degree = mg.pre_degree; t.eig_ratio = mg.eig_ratio;
    t.coarse_dense_max = mg.coarse_dense_max;
    t.comm = nullptr;
    t.own_mask0 = true;
    t.lv.clear();
    t.lv.resize(1 + ctx->global_tail.size());
    t.lv[0].A = mg.globA; t.lv[0].n = mg.n_glob;
    for (size_t l = 0; l < ctx->global_tail.size(); ++l) {
      nsfem_ctx::P1Level* c = ctx->global_tail[l];
      t.lv[l].P = &c->to_finer.P; t.lv[l].R = &c->to_finer.R; t.lv[l].h_inj = &c->to_finer.h_inj;
      t.lv[l].transfer = &c->to_finer;
      t.lv[l + 1].A = momentum ? &c->Lc : &c->K; t.lv[l + 1].n = c->n;
    }
    t.setup_work(ctx->stream);
    mg.tail = &t;
  }
  mg.glob_wrap = ctx->partition_periodic;
  NSFEM_REQUIRE(mg.glob_idx || (mg.glob_off >= 0 && (mg.glob_wrap || mg.glob_off + mg.lv.back().n <= mg.n_glob)),
                "local coarsest level does not fit into the global coarsest mesh");
}

extern "C" int nsfem_mg_finalize(nsfem_ctx* ctx, const nsfem_mg_opts* o) {
  API_BEGIN
  NSFEM_REQUIRE(ctx, "null context");
  NSFEM_REQUIRE(!ctx->mg_built, "hierarchy already finalized");
  hipStream_t s = ctx->stream;
  // P2 <- P1 on the fine mesh: identity at vertices, average at edge midpoints
  {
    const int n2 = ctx->mesh.n_p2, nc = ctx->mesh.n_cells;
    std::vector<int32_t> a((size_t)n2, -1), b((size_t)n2, -1);
    const int dim = ctx->mesh.dim, nl1 = dim + 1, nl2 = dim == 2 ? 6 : 10;
    const int ends2[3][2] = {{1, 2}, {0, 2}, {0, 1}};                               // UFC edges
    const int ends3[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
    for (int c = 0; c < nc; ++c) {
      const int32_t* p2 = &ctx->h_p2map[(size_t)c * nl2];
      const int32_t* p1 = &ctx->h_p1map[(size_t)c * nl1];
      for (int v = 0; v < nl1; ++v) { a[p2[v]] = p1[v]; b[p2[v]] = -1; }
      for (int e = 0; e < nl2 - nl1; ++e) {
        const int e0 = dim == 2 ? ends2[e][0] : ends3[e][0], e1 = dim == 2 ? ends2[e][1] : ends3[e][1];
        a[p2[nl1 + e]] = std::min(p1[e0], p1[e1]);
        b[p2[nl1 + e]] = std::max(p1[e0], p1[e1]);
      }
    }
    std::vector<int32_t> rp((size_t)n2 + 1, 0), col;
    std::vector<double> val;
    for (int i = 0; i < n2; ++i) {
      NSFEM_REQUIRE(a[i] >= 0, "P2 node not referenced by any cell");
      if (b[i] < 0 || b[i] == a[i]) { col.push_back(a[i]); val.push_back(1.0); }
      else { col.push_back(a[i]); val.push_back(0.5); col.push_back(b[i]); val.push_back(0.5); }
      rp[i + 1] = (int32_t)col.size();
    }
    ctx->t_p2p1.build(s, n2, ctx->mesh.n_p1, rp.data(), col.data(), val.data(), 1 /* nested: P1 in P2 */);
  }
  ctx->Lc0.init(&ctx->p11, 1, 1, s);
  const int degree = (o && o->smoother_degree > 0) ? o->smoother_degree : 2;
  const double ratio = (o && o->eig_ratio > 1.0) ? o->eig_ratio : 4.0;
  int dense_max = (o && o->coarse_dense_max > 0) ? o->coarse_dense_max : 1200;
  if (const char* e = std::getenv("NSFEM_DENSE_MAX")) dense_max = std::atoi(e);   // experiments
  // pressure Poisson hierarchy: P1 fine -> coarse P1 levels
  {
    Multigrid& mg = ctx->mg_p;
    mg.nv = 1; mg.degree = degree; mg.eig_ratio = ratio; mg.coarse_dense_max = dense_max;
    mg.lv.clear();
    mg.lv.resize(1 + ctx->coarse.size());
    mg.lv[0].A = &ctx->Ap; mg.lv[0].n = ctx->mesh.n_p1; mg.lv[0].mask = ctx->mask_p.p;
    for (size_t l = 0; l < ctx->coarse.size(); ++l) {
      nsfem_ctx::P1Level* c = ctx->coarse[l];
      mg.lv[l].P = &c->to_finer.P; mg.lv[l].R = &c->to_finer.R; mg.lv[l].h_inj = &c->to_finer.h_inj;
      mg.lv[l].transfer = &c->to_finer;
      mg.lv[l + 1].A = &c->K; mg.lv[l + 1].n = c->n;
    }
    wire_partition(ctx, mg, 0, false);
    mg.setup_work(s);
  }
  // momentum hierarchy: P2 fine -> P1 fine -> coarse P1 levels, operator a M + b K
  {
    Multigrid& mg = ctx->mg_v;
    mg.nv = ctx->mesh.dim; mg.degree = degree; mg.eig_ratio = ratio; mg.coarse_dense_max = dense_max;
    // non-symmetric cycle V(0, degree+1): BiCGStab does not need a symmetric preconditioner, and
    // without pre-smoothing the fine residual is the input itself (two fine SpMVs fewer per
    // cycle; measured 6 % faster steps than V(2,2) at equal iteration counts)
    mg.pre_degree = 0;
    mg.degree = degree + 1;
    mg.identity_rows = true;     // z = r on Dirichlet rows, written by the last smoothing step
    if (const char* e = std::getenv("NSFEM_MGV_PRE")) mg.pre_degree = std::atoi(e);
    if (const char* e = std::getenv("NSFEM_MGV_POST")) mg.degree = std::atoi(e);
    mg.lv.clear();
    mg.lv.resize(2 + ctx->coarse.size());
    mg.lv[0].A = &ctx->L; mg.lv[0].n = ctx->mesh.n_p2; mg.lv[0].mask = ctx->mask_v.p;
    mg.lv[0].P = &ctx->t_p2p1.P; mg.lv[0].R = &ctx->t_p2p1.R; mg.lv[0].h_inj = &ctx->t_p2p1.h_inj;
    mg.lv[0].transfer = &ctx->t_p2p1;
    mg.lv[1].A = &ctx->Lc0; mg.lv[1].n = ctx->mesh.n_p1;
    for (size_t l = 0; l < ctx->coarse.size(); ++l) {
      nsfem_ctx::P1Level* c = ctx->coarse[l];
      mg.lv[l + 1].P = &c->to_finer.P; mg.lv[l + 1].R = &c->to_finer.R;
      mg.lv[l + 1].h_inj = &c->to_finer.h_inj;
      mg.lv[l + 1].transfer = &c->to_finer;
      mg.lv[l + 2].A = &c->Lc; mg.lv[l + 2].n = c->n;
    }
    wire_partition(ctx, mg, 1, true);
    mg.setup_work(s);
  }
  // Schur-complement pressure Laplacian (own Dirichlet set) and pressure mass smoother
  {
    if (!ctx->mask_s.p) { ctx->mask_s.alloc((size_t)npre(ctx)); ctx->mask_s.zero(s); }
    Multigrid& mg = ctx->mg_s;
    mg.nv = 1; mg.degree = degree; mg.eig_ratio = ratio; mg.coarse_dense_max = dense_max;
    mg.lv.clear();
    mg.lv.resize(1 + ctx->coarse.size());
    mg.lv[0].A = &ctx->Ap; mg.lv[0].n = ctx->mesh.n_p1; mg.lv[0].mask = ctx->mask_s.p;
    for (size_t l = 0; l < ctx->coarse.size(); ++l) {
      nsfem_ctx::P1Level* c = ctx->coarse[l];
      mg.lv[l].P = &c->to_finer.P; mg.lv[l].R = &c->to_finer.R; mg.lv[l].h_inj = &c->to_finer.h_inj;
      mg.lv[l].transfer = &c->to_finer;
      mg.lv[l + 1].A = &c->K; mg.lv[l + 1].n = c->n;
    }
    wire_partition(ctx, mg, 0, false);
    mg.setup_work(s);
    Multigrid& mm = ctx->mg_m;          // 4 Chebyshev-Jacobi steps on the P1 mass matrix
    mm.nv = 1; mm.coarse_dense_max = 0; mm.coarse_steps = 4; mm.eig_ratio = 8.0;
    mm.lv.clear();
    mm.lv.resize(1);
    mm.lv[0].A = &ctx->Mp; mm.lv[0].n = ctx->mesh.n_p1; mm.lv[0].mask = nullptr;
    mm.smoother_only = true;
    if (ctx->distributed()) {           // partitioned: halo exchange before every smoothing SpMV
      mm.comm = ctx->comm;
      mm.lv[0].halo = ctx->halo_p1;
      mm.lv[0].has_halo = true;
      mm.lv[0].h_ghost = &ctx->h_ghost_p1;
    }
    mm.setup_work(s);
  }
  ctx->mg_built = true;
  ctx->mg_p_dirty = ctx->mg_v_dirty = ctx->mg_s_dirty = true;
  ctx->L_dirty = true;       // (re)compute the coarse momentum operators
  NSFEM_HIP(hipStreamSynchronize(s));
  API_END(ctx)
}

// communication counters of this rank since the last reset: {all-reduce calls, all-reduce payload
// bytes, halo exchanges, halo bytes sent}
extern "C" int nsfem_comm_stats(nsfem_ctx* ctx, int64_t out[4], int reset) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && out, "null argument");
  out[0] = out[1] = out[2] = out[3] = 0;
  if (ctx->comm) {
    out[0] = ctx->comm->n_allreduce; out[1] = ctx->comm->bytes_allreduce;
    out[2] = ctx->comm->n_exchange; out[3] = ctx->comm->bytes_exchange;
    if (!ctx->comm->hist.empty() && ctx->comm->rank == 0 && !reset) {
      static const char* kind[3] = {"all-reduce", "exchange", "reverse-add"};
      for (auto& e : ctx->comm->hist)
        std::fprintf(stderr, "[comm hist rank 0] %-11s %9lld B x %lld\n", kind[e.first.first], (long long)e.first.second,
                     (long long)e.second);
    }
    if (reset) {
      ctx->comm->n_allreduce = ctx->comm->bytes_allreduce = ctx->comm->n_exchange = ctx->comm->bytes_exchange = 0;
      ctx->comm->hist.clear();
    }
  }
  API_END(ctx)
}

// number of halo exchanges since the last reset that ran on the communicator's stream under the
// interior rows of the product consuming them (nsfem_set_overlap)
extern "C" int nsfem_comm_overlapped(nsfem_ctx* ctx, int64_t* out, int reset) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && out, "null argument");
  *out = ctx->comm ? ctx->comm->n_overlapped : 0;
  if (reset && ctx->comm) ctx->comm->n_overlapped = 0;
  API_END(ctx)
}

// partitioned meshes: run the halo exchanges of the Krylov operators and of the smoothing steps on
// the communicator's own stream, concurrently with the row blocks that reference no ghost column
extern "C" int nsfem_set_overlap(nsfem_ctx* ctx, int enable) {
  API_BEGIN
  NSFEM_REQUIRE(ctx, "null context");
  if (ctx->comm) ctx->comm->overlap = enable != 0;
  ctx->overlap = enable != 0;
  ctx->graph_epoch++;
  API_END(ctx)
}

extern "C" int nsfem_mg_set_halo_mode(nsfem_ctx* ctx, int relaxed) {
  API_BEGIN
  NSFEM_REQUIRE(ctx, "null context");
  // (tried in round 3 and removed: a "local" mode without any exchange inside the velocity cycle -- smoothing,
  // residuals and transfers on the rank-local operator, ranks coupled through the global coarse problem only --
  // cut the exchanges of a strong-scaling step from 187 to 133 but tripled the BiCGStab count (5.2 -> 19.5 on 8
  // ranks, 960^2) and stalled the Poisson CG altogether: dropped interface contributions in the restriction)
  for (Multigrid* mg : {&ctx->mg_v, &ctx->mg_p, &ctx->mg_s, &ctx->mg_m}) mg->relaxed_halo = relaxed != 0;
  ctx->graph_epoch++;
  API_END(ctx)
}

extern "C" int nsfem_mg_set_truncation(nsfem_ctx* ctx, double max_ratio, double coarse_tol) {
  API_BEGIN
  NSFEM_REQUIRE(ctx, "null context");
  NSFEM_REQUIRE(max_ratio >= 0.0 && coarse_tol > 0.0 && coarse_tol < 1.0, "truncation parameters out of range");
  ctx->mg_trunc_ratio = max_ratio;
  ctx->mg_trunc_tol = coarse_tol;
  ctx->L_dirty = true;
  ctx->graph_epoch++;
  API_END(ctx)
}

extern "C" int nsfem_default_step_opts(nsfem_step_opts* o) {
  if (!o) return NSFEM_ERR_ARG;
  std::memset(o, 0, sizeof(*o));
  o->newton_atol = 1e-10;
  o->newton_rtol = 1e-9;
  o->newton_max_iter = 50;
  o->convective_form = 0;
  nsfem_krylov_opts k;
  k.rtol = 1e-12;
  k.atol = 1e-14;
  k.max_iter = 20000;
  k.precond = 0;
  k.check_every = 1;
  k.first_check = 0;
  o->momentum = k;
  o->poisson = k;
  o->correction = k;
  return NSFEM_OK;
}

static nsfem_krylov_opts forced_opts(const nsfem_step_opts* o, const nsfem_krylov_opts& base, double r0);
// Every host convergence check of a Krylov solve is a device -> host round trip during which the
// GPU runs dry.  The same solve of the previous time step is an excellent predictor of the
// iteration count: the first check is postponed to one iteration before that count.
// Round 4: a check costs ~15 - 20 us of idle GPU, an iteration too many costs the iteration.  While the count is
// steady (the last two solves needed exactly the prediction) the first check sits AT the predicted count -- one round
// trip per solve --, otherwise and at every 8th solve (the probe that notices a falling count) one iteration before
// it.  exact: the prediction itself (the Chebyshev mass solve runs exactly that many steps before its only check).
static nsfem_krylov_opts hinted(nsfem_krylov_opts k, const nsfem_ctx::SolveHint& h, bool exact) {
  const bool steady = h.same >= 2 && (h.count & 7) != 7;
  k.first_check = std::max(k.first_check, (exact || steady) ? h.its : h.its - 1);
  return k;
}
static void note_solve(nsfem_ctx::SolveHint& h, const nsfem_solve_info& si, const nsfem_krylov_opts& k, double target) {
  const int next = next_hint(si, k, target);
  h.same = (si.converged && next == h.its && si.iterations == h.its) ? h.same + 1 : 0;
  h.its = next;
  ++h.count;
}
// The predictor for the next solve: the iteration count of this one, reduced when the postponed
// first check found the residual far below the target (linear-convergence estimate of the count
// that would have sufficed) -- otherwise a single long solve would keep all later ones long.
static int next_hint(const nsfem_solve_info& si, const nsfem_krylov_opts& k, double target) {
  (void)k;
  if (!si.converged) return 0;                     // a failed solve predicts nothing
  if (si.iterations <= 1 || !(si.residual > 0.0) || !(si.residual0 > si.residual)) return si.iterations;
  // (target: the solve's own absolute target max(atol, rtol |b|) -- with a good start vector |r0| << |b|)
  if (!(si.residual < target) || !(si.residual0 > target)) return si.iterations;
  const double need = si.iterations * std::log(si.residual0 / target) / std::log(si.residual0 / si.residual);
  return std::max(1, std::min(si.iterations, (int)std::ceil(need)));
}

extern "C" int nsfem_step_ipcs(nsfem_ctx* ctx, const nsfem_step_opts* opts, nsfem_step_info* info) {
  nsfem_step_info local;
  API_BEGIN
  NSFEM_REQUIRE(ctx && opts, "null argument");
  NSFEM_REQUIRE(opts->convective_form >= 0 && opts->convective_form <= 3, "unknown convective form");
  NSFEM_REQUIRE(opts->newton_max_iter > 0 && opts->newton_max_iter < NSFEM_MAX_NEWTON,
                "newton_max_iter out of range");
  NSFEM_REQUIRE(ctx->alpha[0] != 0.0, "the pressure-correction scheme needs alpha0 != 0");
  if (ctx->conv_form != opts->convective_form || ctx->picard) ctx->graph_epoch++;
  ctx->conv_form = opts->convective_form;
  ctx->picard = false;
  nsfem_step_info& inf = info ? *info : local;
  std::memset(&inf, 0, sizeof(inf));
  // ---- diffusion step: Newton (dolfin NewtonSolver, residual criterion)
  momentum_begin_step(ctx);
  double r = momentum_residual(ctx);
  const double r0 = r;
  inf.newton_residuals[0] = r;
  int it = 0;
  bool converged = r < opts->newton_atol;
  ctx->mf_active = use_matrix_free(ctx, opts);
  while (!converged && it < opts->newton_max_iter) {
    if (!ctx->mf_active) momentum_jacobian(ctx);
    nsfem_solve_info si;
    nsfem_ctx::SolveHint& hint = ctx->hint_mom[std::min(it, 3)];
    const nsfem_krylov_opts ko = forced_opts(opts, opts->momentum, r0);
    int rc = momentum_solve_update(ctx, hinted(ko, hint), si, r);
    note_solve(hint, si, ko, ctx->kw.last_target);
    inf.krylov_iterations_momentum += si.iterations;
    if (rc == NSFEM_ERR_BREAKDOWN) throw Error(rc, "BiCGStab breakdown in the diffusion step");
    if (rc == NSFEM_ERR_NOT_CONVERGED)
      throw Error(rc, "BiCGStab did not converge in the diffusion step");
    ++it;
    r = momentum_residual(ctx);
    inf.newton_residuals[it] = r;
    if (!std::isfinite(r)) throw Error(NSFEM_ERR_BREAKDOWN, "Newton residual is not finite");
    converged = (r / r0 < opts->newton_rtol) || (r < opts->newton_atol);
  }
  inf.newton_iterations = it;
  inf.converged = converged ? 1 : 0;
  ctx->mf_active = false;
  if (!converged) throw Error(NSFEM_ERR_NOT_CONVERGED, "Newton solver did not converge");
  // ---- projection step
  {
    nsfem_solve_info si;
    int rc;
    if (opts->poisson.precond == 3 && ctx->nbc_p == 0 && ctx->fd_p.ready() && ctx->distributed() == ctx->fd_p.strip()) {
      rc = poisson_direct_step(ctx, opts->poisson, si);
    } else {
      poisson_assemble(ctx, opts->pressure_extrapolation != 0);
      rc = poisson_solve(ctx, hinted(opts->poisson, ctx->hint_poi), si);
    }
    note_solve(ctx->hint_poi, si, opts->poisson, ctx->kw.last_target);
    inf.krylov_iterations_poisson = si.iterations;
    if (rc != NSFEM_OK) throw Error(rc, "CG failed in the projection step");
  }
  // ---- velocity correction step
  {
    correction_assemble(ctx, opts->correction.precond == 2);
    nsfem_solve_info si;
    int rc = correction_solve(ctx, hinted(opts->correction, ctx->hint_cor, opts->correction.precond == 2), si);
    note_solve(ctx->hint_cor, si, opts->correction, ctx->kw.last_target);
    inf.krylov_iterations_correction = si.iterations;
    if (rc != NSFEM_OK) throw Error(rc, "CG failed in the velocity correction step");
  }
  ctx->assembled_system = -1;
  API_END(ctx)
}

// Krylov tolerances of one Newton linear solve.  newton_forcing = 0: the caller's tolerances
// (a direct solver's accuracy, as the reference's LU).  newton_forcing = eta > 0 (inexact
// Newton): reduce the linear residual by eta, but never below a tenth of what the nonlinear
// criterion itself asks for -- the Newton loop still terminates on the reference's criterion,
// evaluated on the true nonlinear residual.
static nsfem_krylov_opts forced_opts(const nsfem_step_opts* o, const nsfem_krylov_opts& base, double r0) {
  nsfem_krylov_opts k = base;
  if (o->newton_forcing > 0.0) {
    k.rtol = std::max(k.rtol, o->newton_forcing);
    k.atol = std::max(k.atol, 0.1 * std::max(o->newton_atol, o->newton_rtol * r0));
  }
  return k;
}

// ---------------------------------------------------------------- monolithic BDF
// mixed operator  [[J, -c_p D^T], [-c_p D, 0]]  with identity rows on Dirichlet dofs
void nsfem_ctx::MixedOp::apply(hipStream_t s, const double* x, double* y) {
  const int64_t nv = nvel(c), np = npre(c);
  const double cp = c->coef[1];
  const bool dist = c->distributed();
  if (dist) {       // ghost entries of the input (in place: the Krylov vectors keep consistent ghosts)
    if (!c->mf_active) c->comm->exchange(s, c->halo_p2, const_cast<double*>(x), c->mesh.dim);
    c->comm->exchange(s, c->halo_p1, const_cast<double*>(x) + nv, 1);
  }
  if (c->mf_active) c->mom_mf.apply(s, x, y);          // (exchanges the velocity part itself)
  else launch_spmv(s, c->J, 1, x, y, c->mask_v.p, MASK_IDENTITY);
  launch_spmv_axpy(s, c->DT, 1, -cp, x + nv, y, c->mask_v.p, 1);       // (Jacobian product: dictionary copies allowed)
  launch_spmv_scaled(s, c->Dv, 1, -cp, x, y + nv, 1);
  launch_copy_at(s, c->nbc_p, c->bc_p_dofs.p, x + nv, y + nv);
  if (c->ghost_p.p) launch_zero_ghost(s, np, c->mask_p.p, y + nv);
}

// upper block-triangular preconditioner with the Cahouet-Chabard Schur approximation
//   z_p = -(1/c_p^2) (c_v M_p^{-1} + alpha0/k A_p^{+}) r_p ;  z_u = F^{-1} (r_u + c_p D^T z_p)
// F^{-1}, A_p^{+}: one multigrid V-cycle each;  M_p^{-1}: 4 Chebyshev-Jacobi steps
void nsfem_ctx::BlockPrec::apply(hipStream_t s, const double* r, double* z) {
  const int64_t nv = nvel(c), np = npre(c);
  const double cp = c->coef[1], cv = c->coef[2], a = c->alpha[0] / c->k + c->prec_shift;
  const double* rp = r + nv;
  double* zp = z + nv;
  c->mg_s.apply(s, rp, c->tmp_p.p);
  c->mg_m.apply(s, rp, c->rhs_p.p);
  launch_axpby(s, np, -a / (cp * cp), c->tmp_p.p, -cv / (cp * cp), c->rhs_p.p, zp);
  launch_copy_at(s, c->nbc_p, c->bc_p_dofs.p, rp, zp);
  if (c->distributed()) c->comm->exchange(s, c->halo_p1, zp, 1);
  NSFEM_HIP(hipMemcpyAsync(c->tmp_v.p, r, sizeof(double) * nv, hipMemcpyDeviceToDevice, s));
  launch_spmv_axpy(s, c->DT, 1, cp, zp, c->tmp_v.p, c->mask_v.p, 1);   // (preconditioner)
  if (c->ghost_p.p) launch_zero_ghost(s, np, c->mask_p.p, zp);      // keep ghost entries out of the dots
  c->mg_v.apply(s, c->tmp_v.p, z);        // (identity rows on the Dirichlet dofs: Multigrid::identity_rows)
}

// F_u = L u + g + c_c conv(u) - c_p D^T p ; F_p = -c_p D u ; Dirichlet rows x_i - g_i
static double bdf_residual(nsfem_ctx* c) {
  hipStream_t s = c->stream;
  const int64_t nv = nvel(c), np = npre(c);
  double* u = c->state[NSFEM_U0].p;
  double* p = c->state[NSFEM_P].p;
  double* b = c->rhs_m.p;
  momentum_residual_raw(c, u, b);
  launch_spmv_axpy(s, c->DT, 1, -c->coef[1], p, b, nullptr);
  launch_set_bc_residual(s, c->nbc_v, c->bc_v_dofs.p, c->bc_v_vals.p, u, b);
  launch_spmv_scaled(s, c->Dv, 1, -c->coef[1], u, b + nv);
  launch_set_bc_residual(s, c->nbc_p, c->bc_p_dofs.p, c->bc_p_vals.p, p, b + nv);
  if (c->ghost_v.p) {            // owners compute: ghost rows are zero (also for the norm)
    launch_zero_ghost(s, nv, c->mask_v.p, b);
    launch_zero_ghost(s, np, c->mask_p.p, b + nv);
  }
  return global_norm(c, nv + np, b, nullptr);
}

extern "C" int nsfem_step_bdf(nsfem_ctx* ctx, const nsfem_step_opts* opts, nsfem_step_info* info) {
  nsfem_step_info local;
  API_BEGIN
  NSFEM_REQUIRE(ctx && opts, "null argument");
  NSFEM_REQUIRE(opts->convective_form >= 0 && opts->convective_form <= 3, "unknown convective form");
  NSFEM_REQUIRE(opts->newton_max_iter > 0 && opts->newton_max_iter < NSFEM_MAX_NEWTON,
                "newton_max_iter out of range");
  if (ctx->conv_form != opts->convective_form || ctx->picard != (opts->picard != 0)) ctx->graph_epoch++;
  ctx->conv_form = opts->convective_form;
  NSFEM_REQUIRE(!ctx->distributed() || ctx->schur_singular < 0 || ctx->schur_additive,
                "partitioned meshes take the algebraic Schur Laplacian as additive rank parts "
                "(nsfem_mg_set_schur_mode)");
  NSFEM_REQUIRE(ctx->mg_built, "the monolithic step needs the multigrid hierarchy "
                               "(block preconditioner): call nsfem_mg_finalize");
  nsfem_step_info& inf = info ? *info : local;
  std::memset(&inf, 0, sizeof(inf));
  hipStream_t s = ctx->stream;
  const int64_t nv = nvel(ctx), np = npre(ctx);
  if (!ctx->rhs_m.p) {
    ctx->rhs_m.alloc((size_t)(nv + np));
    ctx->dx_m.alloc((size_t)(nv + np));
  }
  ctx->mixed_op.c = ctx;
  ctx->mixed_op.n = nv + np;
  ctx->block_prec.c = ctx;
  ctx->picard = opts->picard != 0;
  ensure_div_dicts(ctx);
  momentum_begin_step(ctx, false);
  double r = bdf_residual(ctx);
  const double r0 = r;
  inf.newton_residuals[0] = r;
  int it = 0;
  bool converged = r < opts->newton_atol;
  ctx->mf_active = use_matrix_free(ctx, opts);
  ctx->mom_mf.c = ctx;
  ctx->mom_mf.n = nv;
  ctx->mom_mf.vel_slot = NSFEM_U0;
  while (!converged && it < opts->newton_max_iter) {
    if (!ctx->mf_active) momentum_jacobian(ctx, NSFEM_U0);
    mg_refresh(ctx, true);
    mg_refresh_schur(ctx);
    LinOp op;
    op.custom = &ctx->mixed_op;
    op.prec = &ctx->block_prec;
    if (ctx->distributed()) op.comm = ctx->comm;        // all-reduce of the partial dot products
    op.graph_epoch = ctx->graph_epoch;
    nsfem_solve_info si;
    nsfem_ctx::SolveHint& hint = ctx->hint_mom[std::min(it, 3)];
    const nsfem_krylov_opts ko = forced_opts(opts, opts->momentum, r0);
    op.x_zero = true;               // (dx_m is not read: the first update of the solve writes it)
    op.known_bnorm = r;             // |rhs_m| = the Newton residual norm just evaluated (bdf_residual)
    int rc = bicgstab(s, ctx->kw, op, ctx->rhs_m.p, ctx->dx_m.p, hinted(ko, hint), si);
    note_solve(hint, si, ko, ctx->kw.last_target);
    inf.krylov_iterations_momentum += si.iterations;
    if (rc == NSFEM_ERR_BREAKDOWN) throw Error(rc, "BiCGStab breakdown in the monolithic step");
    if (rc == NSFEM_ERR_NOT_CONVERGED)
      throw Error(rc, "BiCGStab did not converge in the monolithic step");
    double* u = ctx->state[NSFEM_U0].p;
    double* p = ctx->state[NSFEM_P].p;
    launch_axpby(s, nv, 1.0, u, -1.0, ctx->dx_m.p, u);
    launch_axpby(s, np, 1.0, p, -1.0, ctx->dx_m.p + nv, p);
    ++it;
    r = bdf_residual(ctx);
    inf.newton_residuals[it] = r;
    if (!std::isfinite(r)) throw Error(NSFEM_ERR_BREAKDOWN, "Newton residual is not finite");
    converged = (r / r0 < opts->newton_rtol) || (r < opts->newton_atol);
  }
  inf.newton_iterations = it;
  inf.converged = converged ? 1 : 0;
  ctx->mf_active = false;
  ctx->picard = false;
  if (!converged && !opts->allow_nonconvergence)
    throw Error(NSFEM_ERR_NOT_CONVERGED, "Newton solver did not converge");
  ctx->assembled_system = -1;
  API_END(ctx)
}

extern "C" int nsfem_advance(nsfem_ctx* ctx, int scheme) {
  API_BEGIN
  NSFEM_REQUIRE(ctx, "null context");
  hipStream_t s = ctx->stream;
  // u2 <- u1 (pointer swap), u1 <- u0 (copy; u0 keeps its value as in the reference)
  std::swap(ctx->state[NSFEM_U2].p, ctx->state[NSFEM_U1].p);
  NSFEM_HIP(hipMemcpyAsync(ctx->state[NSFEM_U1].p, ctx->state[NSFEM_U0].p,
                           sizeof(double) * nvel(ctx), hipMemcpyDeviceToDevice, s));
  if (scheme == 0) {
    // (p_(n-1) is kept for the optional extrapolated start vector of the projection step)
    std::swap(ctx->state[NSFEM_P2_OLD].p, ctx->state[NSFEM_P_OLD].p);
    NSFEM_HIP(hipMemcpyAsync(ctx->state[NSFEM_P_OLD].p, ctx->state[NSFEM_P].p,
                             sizeof(double) * npre(ctx), hipMemcpyDeviceToDevice, s));
    if (ctx->pressure_history < 2) ctx->pressure_history++;
  } else {
    std::swap(ctx->state[NSFEM_P2_OLD].p, ctx->state[NSFEM_P_OLD].p);
    NSFEM_HIP(hipMemcpyAsync(ctx->state[NSFEM_P_OLD].p, ctx->state[NSFEM_P].p,
                             sizeof(double) * npre(ctx), hipMemcpyDeviceToDevice, s));
  }
  // (no synchronisation: everything that reads the state is ordered on the context's stream, nsfem_get_state and
  // nsfem_synchronize wait for it -- a host wait here left the GPU idle ~20 us in every time step)
  API_END(ctx)
}

// Post-processing Poisson solve on the P1 space:  (grad phi, grad psi) = rhs  with phi = 0 on the
// given dofs (velocity potential of source/ns_problem.py:105-176); Jacobi-CG, the singular pure
// Neumann case is handled by mean projection.  Not on the per-step path.
extern "C" int nsfem_poisson_solve(nsfem_ctx* ctx, const double* rhs, int64_t n_dirichlet,
                                   const int32_t* dofs, double* x, const nsfem_krylov_opts* opts,
                                   nsfem_solve_info* info) {
  nsfem_solve_info local;
  API_BEGIN
  NSFEM_REQUIRE(ctx && rhs && x && opts && (n_dirichlet == 0 || dofs), "null argument");
  NSFEM_REQUIRE(!ctx->distributed(), "post-processing solves are single-context");
  nsfem_solve_info& inf = info ? *info : local;
  hipStream_t s = ctx->stream;
  const int64_t n = npre(ctx);
  std::vector<uint8_t> hm((size_t)n, 0);
  std::vector<double> hb(rhs, rhs + n);
  for (int64_t i = 0; i < n_dirichlet; ++i) {
    NSFEM_REQUIRE(dofs[i] >= 0 && dofs[i] < n, "Dirichlet dof out of range");
    hm[dofs[i]] = 1;
    hb[dofs[i]] = 0.0;
  }
  DevBuf<uint8_t> mask;
  DevBuf<double> db, dx, dinv;
  mask.upload(hm, s);
  db.upload(hb, s);
  dx.alloc((size_t)n);
  dx.zero(s);
  dinv.alloc((size_t)n);
  LinOp op;
  op.A = &ctx->Ap;
  op.nv = 1;
  op.rowmask = mask.p;
  op.maskmode = MASK_ZERO;
  launch_inv_diag(s, ctx->Ap, 1, mask.p, dinv.p);
  op.dinv = dinv.p;
  op.n_global = n;
  int rc = pcg(s, ctx->kw, op, db.p, dx.p, *opts, inf, n_dirichlet == 0);
  if (rc != NSFEM_OK) throw Error(rc, "CG failed in the post-processing Poisson solve");
  NSFEM_HIP(hipMemcpyAsync(x, dx.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  NSFEM_HIP(hipStreamSynchronize(s));
  API_END(ctx)
}

extern "C" int nsfem_mass_solve(nsfem_ctx* ctx, int field, const double* b, double* x,
                                const nsfem_krylov_opts* opts, nsfem_solve_info* info) {
  nsfem_solve_info local;
  API_BEGIN
  NSFEM_REQUIRE(ctx && b && x && opts, "null argument");
  NSFEM_REQUIRE(field == NSFEM_VELOCITY || field == NSFEM_PRESSURE, "bad field");
  nsfem_solve_info& inf = info ? *info : local;
  hipStream_t s = ctx->stream;
  const bool vel = field == NSFEM_VELOCITY;
  const int64_t n = vel ? nvel(ctx) : npre(ctx);
  DevBuf<double> db, dx, dinv;
  db.alloc((size_t)n);
  dx.alloc((size_t)n);
  dinv.alloc((size_t)n);
  NSFEM_HIP(hipMemcpyAsync(db.p, b, sizeof(double) * n, hipMemcpyHostToDevice, s));
  dx.zero(s);
  LinOp op;
  op.A = vel ? &ctx->M2 : &ctx->Mp;
  op.nv = vel ? ctx->mesh.dim : 1;
  launch_inv_diag(s, *op.A, op.nv, nullptr, dinv.p);
  op.dinv = dinv.p;
  int rc = pcg(s, ctx->kw, op, db.p, dx.p, *opts, inf, false);
  if (rc != NSFEM_OK) throw Error(rc, "CG failed in the mass (projection) solve");
  NSFEM_HIP(hipMemcpyAsync(x, dx.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  NSFEM_HIP(hipStreamSynchronize(s));
  API_END(ctx)
}

extern "C" int nsfem_shift_mean_pressure(nsfem_ctx* ctx, double target, double* mean_before) {
  API_BEGIN
  NSFEM_REQUIRE(ctx, "null context");
  hipStream_t s = ctx->stream;
  const int64_t np = npre(ctx);
  // int p dx = 1^T M_p p ; partitioned meshes: owned rows only (ghost rows are the neighbours'),
  // all-reduced, divided by the GLOBAL measure of the domain
  ctx->kw.ensure(nvel(ctx));
  double* parts = ctx->kw.parts.p + 5 * kParts;
  const bool dist = ctx->distributed();
  double area = ctx->area;
  launch_axpby(s, np, 0.0, ctx->tmp_p.p, 0.0, nullptr, ctx->rhs_p.p);
  launch_add_scalar(s, np, 1.0, ctx->rhs_p.p);                     // rhs_p = 1
  if (dist) {
    if (!(ctx->area_global > 0.0)) {
      launch_spmv(s, ctx->Mp, 1, ctx->rhs_p.p, ctx->tmp_p.p, nullptr, MASK_NONE);
      if (ctx->ghost_p.p) launch_zero_ghost(s, np, ctx->mask_p.p, ctx->tmp_p.p);
      launch_dot(s, np, ctx->tmp_p.p, ctx->rhs_p.p, parts);
      ctx->comm->allreduce_sum(s, parts, kParts);
      ctx->area_global = host_sum_parts(s, ctx->kw, 5);
    }
    area = ctx->area_global;
    ctx->comm->exchange(s, ctx->halo_p1, ctx->state[NSFEM_P].p, 1);
  }
  launch_spmv(s, ctx->Mp, 1, ctx->state[NSFEM_P].p, ctx->tmp_p.p, nullptr, MASK_NONE);
  if (dist && ctx->ghost_p.p) launch_zero_ghost(s, np, ctx->mask_p.p, ctx->tmp_p.p);
  launch_dot(s, np, ctx->tmp_p.p, ctx->rhs_p.p, parts);
  if (dist) ctx->comm->allreduce_sum(s, parts, kParts);
  const double mean = host_sum_parts(s, ctx->kw, 5) / area;
  if (mean_before) *mean_before = mean;
  launch_add_scalar(s, np, -(mean - target), ctx->state[NSFEM_P].p);
  NSFEM_HIP(hipStreamSynchronize(s));
  API_END(ctx)
}

// mass shift 1/tau of the velocity-block / Schur-complement preconditioners (see ensure_L); only
// the preconditioner changes, never the discrete equations
extern "C" int nsfem_set_preconditioner_shift(nsfem_ctx* ctx, double shift) {
  API_BEGIN
  NSFEM_REQUIRE(ctx, "null context");
  NSFEM_REQUIRE(shift >= 0.0 && std::isfinite(shift), "the preconditioner shift must be >= 0");
  if (shift != ctx->prec_shift) { ctx->L_dirty = true; ctx->graph_epoch++; }
  ctx->prec_shift = shift;
  API_END(ctx)
}

// rotating frame (2D): angular velocity omega and its time derivative at the new time level
extern "C" int nsfem_set_angular_velocity(nsfem_ctx* ctx, double omega, double omega_dot) {
  API_BEGIN
  NSFEM_REQUIRE(ctx, "null context");
  NSFEM_REQUIRE(ctx->mesh.dim == 2, "scalar angular velocity: 2D meshes (use nsfem_set_angular_velocity_3d)");
  NSFEM_REQUIRE(std::isfinite(omega) && std::isfinite(omega_dot), "non-finite angular velocity");
  if (ctx->omega != omega || ctx->omega_dot != omega_dot) ctx->graph_epoch++;
  ctx->omega = omega;
  ctx->omega_dot = omega_dot;
  API_END(ctx)
}

// rotating frame (3D): angular velocity vector and its time derivative
extern "C" int nsfem_set_angular_velocity_3d(nsfem_ctx* ctx, const double omega[3], const double omega_dot[3]) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && omega && omega_dot, "null argument");
  NSFEM_REQUIRE(ctx->mesh.dim == 3, "vector angular velocity: 3D meshes only");
  for (int a = 0; a < 3; ++a) {
    NSFEM_REQUIRE(std::isfinite(omega[a]) && std::isfinite(omega_dot[a]), "non-finite angular velocity");
    if (ctx->omega3[a] != omega[a] || ctx->omega_dot3[a] != omega_dot[a]) ctx->graph_epoch++;
    ctx->omega3[a] = omega[a];
    ctx->omega_dot3[a] = omega_dot[a];
  }
  API_END(ctx)
}

// max over cells of the DG2-projected local CFL number  2 |u| k / h  (reference
// source/ns_problem.py:554-587) of velocity slot `slot`; one kernel + a 256-value read-back
extern "C" int nsfem_cfl_number(nsfem_ctx* ctx, int slot, double step_size, double* cfl) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && cfl, "null argument");
  NSFEM_REQUIRE(slot >= 0 && slot < NSFEM_N_SLOTS && slot_size(ctx, slot) == nvel(ctx),
                "not a velocity slot");
  hipStream_t s = ctx->stream;
  const int n_parts = 256;
  ctx->kw.ensure(nvel(ctx));
  double* parts = ctx->kw.parts.p + 5 * kParts;
  if (ctx->mesh.dim == 3) launch_cfl_3d(s, ctx->mesh, ctx->state[slot].p, 2.0 * step_size, parts, n_parts);
  else launch_cfl(s, ctx->mesh, ctx->state[slot].p, 2.0 * step_size, parts, n_parts);
  if (ctx->distributed()) ctx->comm->allreduce_max(s, parts, n_parts);
  double h[256];
  NSFEM_HIP(hipMemcpyAsync(h, parts, sizeof(h), hipMemcpyDeviceToHost, s));
  NSFEM_HIP(hipStreamSynchronize(s));
  double m = 0.0;
  for (double v : h) m = std::max(m, v);
  *cfl = m;
  API_END(ctx)
}

// Surface force, mass flux and measure of a set of boundary facets (csrc/boundary.hip): one thread
// per facet, per-facet values summed here in facet order.
extern "C" int nsfem_boundary_force(nsfem_ctx* ctx, int velocity_slot, int pressure_slot,
                                    int32_t n_facets, const int32_t* facet_cell,
                                    const int32_t* facet_local, double nu, double sym, double* out) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && out && (n_facets == 0 || (facet_cell && facet_local)), "null argument");
  NSFEM_REQUIRE(velocity_slot >= 0 && velocity_slot < NSFEM_N_SLOTS && slot_size(ctx, velocity_slot) == nvel(ctx),
                "not a velocity slot");
  NSFEM_REQUIRE(pressure_slot >= 0 && pressure_slot < NSFEM_N_SLOTS && slot_size(ctx, pressure_slot) == npre(ctx),
                "not a pressure slot");
  const int dim = ctx->mesh.dim, w = dim + 2;
  for (int k = 0; k < w; ++k) out[k] = 0.0;
  if (n_facets == 0) return NSFEM_OK;
  for (int32_t f = 0; f < n_facets; ++f) {
    NSFEM_REQUIRE(facet_cell[f] >= 0 && facet_cell[f] < ctx->mesh.n_cells, "facet cell out of range");
    NSFEM_REQUIRE(facet_local[f] >= 0 && facet_local[f] <= dim, "local facet index out of range");
  }
  hipStream_t s = ctx->stream;
  DevBuf<int32_t> dc, dl;
  DevBuf<double> dout;
  dc.upload(facet_cell, (size_t)n_facets, s);
  dl.upload(facet_local, (size_t)n_facets, s);
  dout.alloc((size_t)n_facets * w);
  launch_boundary_force(s, ctx->mesh, n_facets, dc.p, dl.p, ctx->state[velocity_slot].p,
                        ctx->state[pressure_slot].p, nu, sym, dout.p);
  std::vector<double> h((size_t)n_facets * w);
  NSFEM_HIP(hipMemcpyAsync(h.data(), dout.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, s));
  NSFEM_HIP(hipStreamSynchronize(s));
  for (int32_t f = 0; f < n_facets; ++f)
    for (int k = 0; k < w; ++k) out[k] += h[(size_t)f * w + k];
  API_END(ctx)
}

// ----------------------------------------------------------- operator access
static const BlockMat* get_op(nsfem_ctx* c, int op, int* nv_apply) {
  *nv_apply = 1;
  switch (op) {
    case NSFEM_OP_MASS_P2: return &c->M2;
    case NSFEM_OP_STIFF_P2: return &c->K2;
    case NSFEM_OP_STIFF_P1: return &c->Ap;
    case NSFEM_OP_MASS_P1: return &c->Mp;
    case NSFEM_OP_DIV: return &c->Dv;
    case NSFEM_OP_GRAD: return &c->Gr;
    case NSFEM_OP_DIVT: return &c->DT;
    case NSFEM_OP_MOMENTUM_JAC: return &c->J;
    case NSFEM_OP_VISCOUS_EXTRA:
      NSFEM_REQUIRE(c->have_E, "traction-form block not assembled (nsfem_set_viscous_form)");
      return &c->E;
    default: throw Error(NSFEM_ERR_ARG, "unknown operator id");
  }
}

extern "C" int nsfem_operator_shape(nsfem_ctx* ctx, int op, int64_t* n_rows, int64_t* n_cols,
                                    int64_t* nnz_scalar) {
  API_BEGIN
  NSFEM_REQUIRE(ctx, "null context");
  int nv;
  const BlockMat* A = get_op(ctx, op, &nv);
  if (n_rows) *n_rows = (int64_t)A->pat->n_rows * A->br;
  if (n_cols) *n_cols = (int64_t)A->pat->n_cols * A->bc;
  if (nnz_scalar) *nnz_scalar = (int64_t)A->pat->nnz * A->br * A->bc;
  API_END(ctx)
}

extern "C" int nsfem_operator_export(nsfem_ctx* ctx, int op, int32_t* rowptr, int32_t* col,
                                     double* val) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && rowptr && col && val, "null argument");
  int nv;
  const BlockMat* A = get_op(ctx, op, &nv);
  const Pattern& p = *A->pat;
  const int br = A->br, bc = A->bc;
  std::vector<double> v((size_t)p.nnz * br * bc);
  NSFEM_HIP(hipMemcpyAsync(v.data(), A->vals.p, sizeof(double) * v.size(), hipMemcpyDeviceToHost,
                           ctx->stream));
  NSFEM_HIP(hipStreamSynchronize(ctx->stream));
  int64_t pos = 0;
  for (int R = 0; R < p.n_rows; ++R)
    for (int r = 0; r < br; ++r) {
      rowptr[(size_t)R * br + r] = (int32_t)pos;
      for (int k = p.h_rowptr[R]; k < p.h_rowptr[R + 1]; ++k)
        for (int cc = 0; cc < bc; ++cc) {
          col[pos] = p.h_col[k] * bc + cc;
          val[pos] = v[(size_t)k * br * bc + r * bc + cc];
          ++pos;
        }
    }
  rowptr[(size_t)p.n_rows * br] = (int32_t)pos;
  API_END(ctx)
}

// diagonal of a square scalar operator (one value per block row: 1 x 1 blocks) without exporting the matrix -- the
// algebraic Schur Laplacian needs diag(M_v) of a 1.3e8-entry mass matrix
extern "C" int nsfem_operator_diagonal(nsfem_ctx* ctx, int op, double* out) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && out, "null argument");
  int nv;
  const BlockMat* A = get_op(ctx, op, &nv);
  const Pattern& p = *A->pat;
  NSFEM_REQUIRE(A->br == 1 && A->bc == 1 && p.n_rows == p.n_cols && p.diag.p, "diagonal: square scalar operators only");
  DevBuf<double> d;
  d.alloc((size_t)p.n_rows);
  launch_gather_diag(ctx->stream, p.n_rows, p.diag.p, A->vals.p, d.p);
  NSFEM_HIP(hipMemcpyAsync(out, d.p, sizeof(double) * (size_t)p.n_rows, hipMemcpyDeviceToHost, ctx->stream));
  NSFEM_HIP(hipStreamSynchronize(ctx->stream));
  API_END(ctx)
}

extern "C" int nsfem_operator_apply(nsfem_ctx* ctx, int op, const double* x, double* y) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && x && y, "null argument");
  if (op == NSFEM_OP_MOMENTUM_JAC_MF) {      // matrix-free Jacobian at u = USTAR (parity tests)
    hipStream_t s = ctx->stream;
    const size_t n = (size_t)nvel(ctx);
    DevBuf<double> dx, dy;
    dx.alloc(n);
    dy.alloc(n);
    ensure_L(ctx);
    NSFEM_HIP(hipMemcpyAsync(dx.p, x, sizeof(double) * n, hipMemcpyHostToDevice, s));
    ctx->mom_mf.c = ctx;
    ctx->mom_mf.n = (int64_t)n;
    ctx->mom_mf.vel_slot = NSFEM_USTAR;
    ctx->mom_mf.apply(s, dx.p, dy.p);
    NSFEM_HIP(hipMemcpyAsync(y, dy.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    NSFEM_HIP(hipStreamSynchronize(s));
    return NSFEM_OK;
  }
  int nv;
  const BlockMat* A = get_op(ctx, op, &nv);
  const Pattern& p = *A->pat;
  const size_t nx = (size_t)p.n_cols * A->bc, ny = (size_t)p.n_rows * A->br;
  DevBuf<double> dx, dy;
  dx.alloc(nx);
  dy.alloc(ny);
  hipStream_t s = ctx->stream;
  NSFEM_HIP(hipMemcpyAsync(dx.p, x, sizeof(double) * nx, hipMemcpyHostToDevice, s));
  launch_spmv(s, *A, 1, dx.p, dy.p, nullptr, MASK_NONE);
  NSFEM_HIP(hipMemcpyAsync(y, dy.p, sizeof(double) * ny, hipMemcpyDeviceToHost, s));
  NSFEM_HIP(hipStreamSynchronize(s));
  API_END(ctx)
}

// Test hook: one product / residual / Chebyshev smoothing sequence of a scalar lattice operator
// a M + b K (P2 or P1 space of the fine mesh) through a CHOSEN kernel family, on host data.  Lets the
// parity tests pin every SpMV kernel family and every epilogue to the oracle's matrices directly.
extern "C" int nsfem_kernel_apply(nsfem_ctx* ctx, nsfem_kernel_test* t) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && t && t->x && t->y, "null argument");
  NSFEM_REQUIRE(t->nv >= 1 && t->nv <= 3 && t->steps >= 0 && t->steps <= 8, "bad kernel test description");
  hipStream_t s = ctx->stream;
  const bool p2 = t->space == 0;
  const Pattern& pat = p2 ? ctx->p22 : ctx->p11;
  const BlockMat& Mm = p2 ? ctx->M2 : ctx->Mp;
  const BlockMat& Kk = p2 ? ctx->K2 : ctx->Ap;
  StencilDict& dict = p2 ? ctx->dict22 : ctx->dict11;
  bool& tried = p2 ? ctx->dict22_tried : ctx->dict11_tried;
  bool have_dict = dict.n_stencils > 0;
  if (!tried) {
    tried = true;
    have_dict = build_stencil_dict(s, pat, Mm.vals.p, Kk.vals.p, dict);
  }
  BlockMat T;
  T.init(&pat, 1, 1, s);
  launch_scale_combine(s, pat.nnz, t->a, Mm.vals.p, t->b_coef, Kk.vals.p, T.vals.p);
  if (have_dict) T.dict = &dict;
  T.sell_update(s);
  const int nv = t->nv;
  const size_t n = (size_t)pat.n_rows * nv;
  DevBuf<double> x, b, d, d2, y, y2, r, dinv;
  DevBuf<uint8_t> mask;
  x.upload(t->x, n, s);
  y.alloc(n); y.zero(s);
  y2.alloc(n); y2.zero(s);
  r.alloc(n); r.zero(s);
  b.alloc(n); b.zero(s);
  d.alloc(n); d.zero(s);
  d2.alloc(n); d2.zero(s);
  if (t->b) NSFEM_HIP(hipMemcpyAsync(b.p, t->b, sizeof(double) * n, hipMemcpyHostToDevice, s));
  if (t->d) NSFEM_HIP(hipMemcpyAsync(d.p, t->d, sizeof(double) * n, hipMemcpyHostToDevice, s));
  if (t->mask) mask.upload(t->mask, n, s);
  const uint8_t* mk = t->mask ? mask.p : nullptr;
  int family = t->family;
  if (family == 1) { T.dict_ready = false; T.sell_ready = false; }
  else if (family == 2) { T.dict_ready = false; NSFEM_REQUIRE(T.sell_ready, "no SELL-64 layout for this pattern"); }
  else if (family == 3) { NSFEM_REQUIRE(T.dict_ready, "no stencil dictionary for this pattern"); T.sell_ready = false; }
  else if (family == 4) { NSFEM_REQUIRE(lattice_smoother_available(T, nv), "no lattice structure for this pattern"); }
  // what the dispatcher will pick
  auto picked = [&](bool dict_ok) {
    if (T.dict_ready && (dict_ok || T.dict->exact) && nv <= 3) return 3;
    if (T.sell_ready && pat.n_slices > 0) return 2;
    return 1;
  };
  const double* result = y.p;
  if (t->epilogue == 0) {
    t->used_family = picked(t->dict_ok != 0);
    launch_spmv(s, T, nv, x.p, y.p, mk, mk ? t->maskmode : MASK_NONE, t->ghost, 0, t->dict_ok);
  } else if (t->epilogue == 1) {
    NSFEM_REQUIRE(t->b, "residual needs b");
    t->used_family = picked(false);
    launch_residual(s, T, nv, x.p, b.p, y.p, mk, mk ? t->maskmode : MASK_NONE);
  } else {
    NSFEM_REQUIRE(t->b && t->steps >= 1, "smoothing needs b and steps >= 1");
    if (family == 4) {
      t->used_family = 4;
      NSFEM_REQUIRE(t->steps <= lattice_smoother_max_steps(T, t->from_zero != 0, t->with_residual != 0),
                    "too many steps for one launch of the lattice kernel");
      launch_cheb_lattice(s, T, nv, t->from_zero ? nullptr : x.p, b.p, (t->d && !t->from_zero) ? d.p : nullptr, y.p,
                          d2.p, t->with_residual ? r.p : nullptr, mk, t->steps, t->c1, t->c2, t->ident);
      NSFEM_HIP(hipMemcpyAsync(d.p, d2.p, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
    } else {
      t->used_family = picked(true);
      dinv.alloc(n);
      launch_inv_diag(s, T, nv, mk, dinv.p);
      const double* cur = x.p;
      if (t->from_zero) { x.zero(s); d.zero(s); }
      for (int k = 0; k < t->steps; ++k) {
        double* out = cur == y.p ? y2.p : y.p;
        launch_cheb_step(s, T, nv, cur, b.p, dinv.p, d.p, t->c1[k], t->c2[k], out, mk, t->ghost, 0,
                         t->ident && k == t->steps - 1 ? 1 : 0);
        cur = out;
      }
      result = cur;
      if (t->with_residual) launch_residual(s, T, nv, cur, b.p, r.p, mk, MASK_ZERO);
    }
  }
  NSFEM_HIP(hipMemcpyAsync(t->y, result, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  if (t->d_out) NSFEM_HIP(hipMemcpyAsync(t->d_out, d.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  if (t->r_out) NSFEM_HIP(hipMemcpyAsync(t->r_out, r.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  NSFEM_HIP(hipStreamSynchronize(s));
  t->dict_entries = have_dict ? dict.n_stencils : 0;
  t->dict_exact = have_dict && dict.exact ? 1 : 0;
  t->lattice_w = have_dict ? dict.lat_w : 0;
  API_END(ctx)
}

// Factors of the fast diagonalisation of the pressure Poisson operator (poisson_fd.factors): the P1 space must be
// the W x H lattice in lexicographic numbering, the pressure Dirichlet set a union of whole sides (or empty).
// Krylov option precond = 3 of the projection step then solves it directly (four dense products on the matrix cores).
extern "C" int nsfem_poisson_set_fast_diag(nsfem_ctx* ctx, int32_t W, int32_t H, const double* Vx, const double* Vy,
                                           const double* inv) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && Vx && Vy && inv, "null argument");
  NSFEM_REQUIRE((int64_t)W * H == npre(ctx), "fast diagonalisation: W x H must be the number of pressure dofs");
  NSFEM_REQUIRE(!ctx->distributed(), "partitioned context: nsfem_poisson_set_fast_diag_rows");
  ctx->fd_p.set(ctx->stream, W, H, Vx, Vy, inv);
  API_END(ctx)
}

// Partitioned strips: the factors of the GLOBAL W x H lattice; this rank's P1 space is the lattice lines
// first_line ... first_line + n_p1 / W - 1 (ghost lines included).  The fused step driver then solves the projection
// step with one all-reduce of H x W doubles (FastDiag::apply_strip).
extern "C" int nsfem_poisson_set_fast_diag_rows(nsfem_ctx* ctx, int32_t W, int32_t H, int32_t first_line,
                                                const double* Vx, const double* Vy, const double* inv) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && Vx && Vy && inv && W >= 2, "null argument");
  NSFEM_REQUIRE(ctx->distributed(), "strip factors need a partitioned context (nsfem_set_partition)");
  const int64_t np = npre(ctx);
  NSFEM_REQUIRE(np % W == 0 && first_line >= 0 && first_line + np / W <= H,
                "fast diagonalisation: the local pressure space is not a run of whole lattice lines");
  NSFEM_REQUIRE((int64_t)W * H == ctx->n_p1_global, "fast diagonalisation: W x H must be the global number of pressure dofs");
  ctx->fd_p.set_rows(ctx->stream, W, H, first_line, (int)(np / W), Vx, Vy, inv);
  API_END(ctx)
}

// Test hook: one application z = M^-1 r of a multigrid preconditioner on host vectors -- which = 0 pressure Poisson
// hierarchy (mg_p), 1 velocity hierarchy (mg_v, the identity rows of the Newton preconditioner included).  Lets the
// parity tests compare the fused multi-level launches (mglegs.hip) with the separate launches cycle by cycle.
extern "C" int nsfem_mg_apply(nsfem_ctx* ctx, int which, const double* r, double* z) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && r && z && (which == 0 || which == 1 || which == 2), "bad argument");
  hipStream_t s = ctx->stream;
  if (which == 2) {                    // the fast-diagonalisation Poisson solve z = A^+ r
    const int64_t n = npre(ctx);
    DevBuf<double> dr, dz;
    dr.upload(r, (size_t)n, s);
    dz.alloc((size_t)n);
    ctx->fd_p.apply(s, dr.p, dz.p);
    NSFEM_HIP(hipMemcpyAsync(z, dz.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    NSFEM_HIP(hipStreamSynchronize(s));
    return NSFEM_OK;
  }
  NSFEM_REQUIRE(ctx->mg_built, "no multigrid hierarchy (nsfem_mg_finalize)");
  ensure_L(ctx);                       // (builds the dictionaries of the fine P1 operators as well)
  if (which == 0 && ctx->mg_p.legs_kind == 0) ctx->mg_p_dirty = true;   // planned before the dictionaries existed
  mg_refresh(ctx, which == 1);
  const int64_t n = which == 1 ? nvel(ctx) : npre(ctx);
  DevBuf<double> dr, dz;
  dr.upload(r, (size_t)n, s);
  dz.alloc((size_t)n);
  dz.zero(s);
  (which == 1 ? ctx->mg_v : ctx->mg_p).apply(s, dr.p, dz.p);
  NSFEM_HIP(hipMemcpyAsync(z, dz.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  NSFEM_HIP(hipStreamSynchronize(s));
  API_END(ctx)
}

// How a multigrid cycle runs: out = {kind of the fused legs (0 separate launches, 1 one launch below the finest level
// of a truncated cycle, 2 down-legs + single-workgroup tail + up-legs), launches of k_mg_leg per cycle, levels in use,
// launches of k_mg_leg so far}
extern "C" int nsfem_mg_info(nsfem_ctx* ctx, int which, int64_t out[4]) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && out && which >= 0 && which <= 3, "bad argument");
  NSFEM_REQUIRE(ctx->mg_built, "no multigrid hierarchy (nsfem_mg_finalize)");
  ensure_L(ctx);
  mg_refresh(ctx, (which & 1) == 1);
  Multigrid& mg = (which & 1) ? ctx->mg_v : ctx->mg_p;
  if (which & 2) {                     // the multi-step lattice kernel on this hierarchy (strips: relaxed halo mode)
    const size_t n_used = mg.truncated() ? mg.active : mg.lv.size();
    int64_t levels = 0;
    for (size_t l = 0; l < n_used; ++l) levels += (mg.lattice_ok(mg.lv[l]) || mg.lattice_ok_relaxed(mg.lv[l])) ? 1 : 0;
    out[0] = levels;
    out[1] = mg.lattice_launches;
    out[2] = (int64_t)n_used;
    out[3] = mg.lv[0].ghost_lo >= 0 ? mg.lv[0].ghost_lo * 256 + mg.lv[0].ghost_hi : 0;
    return NSFEM_OK;
  }
  if (mg.legs_kind < 0) mg.build_legs(ctx->stream);
  out[0] = mg.legs_kind;
  out[1] = mg.legs_kind == 1 ? 1 : (mg.legs_kind == 2 ? (int64_t)(mg.legs_down.size() + mg.legs_up.size() + 1) : 0);
  out[2] = (int64_t)(mg.truncated() ? mg.active : mg.lv.size());
  out[3] = mg.leg_launches;
  API_END(ctx)
}

// In-situ timing of the dominant kernel: while enabled, every finest-level Chebyshev smoothing
// launch of the velocity multigrid (k_spmv_stream<1,1,dim,EPI_CHEB>) is bracketed by a HIP-event
// pair on the context's stream.  enable != 0: start (discarding earlier samples); enable == 0:
// stop and return the average launch duration, the number of launches and the algorithmic bytes
// per launch.
// algorithmic bytes of one Chebyshev smoothing launch with operator A on nv interleaved vectors.
// Vectors: x, b, dinv, d read; d, x_out written (8 bytes each) + 1 mask byte per entry.  Matrix: a
// CSR stream (12 bytes per nonzero + row pointers) or -- stencil-dictionary copy -- one byte per
// row + the dictionary once.
static int64_t smoother_launch_bytes(const BlockMat& A, int nv, bool csr_equivalent) {
  const Pattern& p = *A.pat;
  const int64_t n = (int64_t)p.n_rows * nv;
  if (A.dict_ready && !csr_equivalent)     // (no dinv stream: 1 / diagonal comes from the dictionary)
    return (int64_t)p.n_rows + (int64_t)A.dict->n_stencils * (A.dict->lmax * 12 + 4) + n * (5 * 8 + 1);
  return (int64_t)p.nnz * 12 + (int64_t)(p.n_rows + 1) * 4 + n * (6 * 8 + 1);
}

extern "C" int nsfem_profile_smoother(nsfem_ctx* ctx, int enable, double* avg_ms, int64_t* launches,
                                      int64_t* algorithmic_bytes) {
  API_BEGIN
  NSFEM_REQUIRE(ctx, "null context");
  NSFEM_REQUIRE(ctx->mg_built, "no multigrid hierarchy (nsfem_mg_finalize)");
  Multigrid& mg = ctx->mg_v;
  if (enable) {
    if (mg.prof_ev.empty()) {
      mg.prof_ev.resize(8192);
      for (hipEvent_t& e : mg.prof_ev) NSFEM_HIP(hipEventCreate(&e));
    }
    mg.prof_n = 0;
    mg.prof_launches = 0;
    mg.prof_steps = 0;
    mg.prof_bytes = 0;
    mg.prof = true;
    ctx->kw.graphs_suspended = true;     // (HIP events are recorded inside the iteration bodies)
    ctx->kw.clear_graphs();
    return NSFEM_OK;
  }
  mg.prof = false;
  ctx->kw.graphs_suspended = ctx->conv_probe.on;
  NSFEM_HIP(hipStreamSynchronize(ctx->stream));
  double total = 0.0;
  for (size_t i = 0; i + 1 < mg.prof_n; i += 2) {
    float t = 0.f;
    NSFEM_HIP(hipEventElapsedTime(&t, mg.prof_ev[i], mg.prof_ev[i + 1]));
    total += t;
  }
  const int64_t n_launch = mg.prof_launches;
  if (avg_ms) *avg_ms = n_launch ? total / (double)n_launch : 0.0;
  if (launches) *launches = n_launch;
  // (lattice kernel: launches of different shapes -- with / without the carried direction -- were
  // timed; the average over exactly those launches)
  if (algorithmic_bytes)
    *algorithmic_bytes = (mg.prof_bytes > 0 && n_launch > 0) ? mg.prof_bytes / n_launch
                                                             : smoother_launch_bytes(*mg.lv[0].A, mg.nv, false);
  API_END(ctx)
}

// detail of the last nsfem_profile_smoother window: out = {launches, smoothing steps they ran,
// algorithmic bytes they moved in total, 1 when the multi-step lattice kernel ran them}
extern "C" int nsfem_profile_smoother_detail(nsfem_ctx* ctx, int64_t out[4]) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && out, "null argument");
  const Multigrid& mg = ctx->mg_v;
  out[0] = mg.prof_launches;
  out[1] = mg.prof_steps;
  out[2] = mg.prof_bytes > 0 ? mg.prof_bytes
                             : (mg.lv.empty() ? 0 : mg.prof_launches * smoother_launch_bytes(*mg.lv[0].A, mg.nv, false));
  out[3] = mg.prof_bytes > 0 ? 1 : 0;
  API_END(ctx)
}

// which finest-level smoothing kernel of the velocity multigrid runs, and what a CSR stream of the
// same operator would move: out = {kind (0 CSR-stream, 1 SELL-64, 2 stencil dictionary, 3 stencil
// dictionary inside the multi-step lattice kernel),
// dictionary entries, longest row, CSR-equivalent algorithmic bytes of one smoothing launch}
extern "C" int nsfem_smoother_info(nsfem_ctx* ctx, int64_t out[4]) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && out, "null argument");
  NSFEM_REQUIRE(ctx->mg_built, "no multigrid hierarchy (nsfem_mg_finalize)");
  ensure_L(ctx);
  const Multigrid& mg = ctx->mg_v;
  const BlockMat& A = *mg.lv[0].A;
  Multigrid& mgw = ctx->mg_v;
  out[0] = A.dict_ready ? ((mg.lattice_ok(mg.lv[0]) || mgw.lattice_ok_relaxed(mgw.lv[0])) ? 3 : 2) : (A.sell_ready ? 1 : 0);
  out[1] = A.dict_ready ? A.dict->n_stencils : 0;
  out[2] = A.dict_ready ? (A.dict->exact ? -A.dict->lmax : A.dict->lmax) : 0;
  out[3] = smoother_launch_bytes(A, mg.nv, true);
  API_END(ctx)
}

extern "C" int nsfem_jacobian_info(nsfem_ctx* ctx, int64_t out[4]) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && out, "null argument");
  ensure_L(ctx);
  out[0] = jacobian_path(ctx);
  out[1] = ctx->jac_lattice_launches;
  out[2] = jacobian_lattice_bytes(ctx->mesh);
  out[3] = 0;
  API_END(ctx)
}

// Algorithmic bytes of one matrix-free convection action  y += c_c [d conv(u)/du] x  (SURVEY.md
// section 8d, "assembly of a vector"): per cell the vertex coordinates, the P2 dof ids and the
// nodal values of u and x gathered through them; the output vector read-modify-written once.
static int64_t convection_action_bytes(const nsfem_ctx* c) {
  const int64_t dim = c->mesh.dim, nl1 = dim + 1, nl2 = dim == 2 ? 6 : 10;
  return (int64_t)c->mesh.n_cells * (nl1 * dim * 8 + nl2 * 4 + 2 * nl2 * dim * 8) +
         (int64_t)c->mesh.n_p2 * dim * 16;
}
// one GPU, triangles, lattice mesh: the per-node sums run inside the L-product launch (MomentumMF::apply); the
// probe then brackets the element kernel alone, whose bytes are the per-cell part plus the element vectors it
// stores (dim * nl2 doubles per cell, summed by the other launch)
static bool convection_gather_fused(const nsfem_ctx* c) {
  return !c->distributed() && c->mesh.dim == 2 && c->L.dict_ready;
}
static int64_t convection_cells_bytes(const nsfem_ctx* c) {
  const int64_t dim = c->mesh.dim, nl1 = dim + 1, nl2 = dim == 2 ? 6 : 10;
  return (int64_t)c->mesh.n_cells * (nl1 * dim * 8 + nl2 * 4 + 2 * nl2 * dim * 8 + nl2 * dim * 8);
}

// In-situ timing of the matrix-free convection action inside the Newton-Krylov solves (one HIP-event
// pair around every k_conv_cell<FORM,LIN> + k_res_gather pair on the context's stream).
extern "C" int nsfem_profile_convection(nsfem_ctx* ctx, int enable, double* avg_ms,
                                        int64_t* applications, int64_t* algorithmic_bytes) {
  API_BEGIN
  NSFEM_REQUIRE(ctx, "null context");
  nsfem_ctx::Probe& pr = ctx->conv_probe;
  if (enable) {
    if (pr.ev.empty()) {
      pr.ev.resize(4096);
      for (hipEvent_t& e : pr.ev) NSFEM_HIP(hipEventCreate(&e));
    }
    pr.n = 0;
    pr.on = true;
    ctx->kw.graphs_suspended = true;     // (HIP events are recorded inside the iteration bodies)
    ctx->kw.clear_graphs();
    return NSFEM_OK;
  }
  pr.on = false;
  ctx->kw.graphs_suspended = ctx->mg_v.prof;
  NSFEM_HIP(hipStreamSynchronize(ctx->stream));
  double total = 0.0;
  for (size_t i = 0; i + 1 < pr.n; i += 2) {
    float t = 0.f;
    NSFEM_HIP(hipEventElapsedTime(&t, pr.ev[i], pr.ev[i + 1]));
    total += t;
  }
  const int64_t n = (int64_t)(pr.n / 2);
  if (avg_ms) *avg_ms = n ? total / (double)n : 0.0;
  if (applications) *applications = n;
  if (algorithmic_bytes)
    *algorithmic_bytes = jacobian_path(ctx) == 2 ? jacobian_lattice_bytes(ctx->mesh)
                         : convection_gather_fused(ctx) ? -convection_cells_bytes(ctx) : convection_action_bytes(ctx);
  API_END(ctx)
}

extern "C" int nsfem_time_spmv(nsfem_ctx* ctx, int op, int reps, double* ms_per_launch,
                               int64_t* algorithmic_bytes) {
  API_BEGIN
  NSFEM_REQUIRE(ctx && reps != 0 && ms_per_launch, "bad argument");
  // reps < 0 (smoother only): |reps| back-to-back launches without the cache-flushing launches in
  // between (the operands stay in the 256 MiB Infinity Cache when they fit)
  const bool warm = reps < 0;
  if (warm) reps = -reps;
  if (op == NSFEM_OP_CONVECTION_ACTION) {
    // matrix-free convection action at u = U1 in the direction x = U0, each repetition preceded by
    // a cache-flushing product with the block Jacobian array (timed separately and subtracted)
    hipStream_t s = ctx->stream;
    const double cc = cc_of(ctx) != 0.0 ? cc_of(ctx) : 1.0;
    DevBuf<double> fx, fy, y;
    fx.alloc((size_t)nvel(ctx));
    fy.alloc((size_t)nvel(ctx));
    y.alloc((size_t)nvel(ctx));
    fx.zero(s);
    y.zero(s);
    auto flush = [&] { launch_spmv(s, ctx->J, 1, fx.p, fy.p, nullptr, MASK_NONE); };
    auto step = [&] {
      launch_convection_action(s, ctx->mesh, ctx->state[NSFEM_U1].p, ctx->state[NSFEM_U0].p, cc, y.p,
                               ctx->conv_form, false);
    };
    hipEvent_t e0, e1;
    NSFEM_HIP(hipEventCreate(&e0));
    NSFEM_HIP(hipEventCreate(&e1));
    auto timed = [&](bool with_step) {
      for (int i = 0; i < 3; ++i) { flush(); if (with_step) step(); }
      NSFEM_HIP(hipEventRecord(e0, s));
      for (int i = 0; i < reps; ++i) { flush(); if (with_step) step(); }
      NSFEM_HIP(hipEventRecord(e1, s));
      NSFEM_HIP(hipEventSynchronize(e1));
      float t = 0.f;
      NSFEM_HIP(hipEventElapsedTime(&t, e0, e1));
      return (double)t;
    };
    const double t_both = timed(true), t_flush = timed(false);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = (t_both - t_flush) / reps;
    if (algorithmic_bytes) *algorithmic_bytes = convection_action_bytes(ctx);
    return NSFEM_OK;
  }
  if (op == NSFEM_OP_MOMENTUM_SMOOTHER) {
    // one Chebyshev-Jacobi smoothing step of the velocity multigrid on its finest level: the
    // scalar P2 operator L applied to the interleaved components with the fused epilogue
    // d = c1 d + c2 dinv (b - L x), y = x + d  (k_spmv_stream<1,1,dim,EPI_CHEB>)
    NSFEM_REQUIRE(ctx->mg_built, "no multigrid hierarchy (nsfem_mg_finalize)");
    mg_refresh(ctx, true);
    hipStream_t s = ctx->stream;
    Multigrid& mg = ctx->mg_v;
    MGLevel& L0 = mg.lv[0];
    const Pattern& p = *L0.A->pat;
    const int nv = mg.nv;
    const bool lattice = mg.lattice_ok(L0);
    const int lat_steps = mg.degree >= 1 && mg.degree <= 3 ? mg.degree : 3;
    auto step = [&] {
      if (lattice) {       // the post-smoothing launch of the cycle: `degree` steps from a given iterate
        const double c1[4] = {0.0, 0.3, 0.3, 0.3}, c2[4] = {0.7, 0.7, 0.7, 0.7};
        launch_cheb_lattice(s, *L0.A, nv, L0.xa.p, L0.r.p, nullptr, L0.xb.p, nullptr, nullptr, L0.mask,
                            lat_steps, c1, c2, 0, L0.sidm_for == (const void*)L0.A->dict ? L0.sidm.p : nullptr);
      } else {
        launch_cheb_step(s, *L0.A, nv, L0.xa.p, L0.r.p, L0.dinv.p, L0.d.p, 0.3, 0.7, L0.xb.p, L0.mask);
      }
    };
    // In the solver this launch follows other kernels that have streamed hundreds of MB; the
    // operator (145 MB at n = 512) would otherwise sit in the 256 MB Infinity Cache between
    // back-to-back repetitions.  Timed: `reps` pairs (flush, step) minus `reps` flushes, the
    // flush being a product with the 2x2-block Jacobian array (0.47 GB streamed, larger than the
    // cache, so its own time does not depend on what ran before).
    DevBuf<double> fx, fy;
    fx.alloc((size_t)nvel(ctx));
    fy.alloc((size_t)nvel(ctx));
    fx.zero(s);
    auto flush = [&] { launch_spmv(s, ctx->J, 1, fx.p, fy.p, nullptr, MASK_NONE); };
    hipEvent_t e0, e1;
    NSFEM_HIP(hipEventCreate(&e0));
    NSFEM_HIP(hipEventCreate(&e1));
    auto timed = [&](bool with_step) {
      for (int i = 0; i < 3; ++i) { flush(); if (with_step) step(); }
      NSFEM_HIP(hipEventRecord(e0, s));
      for (int i = 0; i < reps; ++i) { flush(); if (with_step) step(); }
      NSFEM_HIP(hipEventRecord(e1, s));
      NSFEM_HIP(hipEventSynchronize(e1));
      float t = 0.f;
      NSFEM_HIP(hipEventElapsedTime(&t, e0, e1));
      return (double)t;
    };
    double ms;
    if (warm) {
      for (int i = 0; i < 3; ++i) step();
      NSFEM_HIP(hipEventRecord(e0, s));
      for (int i = 0; i < reps; ++i) step();
      NSFEM_HIP(hipEventRecord(e1, s));
      NSFEM_HIP(hipEventSynchronize(e1));
      float t = 0.f;
      NSFEM_HIP(hipEventElapsedTime(&t, e0, e1));
      ms = (double)t;
    } else {
      const double t_both = timed(true), t_flush = timed(false);
      ms = t_both - t_flush;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = (double)ms / reps;
    (void)p;
    if (algorithmic_bytes)
      *algorithmic_bytes = lattice ? lattice_launch_bytes(*L0.A, nv, false, false, false, false)
                                   : smoother_launch_bytes(*L0.A, nv, false);
    return NSFEM_OK;
  }
  int nv;
  const BlockMat* A = get_op(ctx, op, &nv);
  const Pattern& p = *A->pat;
  // scalar P2 operators act on all velocity components in the solver
  const int nvv = (op == NSFEM_OP_MASS_P2 || op == NSFEM_OP_STIFF_P2) ? ctx->mesh.dim : 1;
  const size_t nx = (size_t)p.n_cols * A->bc * nvv, ny = (size_t)p.n_rows * A->br * nvv;
  DevBuf<double> dx, dy;
  dx.alloc(nx);
  dy.alloc(ny);
  hipStream_t s = ctx->stream;
  std::vector<double> hx(nx);
  for (size_t i = 0; i < nx; ++i) hx[i] = std::sin((double)i);
  NSFEM_HIP(hipMemcpyAsync(dx.p, hx.data(), sizeof(double) * nx, hipMemcpyHostToDevice, s));
  for (int i = 0; i < 3; ++i) launch_spmv(s, *A, nvv, dx.p, dy.p, nullptr, MASK_NONE);
  hipEvent_t e0, e1;
  NSFEM_HIP(hipEventCreate(&e0));
  NSFEM_HIP(hipEventCreate(&e1));
  NSFEM_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) launch_spmv(s, *A, nvv, dx.p, dy.p, nullptr, MASK_NONE);
  NSFEM_HIP(hipEventRecord(e1, s));
  NSFEM_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  NSFEM_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_launch = (double)ms / reps;
  if (algorithmic_bytes)
    *algorithmic_bytes = (int64_t)p.nnz * (8 * A->br * A->bc + 4) + (int64_t)(p.n_rows + 1) * 4 +
                         (int64_t)(nx + ny) * 8;
  API_END(ctx)
}
