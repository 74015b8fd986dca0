#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: DoF-updates/sec and time-steps/sec of the
2D lid-driven cavity, Re = 100, IPCS pressure-projection, Taylor-Hood P2/P1, on MI355X.

N = 1 workload = BASELINE.json configs[1]: 512 x 512 right-diagonal triangles
(2,364,419 dofs), k = 1e-3.  One "step" = one full IPCS time step (Newton diffusion step,
pressure Poisson, velocity correction, time-level shift) on device-resident state.

N > 1 (launched by torch.distributed.run, one process per GPU): WEAK scaling -- every rank
owns a 512 x 512-cell strip of a 512 x (512 N) cavity [0,1] x [0,N] (lid on top), i.e. the
per-GPU work is fixed; the strips are coupled through RCCL (halo exchange of 1-2 lattice
lines per SpMV, all-reduce of the partial dot products; csrc/comm.hip).  The aggregate
`value` is DoF-updates/s = (global dofs) x (time steps/s).

Solver settings of the timed steps (SURVEY.md section 8d "throughput runs"): Newton stops on the
reference's criterion (|F| < 1e-10 or |F|/|F0| < 1e-9, checked on the true nonlinear residual
every iteration), the linear solves inside it are inexact (residual reduced by
--newton-forcing = 1e-4, never below a tenth of the nonlinear target), Poisson / mass solves to
--krylov-rtol = 1e-8.  The same steps with direct-solver accuracy (rtol 1e-12, exact Newton) are
timed right after and reported as config.ms_per_step_with_krylov_rtol_1e-12_exact_newton.

Prints ONE JSON line (driver contract) that also carries
  "roofline":     dominant kernel (block-CSR SpMV of the momentum Jacobian), algorithmic
                  bytes per launch / HIP-event time on the kernel's stream vs 8 TB/s HBM;
  "cpu_baseline": the CPU oracle configured as the reference works (full re-assembly +
                  sparse LU every Newton iteration, Poisson/mass re-factorised every step)
                  timed on a bounded sample on this host (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "navierstokes-with-fenics_amd")
sys.path[:0] = [PKG]

import numpy as np  # noqa: E402

# load the HIP library (and with it ROCm's libamdhip64 / librccl) BEFORE torch is imported
import _native as nat  # noqa: E402
nat.load_library()

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cavity_dirichlet(dm, height):
    """no-slip on left/right/bottom, lid (1, 0) on top (wins at the corners) for the boundary
    nodes this (local) dof map holds; reference: demo/cavity_flow.py:21-29."""
    X = dm.p2_coords
    tol = 1e-12
    on = (np.abs(X[:, 0]) < tol) | (np.abs(X[:, 0] - 1.0) < tol) | (np.abs(X[:, 1]) < tol) | \
        (np.abs(X[:, 1] - height) < tol)
    nodes = np.nonzero(on)[0]
    lid = np.abs(X[nodes, 1] - height) < tol
    dofs = np.concatenate([2 * nodes, 2 * nodes + 1]).astype(np.int32)
    vals = np.concatenate([np.where(lid, 1.0, 0.0), np.zeros(nodes.size)])
    return dofs, vals


def cpu_baseline(n_sample, k, steps):
    """Reference algorithm on the host CPU (1 process, as the reference runs): per Newton
    iteration full re-assembly + SuperLU; Poisson and mass matrices re-assembled and
    re-factorised each step (dolfin LinearVariationalSolver behaviour)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fem_oracle as fo
    from fem_mesh import TaylorHoodDofMap
    from grid_generator import hyper_cube
    mesh, _ = hyper_cube(2, n_sample)
    dm = TaylorHoodDofMap(mesh)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    bd, bv = cavity_dirichlet(dm, 1.0)
    order = np.argsort(bd, kind="stable")
    bd, bv = bd[order].astype(np.int64), bv[order]
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.01, body_force_term=None)
    orc = fo.IPCSOracle(s, coef, refactor_every_step=True)
    orc.step(fo.bdf_alpha(0, 1.0), k, (bd, bv))     # warm-up (BDF-1 start step)
    orc.advance()
    t0 = time.perf_counter()
    for step in range(1, steps + 1):
        orc.step(fo.bdf_alpha(step, 1.0), k, (bd, bv))
        orc.advance()
    dt = time.perf_counter() - t0
    sps = steps / dt
    return {"value": sps * dm.n_dofs, "unit": "DoF-updates/s", "cores": 1, "kind": "port",
            "sample": "oracle IPCS (re-assembly + SuperLU per Newton iteration, Poisson/mass "
                      "re-factorised per step) on the n=%d cavity (%d dofs), %d steps after 1 warm-up: "
                      "%.3f steps/s x %d dofs (sparse LU scales super-linearly, so the rate at "
                      "2.36 M dofs would be lower)" % (n_sample, dm.n_dofs, steps, sps, dm.n_dofs),
            "sample_steps_per_s": sps, "sample_dofs": dm.n_dofs, "host_cpus": os.cpu_count()}


def dfg_bdf_bench(args):
    """BASELINE.json configs[2]: DFG 2D-2 cylinder channel (reference demo/dfg_benchmark.py),
    Re = 100, BDF-2 monolithic scheme, dt = 0.005, curved-boundary refinement hierarchy of the
    in-repo block mesh (m = 4, 5 refinements: 589,824 cells, ~2.65 M dofs).  1 GPU only."""
    import grid_generator as gg
    from fem_mesh import TaylorHoodDofMap
    from multigrid import attach_hierarchy, attach_schur_laplacian
    t_setup = time.perf_counter()
    mesh, marks = gg.dfg_channel(4, args.dfg_refine)
    dm = TaylorHoodDofMap(mesh)
    ctx = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1)
    levels = attach_hierarchy(ctx, mesh, args.mg_degree, args.mg_eig_ratio)
    ids = gg.DFGBoundaryMarkers
    last = {}
    for mid in (ids.inlet.value, ids.bottom.value, ids.top.value, ids.cylinder.value):
        nodes = np.unique(dm.facet_p2_nodes(marks.facets_with_id(mid)))
        y = dm.p2_coords[nodes, 1]
        ux = 6.0 * y / 4.1 * (1 - y / 4.1) if mid == ids.inlet.value else np.zeros_like(y)
        last.update(zip((2 * nodes).tolist(), ux.tolist()))
        last.update(zip((2 * nodes + 1).tolist(), [0.0] * nodes.size))
    bd = np.array(sorted(last), dtype=np.int32)
    bv = np.array([last[d] for d in bd.tolist()])
    ctx.set_coeffs(1.0, 1.0, 1.0 / 100.0)
    ctx.set_dirichlet(nat.VELOCITY, bd, bv)
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    ctx.set_dirichlet(nat.PRESSURE_PRECOND, np.zeros(0, np.int32), np.zeros(0))
    attach_schur_laplacian(ctx, bd)
    t_setup = time.perf_counter() - t_setup
    _apply_truncation(ctx, args)
    opts = ctx.default_step_opts()
    opts.momentum.rtol, opts.momentum.precond, opts.momentum.max_iter = args.krylov_rtol, 1, 500
    opts.newton_forcing = args.newton_forcing
    dt = 0.005

    def one_step(i):
        ctx.set_bdf((1.0, -1.0, 0.0) if i == 0 else (1.5, -2.0, 0.5), dt)
        info = ctx.step_bdf(opts)
        ctx.advance(1)
        return info

    for i in range(args.warmup):
        one_step(i)
    ctx.synchronize()
    t0 = time.perf_counter()
    newton = kry = 0
    for i in range(args.warmup, args.warmup + args.steps):
        info = one_step(i)
        newton += info.newton_iterations
        kry += info.krylov_iterations_momentum
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    sps = args.steps / elapsed
    ms_spmv, nbytes = ctx.time_spmv(nat.OP_MOMENTUM_JAC, 200)
    achieved = nbytes / (ms_spmv * 1e-3) / 1e9
    print(json.dumps({
        "metric": "dof_updates_per_sec", "value": sps * dm.n_dofs, "unit": "DoF-updates/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 / sps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic", "time_steps_per_sec": sps,
        "config": {"workload": "DFG 2D-2 cylinder channel Re=100, %d unstructured triangles (%d dofs), "
                               "BDF-2 monolithic, dt=%g, impulsive start" % (mesh.num_cells(), dm.n_dofs, dt),
                   "n_dofs": dm.n_dofs, "newton_tol": 1e-10, "krylov_rtol": args.krylov_rtol,
                   "newton_forcing": args.newton_forcing,
                   "preconditioner": "block-triangular (V-cycle velocity block, Cahouet-Chabard Schur "
                                     "with algebraic pressure Laplacian), %d coarse P1 levels" % levels,
                   "parallelism": "1 GPU", "newton_its_per_step": newton / args.steps,
                   "bicgstab_its_per_step": kry / args.steps, "host_setup_s": t_setup},
        "roofline": {"bound": "hbm", "kernel": "k_spmv_stream<2,2,1,0> (velocity Jacobian block)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": nbytes, "ms_per_launch": ms_spmv}}))
    ctx.close()


def _init_dist(args):
    """(rank, world, local_rank, torch.distributed | None): gloo bootstrap used only for the
    unique-id broadcast, the barrier and the MAX reduction of the bench contract"""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and args.gpus != world:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    if world == 1 and args.gpus > 1:
        raise SystemExit("--gpus %d needs one process per GPU: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node %d --master-addr 127.0.0.1 bench.py --gpus %d ..." % (
                             args.gpus, args.gpus, args.gpus))
    dist = None
    if world > 1 or os.environ.get("NSFEM_FORCE_COMM") is not None:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, local_rank, dist


def tgv3d_bench(args):
    """BASELINE.json configs[3] on ONE GPU: Taylor-Green vortex on the triple-periodic unit cube
    (Kuhn tetrahedra instead of the hexahedra the config names -- the reference has simplices
    only), Re = 100, IPCS, BDF-2, dt = 0.25 / n, periodic multigrid hierarchy, mean pressure
    shifted to zero after every step (ns_solver_base.py:1190-1203).  The initial state is the
    nodal interpolant of the analytic vortex (convergence_test/taylor_green_vortex.py:111-117)."""
    import dlfn_compat as dlfn
    from fem_mesh import TaylorHoodDofMap, box_mesh, periodic_entity_map
    from multigrid import attach_hierarchy

    class TriplePeriodic(dlfn.SubDomain):
        def inside(self, x, on_boundary):
            return bool(on_boundary and (dlfn.near(x[0], 0.0) or dlfn.near(x[1], 0.0) or dlfn.near(x[2], 0.0)))

        def map(self, x_slave, x_master):
            for a in range(3):
                if dlfn.near(x_slave[a], 1.0):
                    x_master[:] = x_slave
                    x_master[a] -= 1.0
                    return
            x_master[:] = -10.0

    rank, world, local_rank, dist = _init_dist(args)
    n = args.n
    t_setup = time.perf_counter()
    if world == 1:
        mesh = box_mesh((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), n, n, n)
        domain = TriplePeriodic()
        dm = TaylorHoodDofMap(mesh, periodic_map=periodic_entity_map(mesh, domain))
        ctx = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1)
        levels = attach_hierarchy(ctx, mesh, args.mg_degree, args.mg_eig_ratio,
                                  periodic=(domain, dm.p1_vertex_node))
        n_dofs = dm.n_dofs
    else:
        # N > 1, weak scaling: every rank owns n cube layers of the n x n x (n N) box [0,1]^2 x [0,N],
        # periodic in all three directions (the planar vortex is z-invariant, so any z period fits);
        # z wraps around the ranks (PeriodicSlabPartition)
        from partition import PeriodicSlabPartition
        part = PeriodicSlabPartition((0.0, 0.0, 0.0), (1.0, 1.0, float(world)), n, n, n * world, rank, world,
                                     coarsest=args.coarsest if args.coarsest else 16, global_coarsest=4)
        mesh, dm = part.mesh, part.dofmap
        device = 0 if os.environ.get("NSFEM_SHARE_GPU") else local_rank
        ctx = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1, device)
        ids = [nat.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ctx.attach_rccl_comm(ids[0], rank, world)
        levels = part.attach(ctx, args.mg_degree, args.mg_eig_ratio)
        n_dofs = 3 * part.n_p2_global + part.n_p1_global
    _apply_truncation(ctx, args)
    g = 2.0 * np.pi
    X = dm.p2_coords
    u0 = np.stack([np.cos(g * X[:, 0]) * np.sin(g * X[:, 1]), -np.sin(g * X[:, 0]) * np.cos(g * X[:, 1]),
                   np.zeros(dm.n_p2)], axis=1).ravel()
    Y = dm.p1_coords
    p0 = -0.25 * (np.cos(2 * g * Y[:, 0]) + np.cos(2 * g * Y[:, 1]))
    for slot in (nat.U0, nat.U1, nat.U2):
        ctx.set_state(slot, u0)
    for slot in (nat.P, nat.P_OLD):
        ctx.set_state(slot, p0)
    ctx.set_coeffs(1.0, 1.0, 1.0 / 100.0)
    ctx.set_dirichlet(nat.VELOCITY, np.zeros(0, np.int32), np.zeros(0))
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    t_setup = time.perf_counter() - t_setup
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = args.krylov_rtol
    opts.momentum.precond = opts.poisson.precond = 1
    opts.correction.precond = 2 if args.mass_solver == "chebyshev" else 0
    opts.newton_forcing = args.newton_forcing
    dt = args.dt if args.dt != 1.0e-3 else 0.25 / n

    def one_step(i):
        ctx.set_bdf((1.0, -1.0, 0.0) if i == 0 else (1.5, -2.0, 0.5), dt)
        info = ctx.step_ipcs(opts)
        ctx.shift_mean_pressure(0.0)
        ctx.advance(0)
        return info

    for i in range(args.warmup):
        one_step(i)
    ctx.synchronize()
    if dist is not None:
        dist.barrier()
    ctx.comm_stats(reset=True)
    t0 = time.perf_counter()
    newton = kry = poi = 0
    for i in range(args.warmup, args.warmup + args.steps):
        info = one_step(i)
        newton += info.newton_iterations
        kry += info.krylov_iterations_momentum
        poi += info.krylov_iterations_poisson
    ctx.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    comm_per_step = {k: v / args.steps for k, v in ctx.comm_stats().items()}
    # the vortex decays like exp(-2 g^2 t / Re): check the run against the analytic solution
    t_end = dt * (args.warmup + args.steps)
    u = ctx.get_state(nat.U1).reshape(-1, 3)
    err = float(np.abs(u - np.exp(-2.0 * g * g * t_end / 100.0) * u0.reshape(-1, 3)).max())
    if dist is not None:
        import torch
        t = torch.tensor([elapsed, err], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, err = float(t[0]), float(t[1])
    sps = args.steps / elapsed
    ms_spmv, nbytes = ctx.time_spmv(nat.OP_MOMENTUM_SMOOTHER, 100) if world == 1 else (float("nan"), 0)
    achieved = nbytes / (ms_spmv * 1e-3) / 1e9 if world == 1 else None
    if rank != 0:
        ctx.close()
        dist.destroy_process_group()
        return
    print(json.dumps({
        "metric": "dof_updates_per_sec", "value": sps * n_dofs, "unit": "DoF-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 / sps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic", "time_steps_per_sec": sps,
        "config": {"workload": "3D Taylor-Green vortex, triple-periodic unit cube, %d^3 cubes x 6 Kuhn "
                               "tetrahedra per GPU (%d dofs), Re=100, IPCS, BDF-2, dt=%g" % (n, n_dofs, dt),
                   "n_dofs": n_dofs, "newton_tol": 1e-10, "krylov_rtol": args.krylov_rtol,
                   "newton_forcing": args.newton_forcing, "coarse_p1_levels": levels,
                   "parallelism": "1 GPU" if world == 1 else
                   "%d periodic slabs of %d cube layers, RCCL wrap-around halo exchange (%s mode) + all-reduce" % (
                       world, n, args.halo_mode),
                   "newton_its_per_step": newton / args.steps,
                   "bicgstab_its_per_step": kry / args.steps, "poisson_cg_its_per_step": poi / args.steps,
                   "max_abs_velocity_error_vs_analytic": err, "host_setup_s": t_setup,
                   "comm_per_step_rank0": comm_per_step},
        "roofline": {"bound": "hbm", "kernel": "k_spmv_stream<1,1,3,3> (finest-level Chebyshev smoothing step, "
                                               "scalar P2 operator on 3 components; launches interleaved with a cache-flushing SpMV, nsfem_time_spmv)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": None,
                     "algorithmic_bytes_per_launch": nbytes, "ms_per_launch": ms_spmv if world == 1 else None}}))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def cavity3d_bench(args):
    """3D lid-driven cavity on a Kuhn (BoxMesh) tetrahedral mesh, Re = 100, IPCS or monolithic
    BDF-2 -- the relative of BASELINE.json configs[3:5] (3D configurations beyond what the
    reference itself exercises).  N > 1 (IPCS): weak scaling, every rank owns a
    slab of n cube layers of the n x n x (n N) box [0,1]^2 x [0,N] (lid on top), coupled through
    RCCL exactly like the 2D strips."""
    from partition import SlabPartition, global_dof_counts
    rank, world, local_rank, dist = _init_dist(args)
    bdf = args.workload == "cavity3d-bdf"
    n = args.n
    t_setup = time.perf_counter()
    part = SlabPartition((0.0, 0.0, 0.0), (1.0, 1.0, float(world)), n, n, n * world, rank, world,
                         coarsest=args.coarsest if args.coarsest else (_serial_coarsest(n, 3) if world == 1 else 16),
                         global_coarsest=None if world == 1 else 4)
    mesh, dm = part.mesh, part.dofmap
    device = 0 if os.environ.get("NSFEM_SHARE_GPU") else local_rank
    ctx = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1, device)
    if dist is not None:
        ids = [nat.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ctx.attach_rccl_comm(ids[0], rank, world)
    levels = part.attach(ctx, args.mg_degree, args.mg_eig_ratio)
    X = dm.p2_coords
    on = (np.abs(X[:, 2]) < 1e-12) | (np.abs(X[:, 2] - world) < 1e-12)
    for a in range(2):
        on |= (np.abs(X[:, a]) < 1e-12) | (np.abs(X[:, a] - 1.0) < 1e-12)
    nodes = np.nonzero(on)[0]
    lid = np.abs(X[nodes, 2] - world) < 1e-12
    dofs = np.concatenate([3 * nodes, 3 * nodes + 1, 3 * nodes + 2]).astype(np.int32)
    vals = np.concatenate([np.where(lid, 1.0, 0.0), np.zeros(2 * nodes.size)])
    ctx.set_coeffs(1.0, 1.0, 1.0 / 100.0)
    ctx.set_dirichlet(nat.VELOCITY, dofs, vals)
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    ctx.set_dirichlet(nat.PRESSURE_PRECOND, np.zeros(0, np.int32), np.zeros(0))
    t_setup = time.perf_counter() - t_setup
    n2g, n1g = global_dof_counts(n, n, n * world)
    n_dofs = 3 * n2g + n1g
    _apply_truncation(ctx, args)
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = args.krylov_rtol
    opts.momentum.precond = opts.poisson.precond = 1
    opts.correction.precond = 2 if args.mass_solver == "chebyshev" else 0
    opts.newton_forcing = args.newton_forcing
    opts.matrix_free = args.matrix_free
    dt = args.dt if args.dt != 1.0e-3 else 0.5 / n          # CFL ~ 0.5 for the unit lid speed

    def one_step(i):
        ctx.set_bdf((1.0, -1.0, 0.0) if i == 0 else (1.5, -2.0, 0.5), dt)
        info = ctx.step_bdf(opts) if bdf else ctx.step_ipcs(opts)
        ctx.advance(1 if bdf else 0)
        return info

    for i in range(args.warmup):
        one_step(i)
    ctx.synchronize()
    if dist is not None:
        dist.barrier()
    ctx.comm_stats(reset=True)
    t0 = time.perf_counter()
    newton = kry = poi = 0
    for i in range(args.warmup, args.warmup + args.steps):
        info = one_step(i)
        newton += info.newton_iterations
        kry += info.krylov_iterations_momentum
        poi += info.krylov_iterations_poisson
    ctx.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    comm_per_step = {k: v / args.steps for k, v in ctx.comm_stats().items()}
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    sps = args.steps / elapsed
    ms_spmv, nbytes = ctx.time_spmv(nat.OP_MOMENTUM_JAC, 100)
    achieved = nbytes / (ms_spmv * 1e-3) / 1e9
    if rank == 0:
        print(json.dumps({
            "metric": "dof_updates_per_sec", "value": sps * n_dofs, "unit": "DoF-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 / sps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "time_steps_per_sec": sps,
            "config": {"workload": "3D lid-driven cavity Re=100, %dx%dx%d cubes x 6 Kuhn tetrahedra "
                                   "(%d dofs), %s, dt=%g" % (n, n, n * world, n_dofs,
                                                             "BDF-2 monolithic" if bdf else "IPCS", dt),
                       "n_dofs": n_dofs, "newton_tol": 1e-10, "krylov_rtol": args.krylov_rtol,
                       "newton_forcing": args.newton_forcing, "coarse_p1_levels": levels,
                       "parallelism": "1 GPU" if world == 1 else
                       "%d slabs of %d cube layers, RCCL halo exchange (%s mode) + all-reduce" % (world, n, args.halo_mode),
                       "newton_its_per_step": newton / args.steps,
                       "bicgstab_its_per_step": kry / args.steps, "poisson_cg_its_per_step": poi / args.steps,
                       "host_setup_s": t_setup, "comm_per_step_rank0": comm_per_step},
            "roofline": {"bound": "hbm", "kernel": "k_spmv_stream<3,3,1,0> (velocity Jacobian, 3x3 block CSR)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": nbytes, "ms_per_launch": ms_spmv}}))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def _apply_truncation(ctx, args):
    parts = [float(v) for v in str(args.mg_truncation).split(",")]
    ctx.mg_set_truncation(parts[0], parts[1] if len(parts) > 1 else 0.1)
    ctx.mg_set_halo_mode(args.halo_mode == "relaxed")


def _serial_coarsest(n, dim=2):
    """cells across the coarsest mesh on one GPU: the first level with <= 1200 nodes (dense
    solve): 2D 512 -> 32 (1089 nodes), 336 -> 21 (484 nodes); 3D 64 -> 8 (729), 48 -> 6 (343)"""
    while n % 2 == 0 and (n + 1) ** dim > 1200:
        n //= 2
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cells", dest="n", type=int, default=512, help="cells per side and per rank (512 = BASELINE config 2)")
    ap.add_argument("--dt", type=float, default=1.0e-3)
    ap.add_argument("--krylov-rtol", type=float, default=1.0e-8,
                    help="relative residual of the linear solves (Poisson, mass; Newton solves when exact)")
    ap.add_argument("--newton-forcing", type=float, default=1.0e-4,
                    help="inexact Newton: reduce each Newton linear residual only by this factor "
                         "(0 = exact Newton with --krylov-rtol)")
    ap.add_argument("--cpu-sample-n", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-multigrid", action="store_true")
    ap.add_argument("--mg-degree", type=int, default=None,
                    help="Chebyshev smoother degree (default: 2 on structured meshes, 3 on the DFG mesh)")
    ap.add_argument("--mg-eig-ratio", type=float, default=None,
                    help="smoothing interval [lmax / ratio, lmax] (default: 4 structured, 16 DFG)")
    ap.add_argument("--matrix-free", type=int, default=0, choices=(0, 1, 2),
                    help="velocity Jacobian in the step driver: 0 auto, 1 assembled, 2 matrix-free")
    ap.add_argument("--mass-solver", choices=("chebyshev", "cg"), default="chebyshev",
                    help="velocity-correction mass solve: Chebyshev with a-priori bounds (no dots) or Jacobi-CG")
    ap.add_argument("--halo-mode", choices=("relaxed", "exact"), default="relaxed",
                    help="N > 1: multigrid smoothing with one halo exchange per smoothing sequence "
                         "(frozen ghosts in between) or per SpMV (the serial algorithm)")
    ap.add_argument("--mg-truncation", default="4,0.1",
                    help="R[,TOL]: truncate the velocity multigrid cycle at the first level with "
                         "c_v K_ii <= R alpha0/k M_ii, solved there by Chebyshev iteration to TOL (0: off)")
    ap.add_argument("--coarsest", type=int, default=0, help="cells across the coarsest multigrid mesh (0: default)")
    ap.add_argument("--workload", choices=("cavity-ipcs", "dfg-bdf", "cavity3d-ipcs", "cavity3d-bdf", "tgv3d-ipcs"),
                    default="cavity-ipcs",
                    help="cavity-ipcs = BASELINE configs[1] (headline); dfg-bdf = configs[2], 1 GPU; "
                         "cavity3d-* = 3D tetrahedral cavity (--cells cubes per side); tgv3d-ipcs = configs[3] "
                         "(triple-periodic Taylor-Green vortex) on 1 GPU")
    ap.add_argument("--dfg-refine", type=int, default=5)
    args = ap.parse_args()
    if args.workload == "dfg-bdf":
        if int(os.environ.get("WORLD_SIZE", "1")) != 1:
            raise SystemExit("the dfg-bdf workload is a single-GPU configuration")
        return dfg_bdf_bench(args)
    if args.workload.startswith("cavity3d"):
        if args.n == 512:
            args.n = 32
        return cavity3d_bench(args)
    if args.workload == "tgv3d-ipcs":
        if args.n == 512:
            args.n = 32
        return tgv3d_bench(args)

    rank, world, local_rank, dist = _init_dist(args)

    # ---- local problem: strip `rank` of the 512 x (512 * world) mesh (own rows + ghost row)
    from partition import StripPartition, global_dof_counts
    n = args.n
    # one GPU: coarsest mesh 32 cells across (1089-node dense solve, inverse computed on the device)
    # partitioned: the distributed levels stop at 64 cells across; below that every rank runs the
    # replicated global hierarchy (64 -> 8 cells across, dense solve at the bottom) without any
    # halo exchange
    part = StripPartition((0.0, 0.0), (1.0, float(world)), n, n * world, rank, world,
                          coarsest=args.coarsest if args.coarsest else (_serial_coarsest(n) if world == 1 else 64),
                          global_coarsest=None if world == 1 else 8)
    dm = part.dofmap
    device = local_rank
    if os.environ.get("NSFEM_SHARE_GPU"):        # rehearsal of several ranks on a one-GPU box
        device = 0
    ctx = nat.NsfemContext(part.mesh.coords, part.mesh.cells, dm.p2_dofmap, dm.p1_dofmap,
                           dm.n_p2, dm.n_p1, device)
    if dist is not None:
        ids = [nat.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        ctx.attach_rccl_comm(ids[0], rank, world)
    mg_levels = None
    if not args.no_multigrid:
        mg_levels = part.attach(ctx, args.mg_degree, args.mg_eig_ratio)
    else:
        n2g, n1g = global_dof_counts(n, n * world)
        ctx.set_partition(rank, world, part.p2_ghost, part.p1_ghost, part.p2_halo, part.p1_halo,
                          n2g, n1g)
    ctx.set_coeffs(1.0, 1.0, 1.0 / 100.0)
    ctx.set_dirichlet(nat.VELOCITY, *cavity_dirichlet(dm, float(world)))
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    n2g, n1g = global_dof_counts(n, n * world)
    n_dofs = 2 * n2g + n1g

    _apply_truncation(ctx, args)
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = args.krylov_rtol
    if mg_levels is not None:
        opts.momentum.precond = opts.poisson.precond = 1
    opts.correction.precond = 2 if args.mass_solver == "chebyshev" else 0
    opts.newton_forcing = args.newton_forcing
    opts.matrix_free = args.matrix_free

    def one_step(i):
        ctx.set_bdf((1.0, -1.0, 0.0) if i == 0 else (1.5, -2.0, 0.5), args.dt)
        info = ctx.step_ipcs(opts)
        ctx.advance(0)
        return info

    for i in range(args.warmup):
        one_step(i)
    ctx.synchronize()
    if dist is not None:
        dist.barrier()
    ctx.comm_stats(reset=True)
    t0 = time.perf_counter()
    newton = kry = poi = 0
    for i in range(args.warmup, args.warmup + args.steps):
        info = one_step(i)
        newton += info.newton_iterations
        kry += info.krylov_iterations_momentum
        poi += info.krylov_iterations_poisson
    ctx.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    comm_per_step = {k: v / args.steps for k, v in ctx.comm_stats().items()}
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    steps_per_s = args.steps / elapsed
    # the same steps with direct-solver accuracy in every linear solve (rtol 1e-12 of SURVEY 8d's
    # parity runs, exact Newton): reported beside the headline, never as `value`
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = 1.0e-12
    opts.newton_forcing = 0.0
    n_par = max(2, min(5, args.steps))
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup + args.steps, args.warmup + args.steps + n_par):
        one_step(i)
    ctx.synchronize()
    ms_parity = 1e3 * (time.perf_counter() - t0) / n_par
    # dominant kernel of the timed steps: the Chebyshev-Jacobi smoothing step of the velocity
    # multigrid on its finest level (scalar P2 operator applied to both components, fused
    # epilogue).  Its launches are timed IN SITU on the context's stream -- one HIP-event pair around
    # every run of consecutive launches inside a smoothing sequence (a pair per launch would add
    # ~4 us of event overhead to each 40 us kernel) -- over 5 further steps with the throughput settings (the event pairs stay out of the
    # timed region above).  Back-to-back repetitions of the same launch would be flattered by the
    # 256 MB Infinity Cache holding the 145 MB operator.
    if mg_levels is not None:
        for o in (opts.momentum, opts.poisson, opts.correction):
            o.rtol = args.krylov_rtol
        opts.newton_forcing = args.newton_forcing
        ctx.profile_smoother(True)
        for i in range(args.warmup + args.steps + n_par, args.warmup + args.steps + n_par + 5):
            one_step(i)
        ms_spmv, n_launches, nbytes = ctx.profile_smoother(False)
    else:
        ms_spmv, nbytes = ctx.time_spmv(nat.OP_MOMENTUM_JAC, 200)
        n_launches = 200
    achieved = nbytes / (ms_spmv * 1e-3) / 1e9
    ms_jac, nbytes_jac = ctx.time_spmv(nat.OP_MOMENTUM_JAC, 200)
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_h_pmc_fetch_write_size.json")
    if world == 1 and n == 512 and mg_levels is not None and os.path.exists(pmc):
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this workload:
        # 2 x FETCH_SIZE (gfx950 wide-read correction, MI355X_MICROARCH.md section HBM) + WRITE_SIZE;
        # the symbol is launched on every multigrid level, the finest-level launches are the maxima
        c = json.load(open(pmc))
        key = "void nsfem::k_spmv_stream<1, 1, 2, 3>"
        if key in c["fetch"] and key in c["write"]:
            traffic = (2.0 * c["fetch"][key]["max_KB"] + c["write"][key]["max_KB"]) * 1024.0
    out = {
        "metric": "dof_updates_per_sec", "value": steps_per_s * n_dofs, "unit": "DoF-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "time_steps_per_sec": steps_per_s,
        "config": {"workload": "2D lid-driven cavity Re=100, %dx%d right-diagonal Taylor-Hood P2/P1 "
                               "(%d dofs), IPCS, BDF-2, dt=%g, zero initial state" % (
                                   n, n * world, n_dofs, args.dt),
                   "n_dofs": n_dofs, "newton_tol": 1e-10, "krylov_rtol": args.krylov_rtol,
                   "newton_forcing": args.newton_forcing,
                   "velocity_jacobian": {0: "matrix-free (default)", 1: "assembled block CSR",
                                         2: "matrix-free"}[args.matrix_free],
                   "ms_per_step_with_krylov_rtol_1e-12_exact_newton": ms_parity,
                   "preconditioner": "jacobi" if mg_levels is None else
                   "geometric multigrid, Chebyshev-Jacobi smoothing: V(0,3) momentum, V(2,2) Poisson; %d coarse P1 levels" % mg_levels,
                   "parallelism": "1 GPU" if world == 1 else
                   "%d strips of %d cell rows, RCCL halo exchange (%s mode) + all-reduce" % (world, n, args.halo_mode),
                   "newton_its_per_step": newton / args.steps,
                   "bicgstab_its_per_step": kry / args.steps, "poisson_cg_its_per_step": poi / args.steps,
                   "comm_per_step_rank0": comm_per_step},
        "roofline": {"bound": "hbm",
                     "kernel": "k_spmv_stream<1,1,2,3>, finest multigrid level: Chebyshev-Jacobi smoothing step "
                               "y = x + c1 d + c2 dinv (b - L x) on the scalar P2 operator L (12.07 M nnz), "
                               "both velocity components" if mg_levels is not None else
                               "k_spmv_stream<2,2,1,0> (momentum Jacobian)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": nbytes, "ms_per_launch": ms_spmv,
                     "launches_timed": n_launches,
                     "timing": "HIP-event pairs around the runs of consecutive finest-level launches (one pair per smoothing sequence) during 5 solver steps"},
        "jacobian_spmv": {"kernel": "k_spmv_stream<2,2,1,0> (assembled momentum Jacobian, 2x2 block CSR; "
                                    "used by the explicit assembly seam / matrix_free=1)",
                          "achieved": nbytes_jac / (ms_jac * 1e-3) / 1e9, "unit": "GB/s",
                          "frac": nbytes_jac / (ms_jac * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "algorithmic_bytes_per_launch": nbytes_jac, "ms_per_launch": ms_jac},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample_n, args.dt, 6)
    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
