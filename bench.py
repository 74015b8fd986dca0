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


def _ranks_to_spawn(argv):
    """N when this process was started as plain `python bench.py --gpus N` (N > 1, no launcher
    environment, no --local-ranks): it then only spawns and supervises the N rank processes"""
    if "WORLD_SIZE" in os.environ:
        return 0
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
        elif a == "--local-ranks" or a.startswith("--local-ranks="):
            return 0
    return n if n > 1 else 0


def _spawn_ranks(argv, n):
    """`python bench.py --gpus N` as typed: this parent NEVER touches the GPU (no HIP library is
    loaded, torch is not imported); it starts N fresh child processes -- one rank per GPU, the
    environment torch.distributed.run would give them (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR,
    MASTER_PORT) -- relays rank 0's JSON line and exits non-zero when any child does (the other
    ranks are then terminated: they would wait in a barrier for ever)."""
    import socket
    import subprocess
    # (the rendezvous port is picked by bind(0) / close: another process may take it before rank 0 listens -- a job
    # that dies within its first seconds is started once more on a fresh port)
    attempt = int(os.environ.get("NSFEM_SPAWN_ATTEMPT", "0"))
    t_start = time.time()
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    failed = None
    try:
        import threading
        chunks = []
        reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        while failed is None and any(p.poll() is None for p in procs):
            for r, p in enumerate(procs):
                if p.poll() not in (None, 0):
                    failed = (r, p.returncode)
            time.sleep(0.1)
        for r, p in enumerate(procs):
            if failed is None and p.poll() not in (None, 0):
                failed = (r, p.returncode)
        if failed is None:
            reader.join(timeout=30)
            sys.stdout.write(b"".join(chunks).decode())
            sys.stdout.flush()
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except Exception:
                p.kill()
    if failed is not None:
        sys.stderr.write("bench.py: rank %d exited with code %d\n" % failed)
        if attempt == 0 and time.time() - t_start < 20.0 and not b"".join(chunks).strip():
            sys.stderr.write("bench.py: the job died during start-up; one more attempt on a fresh rendezvous port\n")
            os.environ["NSFEM_SPAWN_ATTEMPT"] = "1"
            return _spawn_ranks(argv, n)
        return failed[1] if failed[1] > 0 else 1
    return 0


if __name__ == "__main__" and _ranks_to_spawn(sys.argv[1:]):
    sys.exit(_spawn_ranks(sys.argv[1:], _ranks_to_spawn(sys.argv[1:])))

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


def cpu_baseline(sizes, k, workload_dofs):
    """Reference algorithm on the host CPU (1 process, as the reference runs): per Newton
    iteration full re-assembly + SuperLU; Poisson and mass matrices re-assembled and
    re-factorised each step (dolfin LinearVariationalSolver behaviour).  Timed on a bounded
    ladder of cavity sizes `sizes` = [(n, steps), ...]; the cost exponent (time per step ~
    dofs^e) is fitted over the ladder and the rate at the workload size extrapolated with it
    (the LU of the workload itself is out of reach: SURVEY.md section 8d)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fem_oracle as fo
    from fem_mesh import TaylorHoodDofMap
    from grid_generator import hyper_cube
    samples = []
    for n_sample, steps in sizes:
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
        samples.append({"cells_per_side": n_sample, "dofs": dm.n_dofs, "steps": steps,
                        "steps_per_s": steps / dt, "s_per_step": dt / steps})
    big = samples[-1]
    out = {"value": big["steps_per_s"] * big["dofs"], "unit": "DoF-updates/s", "cores": 1, "kind": "port",
           "sample": "oracle IPCS (re-assembly + SuperLU per Newton iteration, Poisson/mass "
                     "re-factorised per step) on the n=%d cavity (%d dofs), %d steps after 1 warm-up: "
                     "%.3f steps/s x %d dofs; smaller sizes in `samples`" % (
                         big["cells_per_side"], big["dofs"], big["steps"], big["steps_per_s"], big["dofs"]),
           "sample_steps_per_s": big["steps_per_s"], "sample_dofs": big["dofs"],
           "samples": samples, "host_cpus": os.cpu_count()}
    if len(samples) >= 2:
        x = np.log([v["dofs"] for v in samples])
        y = np.log([v["s_per_step"] for v in samples])
        e, c = np.polyfit(x, y, 1)
        t_work = float(np.exp(c + e * np.log(workload_dofs)))
        out["cost_exponent"] = float(e)
        out["extrapolated_steps_per_s_at_workload"] = 1.0 / t_work
        out["extrapolated_value_at_workload"] = workload_dofs / t_work
        out["extrapolation"] = "s/step ~ dofs^%.2f fitted over `samples`, evaluated at %d dofs " \
                               "(not measured: sparse LU fill at that size exceeds the host)" % (e, workload_dofs)
    return out


def dfg_bdf_bench(args):
    """BASELINE.json configs[2]: DFG 2D-2 cylinder channel (reference demo/dfg_benchmark.py),
    Re = 100, BDF-2 monolithic scheme, dt = 0.005, curved-boundary refinement hierarchy of the
    in-repo block mesh (m = 4, 5 refinements: 589,824 cells, ~2.65 M dofs).
    N > 1 (strong scaling -- the mesh is what it is): recursive coordinate bisection of the coarsest
    cells (partition.GraphPartition), index-list halos, additive parts of the algebraic Schur
    Laplacian; drag / lift are integrated by every rank over the cylinder facets of its own cells
    and summed."""
    import grid_generator as gg
    from fem_mesh import TaylorHoodDofMap
    from multigrid import attach_hierarchy, attach_schur_laplacian
    rank, world, local_rank, dist = _init_dist(args)
    t_setup = time.perf_counter()
    mesh, marks = gg.dfg_channel(4, args.dfg_refine)
    n_cells_global = mesh.num_cells()
    device = 0 if os.environ.get("NSFEM_SHARE_GPU") else local_rank
    part = None
    if world == 1:
        dm = TaylorHoodDofMap(mesh)
        ctx = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1, device)
        if dist is not None:
            _attach_comm(ctx, dist, rank, world)
        levels = attach_hierarchy(ctx, mesh, args.mg_degree, args.mg_eig_ratio)
        n_dofs = dm.n_dofs
        own_cell = np.ones(mesh.num_cells(), dtype=bool)
    else:
        from partition import GraphPartition
        part = GraphPartition(mesh, rank, world, marks)
        own_cell = part.cell_owner[0][part.fine.cells] == rank
        mesh, marks, dm = part.mesh, part.markers, part.dofmap
        ctx = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1, device)
        _attach_comm(ctx, dist, rank, world)
        levels = part.attach(ctx, args.mg_degree, args.mg_eig_ratio)
        n_dofs = 2 * part.n_p2_global + part.n_p1_global
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
    attach_schur_laplacian(ctx, bd, part=part)
    t_setup = time.perf_counter() - t_setup
    _apply_truncation(ctx, args)
    ctx.set_overlap(args.overlap == "on")
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
    if dist is not None:
        dist.barrier()
    ctx.comm_stats(reset=True)
    t0 = time.perf_counter()
    newton = kry = 0
    for i in range(args.warmup, args.warmup + args.steps):
        info = one_step(i)
        newton += info.newton_iterations
        kry += info.krylov_iterations_momentum
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
    # the physical output of the configuration (demo/dfg_benchmark.py:44-66): drag / lift coefficients
    # from the surface traction on the cylinder, c = 2 F / (U_mean^2 D) with U_mean = D = 1.  The
    # reference integrates  -p n + 1/Re sym(grad u) n  (its d lacks the factor 2 of the Newtonian
    # stress); both variants are reported.  (The benchmark's published maxima, c_D ~ 3.23, c_L ~ 1.0,
    # belong to the periodic vortex-shedding state at t > 30, far beyond these few steps.)
    # Every boundary facet is integrated by the rank that owns its cell; the parts are summed.
    def mine(facets):
        fc, fl = mesh.facet_cell_local(facets)
        keep = own_cell[fc]
        return fc[keep], fl[keep]

    fc, fl = mine(marks.facets_with_id(ids.cylinder.value))
    zero = (np.zeros(2), 0.0, 0.0)
    f_ref, _, perimeter = ctx.boundary_force(fc, fl, 0.5 / 100.0, 1.0, nat.U0, nat.P) if fc.size else zero
    f_std, _, _ = ctx.boundary_force(fc, fl, 1.0 / 100.0, 1.0, nat.U0, nat.P) if fc.size else zero
    fc_all, fl_all = mine(np.nonzero(mesh.facet_on_boundary)[0])
    _, net_flux, _ = ctx.boundary_force(fc_all, fl_all, 0.0, 0.0, nat.U0, nat.P) if fc_all.size else zero
    f_ref0, f_ref1, perimeter, f_std0, f_std1, net_flux = ctx.comm_allreduce(
        [f_ref[0], f_ref[1], perimeter, f_std[0], f_std[1], net_flux], "sum")
    ms_spmv, nbytes = ctx.time_spmv(nat.OP_MOMENTUM_JAC, 200)
    achieved = nbytes / (ms_spmv * 1e-3) / 1e9
    if rank == 0:
        print(json.dumps({
            "metric": "dof_updates_per_sec", "value": sps * n_dofs, "unit": "DoF-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 / sps,
            "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "time_steps_per_sec": sps,
            "config": {"workload": "DFG 2D-2 cylinder channel Re=100, %d unstructured triangles (%d dofs), "
                                   "BDF-2 monolithic, dt=%g, impulsive start" % (n_cells_global, n_dofs, dt),
                       "n_dofs": n_dofs, "newton_tol": 1e-10, "krylov_rtol": args.krylov_rtol,
                       "newton_forcing": args.newton_forcing,
                       "preconditioner": "block-triangular (V-cycle velocity block, Cahouet-Chabard Schur "
                                         "with algebraic pressure Laplacian), %d coarse P1 levels" % levels,
                       "parallelism": "1 GPU" if world == 1 else
                       "%d parts by recursive coordinate bisection of the coarsest cells, index-list halos "
                       "(overlap %s) + all-reduce" % (world, args.overlap),
                       "newton_its_per_step": newton / args.steps,
                       "bicgstab_its_per_step": kry / args.steps, "host_setup_s": t_setup,
                       "comm_per_step_rank0": comm_per_step,
                       "t_end": dt * (args.warmup + args.steps),
                       "drag_lift_reference_formula": [-2.0 * f_ref0, -2.0 * f_ref1],
                       "drag_lift_newtonian_stress": [-2.0 * f_std0, -2.0 * f_std1],
                       "cylinder_perimeter_of_the_mesh": perimeter, "net_boundary_mass_flux": net_flux},
            "roofline": {"bound": "hbm", "kernel": "k_spmv_stream<2,2,1,0> (velocity Jacobian block)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": nbytes, "ms_per_launch": ms_spmv}}))
    ctx.close()
    _finish_dist(dist)


class _ThreadRanks:
    """torch.distributed look-alike for ranks that are THREADS of this process sharing one GPU
    (--local-ranks N): a functional rehearsal of the N-rank code paths of every workload on a
    one-GPU box -- partitions, halo exchanges, all-reduces, the bench's own reductions -- through
    the in-process communicator (nsfem_comm_attach_local).  Not a performance mode: the ranks
    serialise on one device and every collective synchronises the host."""

    class ReduceOp:
        SUM, MAX = "sum", "max"

    def __init__(self, world):
        import threading
        self.world = world
        self.group = nat.local_group_create(world)
        self._barrier = threading.Barrier(world)
        self._slots = [None] * world
        self._tls = threading.local()

    def bind(self, rank):
        self._tls.rank = rank

    def barrier(self):
        self._barrier.wait()

    def all_reduce(self, t, op="sum"):
        self._slots[self._tls.rank] = t.clone()
        self._barrier.wait()
        acc = self._slots[0].clone()
        for other in self._slots[1:]:
            acc = acc + other if op == "sum" else acc.maximum(other)
        self._barrier.wait()
        t.copy_(acc)

    def broadcast_object_list(self, objs, src=0):
        self._slots[self._tls.rank] = list(objs)
        self._barrier.wait()
        objs[:] = self._slots[src]
        self._barrier.wait()

    def destroy_process_group(self):
        pass


def _attach_comm(ctx, dist, rank, world):
    """one RCCL rank per process -- or, for thread ranks, the in-process communicator"""
    if isinstance(dist, _ThreadRanks):
        ctx.attach_local_comm(dist.group, rank)
        return
    if os.environ.get("NSFEM_SHARE_GPU"):
        # several rank PROCESSES on one device (one-GPU box): RCCL refuses that, the host-staged
        # shared-memory communicator carries the same collectives
        _attach_comm.counter = getattr(_attach_comm, "counter", 0) + 1
        names = ["/nsfem_%d_%d" % (os.getpid(), _attach_comm.counter) if rank == 0 else None]
        dist.broadcast_object_list(names, src=0)
        ctx.attach_shm_comm(names[0], rank, world)
        return
    ids = [nat.rccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    ctx.attach_rccl_comm(ids[0], rank, world)


def _init_dist(args):
    """(rank, world, local_rank, torch.distributed | None): gloo bootstrap used only for the
    unique-id broadcast, the barrier and the MAX reduction of the bench contract"""
    threads = getattr(args, "thread_ranks", None)
    if threads is not None:
        return args.thread_rank, threads.world, 0, threads
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and args.gpus != world:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    if world == 1 and args.gpus > 1:      # (plain `bench.py --gpus N` never gets here: _spawn_ranks)
        raise SystemExit("--gpus %d: WORLD_SIZE=1 in the environment contradicts it" % args.gpus)
    dist = None
    if world > 1 or os.environ.get("NSFEM_FORCE_COMM") is not None:
        import torch.distributed as dist
        if not dist.is_initialized():     # (one process group for all the workloads of a job)
            if "MASTER_ADDR" not in os.environ:
                os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
            dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, local_rank, dist


def _finish_dist(dist):
    """the torch process group is shared by the workloads of one job and torn down once, in main()"""
    if dist is not None and isinstance(dist, _ThreadRanks):
        dist.destroy_process_group()


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
        from fem_mesh import preferred_p2_order
        dm = TaylorHoodDofMap(mesh, reorder=preferred_p2_order(3), periodic_map=periodic_entity_map(mesh, domain))
        ctx = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1)
        levels = attach_hierarchy(ctx, mesh, args.mg_degree, args.mg_eig_ratio,
                                  periodic=(domain, dm.p1_vertex_node))
        n_dofs = dm.n_dofs
    else:
        # N > 1, weak scaling: every rank owns n cube layers of the n x n x (n N) box [0,1]^2 x [0,N],
        # periodic in all three directions (the planar vortex is z-invariant, so any z period fits);
        # z wraps around the ranks (PeriodicSlabPartition)
        # (--scaling strong: ONE n^3 box cut into `world` periodic slabs of n / world cube layers)
        from partition import PeriodicSlabPartition
        strong = args.scaling == "strong"
        nz_global, zlen = (n, 1.0) if strong else (n * world, float(world))
        part = PeriodicSlabPartition((0.0, 0.0, 0.0), (1.0, 1.0, zlen), n, n, nz_global, rank, world,
                                     coarsest=args.coarsest if args.coarsest else 16, global_coarsest=4)
        mesh, dm = part.mesh, part.dofmap
        device = 0 if os.environ.get("NSFEM_SHARE_GPU") else local_rank
        ctx = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1, device)
        _attach_comm(ctx, dist, rank, world)
        levels = part.attach(ctx, args.mg_degree, args.mg_eig_ratio)
        n_dofs = 3 * part.n_p2_global + part.n_p1_global
    _apply_truncation(ctx, args)
    ctx.set_overlap(args.overlap == "on")
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
    opts.pressure_extrapolation = 1 if args.pressure_start == "extrapolated" else 0
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
    # dominant kernel: finest-level smoothing launch of the velocity multigrid, in situ (HIP-event pairs
    # around the smoothing sequences of 2 further steps) and cache-cold (flush-interleaved)
    ms_spmv = ms_cold = None
    nbytes, n_sm, label, extra = 0, 0, "", {}
    if world == 1:
        ctx.profile_smoother(True)
        for i in range(args.warmup + args.steps, args.warmup + args.steps + 2):
            one_step(i)
        ms_spmv, n_sm, nbytes = ctx.profile_smoother(False)
        ms_cold, _ = ctx.time_spmv(nat.OP_MOMENTUM_SMOOTHER, 100)
        label, extra = _smoother_roofline_extras(ctx, 3, ms_spmv, ms_cold)
    achieved = nbytes / (ms_spmv * 1e-3) / 1e9 if world == 1 else None
    if rank != 0:
        ctx.close()
        _finish_dist(dist)
        return
    print(json.dumps({
        "metric": "dof_updates_per_sec", "value": sps * n_dofs, "unit": "DoF-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 / sps,
        "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic", "time_steps_per_sec": sps,
        "config": {"workload": "3D Taylor-Green vortex, triple-periodic unit cube, %d^3 cubes x 6 Kuhn "
                               "tetrahedra %s (%d dofs), Re=100, IPCS, BDF-2, dt=%g" % (
                                   n, "in total" if (world > 1 and args.scaling == "strong") else "per GPU", n_dofs, dt),
                   "n_dofs": n_dofs, "newton_tol": 1e-10, "krylov_rtol": args.krylov_rtol,
                   "newton_forcing": args.newton_forcing, "coarse_p1_levels": levels,
                   "parallelism": "1 GPU" if world == 1 else
                   "%d periodic slabs of %d cube layers, RCCL wrap-around halo exchange (%s mode, overlap %s) + all-reduce" % (
                       world, n // world if args.scaling == "strong" else n, args.halo_mode, args.overlap),
                   "newton_its_per_step": newton / args.steps,
                   "bicgstab_its_per_step": kry / args.steps, "poisson_cg_its_per_step": poi / args.steps,
                   "max_abs_velocity_error_vs_analytic": err, "host_setup_s": t_setup,
                   "comm_per_step_rank0": comm_per_step},
        "roofline": dict({"bound": "hbm", "kernel": label,
                          "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": None,
                          "algorithmic_bytes_per_launch": nbytes, "ms_per_launch": ms_spmv, "launches_timed": n_sm,
                          "timing": "in situ, HIP-event pairs around the smoothing sequences of 2 steps",
                          "cold_cache": {"achieved": nbytes / (ms_cold * 1e-3) / 1e9,
                                         "frac": nbytes / (ms_cold * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                         "ms_per_launch": ms_cold} if ms_cold else None}, **extra)}))
    ctx.close()
    _finish_dist(dist)


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
    strong = args.scaling == "strong"
    nz_global, zlen = (n, 1.0) if strong else (n * world, float(world))
    part = SlabPartition((0.0, 0.0, 0.0), (1.0, 1.0, zlen), n, n, nz_global, rank, world,
                         coarsest=args.coarsest if args.coarsest else (_serial_coarsest(n, 3) if world == 1 else 16),
                         global_coarsest=None if world == 1 else 4)
    mesh, dm = part.mesh, part.dofmap
    device = 0 if os.environ.get("NSFEM_SHARE_GPU") else local_rank
    ctx = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1, device)
    if dist is not None:
        _attach_comm(ctx, dist, rank, world)
    levels = part.attach(ctx, args.mg_degree, args.mg_eig_ratio)
    X = dm.p2_coords
    on = (np.abs(X[:, 2]) < 1e-12) | (np.abs(X[:, 2] - zlen) < 1e-12)
    for a in range(2):
        on |= (np.abs(X[:, a]) < 1e-12) | (np.abs(X[:, a] - 1.0) < 1e-12)
    nodes = np.nonzero(on)[0]
    lid = np.abs(X[nodes, 2] - zlen) < 1e-12
    dofs = np.concatenate([3 * nodes, 3 * nodes + 1, 3 * nodes + 2]).astype(np.int32)
    vals = np.concatenate([np.where(lid, 1.0, 0.0), np.zeros(2 * nodes.size)])
    ctx.set_coeffs(1.0, 1.0, 1.0 / 100.0)
    ctx.set_dirichlet(nat.VELOCITY, dofs, vals)
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    ctx.set_dirichlet(nat.PRESSURE_PRECOND, np.zeros(0, np.int32), np.zeros(0))
    t_setup = time.perf_counter() - t_setup
    n2g, n1g = global_dof_counts(n, n, nz_global)
    n_dofs = 3 * n2g + n1g
    _apply_truncation(ctx, args)
    ctx.set_overlap(args.overlap == "on")
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = args.krylov_rtol
    opts.momentum.precond = opts.poisson.precond = 1
    opts.correction.precond = 2 if args.mass_solver == "chebyshev" else 0
    opts.newton_forcing = args.newton_forcing
    opts.pressure_extrapolation = 1 if args.pressure_start == "extrapolated" else 0
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
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "time_steps_per_sec": sps,
            "config": {"workload": "3D lid-driven cavity Re=100, %dx%dx%d cubes x 6 Kuhn tetrahedra "
                                   "(%d dofs), %s, dt=%g" % (n, n, nz_global, n_dofs,
                                                             "BDF-2 monolithic" if bdf else "IPCS", dt),
                       "n_dofs": n_dofs, "newton_tol": 1e-10, "krylov_rtol": args.krylov_rtol,
                       "newton_forcing": args.newton_forcing, "coarse_p1_levels": levels,
                       "parallelism": "1 GPU" if world == 1 else
                       "%d slabs of %d cube layers, RCCL halo exchange (%s mode, overlap %s) + all-reduce" % (
                           world, nz_global // world, args.halo_mode, args.overlap),
                       "newton_its_per_step": newton / args.steps,
                       "bicgstab_its_per_step": kry / args.steps, "poisson_cg_its_per_step": poi / args.steps,
                       "host_setup_s": t_setup, "comm_per_step_rank0": comm_per_step},
            "roofline": {"bound": "hbm", "kernel": "k_spmv_stream<3,3,1,0> (velocity Jacobian, 3x3 block CSR)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": nbytes, "ms_per_launch": ms_spmv}}))
    ctx.close()
    _finish_dist(dist)


def channel3d_bdf_bench(args):
    """BASELINE.json configs[4]: 3D channel, 2 : 1 : 1 box [0,2] x [0,1]^2 cut into (2n, n, n)
    cubes x 6 Kuhn tetrahedra (the reference has simplices only), inlet
    u_x = 16 y (1 - y) z (1 - z) at x = 0, no-slip side walls, natural outflow at x = 2,
    Re = 1000 (--reynolds), fully implicit BDF-2 on the mixed P2^3 x P1 system
    (source/ns_bdf_solver.py:36-106): Newton with the reference's criterion, BiCGStab on the mixed
    operator (matrix-free velocity Jacobian), block-triangular preconditioner with the algebraic
    Schur Laplacian of the open outlet.  Impulsive start from rest.  Checked invariants: Newton
    converged in every step; discrete mass balance (flux in + out + walls = 0, the continuity
    rows tested with the constant P1 function); inflow flux = integral of the interpolated inlet
    profile (-4/9 + O(h^4)).

    N > 1: slabs of cube layers along z over RCCL.  --scaling strong: the same box, n / N layers
    per rank; weak (default): a duct [0,2] x [0,1] x [0,N] of (2n, n, n N) cubes, n layers per
    rank, inlet profile 16 y (1 - y) (z / N)(1 - z / N).  Every rank holds only its additive part
    of the algebraic Schur Laplacian (multigrid.attach_schur_laplacian(part=...))."""
    from fem_mesh import Mesh
    from multigrid import attach_schur_laplacian
    from partition import SlabPartition, global_dof_counts
    rank, world, local_rank, dist = _init_dist(args)
    n = args.n
    strong = args.scaling == "strong" or world == 1
    nz, zlen = (n, 1.0) if strong else (n * world, float(world))
    t_setup = time.perf_counter()
    part = SlabPartition((0.0, 0.0, 0.0), (2.0, 1.0, zlen), 2 * n, n, nz, rank, world,
                         coarsest=args.coarsest if args.coarsest else 4)
    mesh, dm = part.mesh, part.dofmap
    device = 0 if os.environ.get("NSFEM_SHARE_GPU") else local_rank
    ctx = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1, device)
    if dist is not None:
        _attach_comm(ctx, dist, rank, world)
    levels = part.attach(ctx, args.mg_degree, args.mg_eig_ratio)
    X = dm.p2_coords
    near = lambda v, c: np.abs(v - c) < 1e-12
    inlet = np.nonzero(near(X[:, 0], 0.0))[0]
    walls = np.nonzero(near(X[:, 1], 0.0) | near(X[:, 1], 1.0) | near(X[:, 2], 0.0) | near(X[:, 2], zlen))[0]
    profile = lambda P: 16.0 * P[..., 1] * (1.0 - P[..., 1]) * (P[..., 2] / zlen) * (1.0 - P[..., 2] / zlen)
    # list order of the reference's DirichletBC.apply: the inlet first, the walls win on shared edges
    bd = np.concatenate([3 * inlet, 3 * inlet + 1, 3 * inlet + 2,
                         3 * walls, 3 * walls + 1, 3 * walls + 2]).astype(np.int32)
    bv = np.concatenate([profile(X[inlet]), np.zeros(2 * inlet.size + 3 * walls.size)])
    ctx.set_coeffs(1.0, 1.0, 1.0 / args.reynolds)
    ctx.set_dirichlet(nat.VELOCITY, bd, bv)
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    ctx.set_dirichlet(nat.PRESSURE_PRECOND, np.zeros(0, np.int32), np.zeros(0))
    t_schur = time.perf_counter()
    attach_schur_laplacian(ctx, np.unique(bd), part=part)
    t_schur = time.perf_counter() - t_schur
    t_setup = time.perf_counter() - t_setup
    n2g, n1g = global_dof_counts(2 * n, n, nz)
    n_dofs = 3 * n2g + n1g
    _apply_truncation(ctx, args)
    ctx.set_overlap(args.overlap == "on")
    opts = ctx.default_step_opts()
    opts.momentum.rtol, opts.momentum.precond, opts.momentum.max_iter = args.krylov_rtol, 1, 500
    opts.newton_forcing = args.newton_forcing
    opts.matrix_free = args.matrix_free
    dt = args.dt if args.dt != 1.0e-3 else 0.5 / n          # CFL ~ 1 for the unit peak inflow speed

    def one_step(i):
        ctx.set_bdf((1.0, -1.0, 0.0) if i == 0 else (1.5, -2.0, 0.5), dt)
        info = ctx.step_bdf(opts)
        ctx.advance(1)
        return info

    for i in range(args.warmup):
        one_step(i)
    ctx.synchronize()
    if dist is not None:
        dist.barrier()
    ctx.comm_stats(reset=True)
    t0 = time.perf_counter()
    newton = kry = 0
    all_converged, worst = True, 0.0
    for i in range(args.warmup, args.warmup + args.steps):
        info = one_step(i)
        newton += info.newton_iterations
        kry += info.krylov_iterations_momentum
        r = list(info.newton_residuals[: info.newton_iterations + 1])
        all_converged &= bool(info.converged) and (r[-1] < 1e-10 or r[-1] / r[0] < 1e-9)
        worst = max(worst, r[-1] / r[0])
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
    # ---- invariants of the computed state: boundary facets of the box whose cell lies in this
    # rank's own layers, summed over the ranks
    nu = 1.0 / args.reynolds
    own_top = mesh.coords[:, 2].min() + part.fine.own_rows * (zlen / nz)
    fb = np.nonzero(mesh.facet_on_boundary)[0]
    mid = mesh.facet_midpoints()[fb]
    cz = mesh.coords[mesh.cells[mesh.facet_cell[fb]].astype(np.int64)][:, :, 2].mean(axis=1)
    mine = cz < own_top
    sides = {"inlet": near(mid[:, 0], 0.0), "outlet": near(mid[:, 0], 2.0), "bottom": near(mid[:, 1], 0.0),
             "top": near(mid[:, 1], 1.0), "back": near(mid[:, 2], 0.0), "front": near(mid[:, 2], zlen)}
    local = []
    for name in sides:
        fc, fl = mesh.facet_cell_local(fb[sides[name] & mine])
        local.append(ctx.boundary_force(fc, fl, nu, 0.0, nat.U0, nat.P)[1] if fc.size else 0.0)
    # inflow: minus the integral of the P2 nodal interpolant of the inlet profile (face rule: area / 3
    # x the three edge-midpoint values); it tends to the analytic -4/9 (x N, weak scaling) like h^4
    fin = fb[sides["inlet"] & mine]
    Xm = dm.p2_coords[dm.facet_p2_nodes(fin)[:, 3:]] if fin.size else np.zeros((0, 3, 3))
    tri = mesh.coords[mesh.facets[fin].astype(np.int64)]
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    local.append(-float((area / 3.0 * profile(Xm).sum(axis=1)).sum()))
    u = ctx.get_state(nat.U0).reshape(-1, 3)
    total = ctx.comm_allreduce(local, "sum")
    flux = dict(zip(sides, (float(v) for v in total[:6])))
    inflow_expected = float(total[6])
    balance = abs(sum(flux.values())) / abs(flux["inlet"])
    inflow_err = abs(flux["inlet"] - inflow_expected)
    umax, bad = ctx.comm_allreduce([float(np.abs(u).max()), 0.0 if np.isfinite(u).all() else 1.0], "max")
    finite = bad == 0.0
    # ---- dominant kernel: finest-level Chebyshev smoothing step of the velocity multigrid inside the
    # block preconditioner (scalar P2 operator on 3 interleaved components), in situ + cache-cold
    ctx.profile_smoother(True)
    ctx.profile_convection(True)
    for i in range(args.warmup + args.steps, args.warmup + args.steps + 2):
        one_step(i)
    ms_sm, n_sm, nbytes = ctx.profile_smoother(False)
    ms_conv, n_conv, nbytes_conv = ctx.profile_convection(False)
    nbytes_conv = abs(nbytes_conv)
    ms_cold, _ = ctx.time_spmv(nat.OP_MOMENTUM_SMOOTHER, 50)
    achieved = nbytes / (ms_sm * 1e-3) / 1e9
    label, extra = _smoother_roofline_extras(ctx, 3, ms_sm, ms_cold)
    if rank == 0:
        print(json.dumps({
            "metric": "dof_updates_per_sec", "value": sps * n_dofs, "unit": "DoF-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 / sps,
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "time_steps_per_sec": sps,
            "config": {"workload": "3D channel flow Re=%g, %g:1:%g box, %dx%dx%d cubes x 6 Kuhn tetrahedra (%d cells, "
                                   "%d dofs), BDF-2 monolithic, parabolic inlet, natural outflow, dt=%g, "
                                   "impulsive start" % (args.reynolds, 2.0, zlen, 2 * n, n, nz, 12 * n * n * nz,
                                                        n_dofs, dt),
                       "n_dofs": n_dofs, "newton_tol": 1e-10, "krylov_rtol": args.krylov_rtol,
                       "newton_forcing": args.newton_forcing,
                       "preconditioner": "block-triangular (V-cycle velocity block, Cahouet-Chabard Schur with "
                                         "the algebraic pressure Laplacian D diag(M)^-1 D^T%s), %d coarse P1 levels" % (
                                             "" if world == 1 else ", additive rank parts", levels),
                       "parallelism": "1 GPU" if world == 1 else
                       "%d slabs of %d cube layers, RCCL halo exchange (forward + reverse-add, overlap %s) + all-reduce" % (
                           world, nz // world, args.overlap),
                       "newton_its_per_step": newton / args.steps,
                       "bicgstab_its_per_step": kry / args.steps, "host_setup_s": t_setup,
                       "host_schur_laplacian_s": t_schur, "comm_per_step_rank0": comm_per_step,
                       "invariants": {"newton_converged_every_step": all_converged,
                                      "worst_newton_reduction": worst,
                                      "flux": flux, "mass_balance_rel": balance,
                                      "inflow_flux_error_vs_interpolated_profile": inflow_err,
                                      "inflow_flux_minus_analytic": flux["inlet"] + 4.0 / 9.0 * zlen,
                                      "max_velocity": float(umax), "finite": bool(finite)}},
            "roofline": dict({"bound": "hbm", "kernel": label,
                              "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                              "traffic": None, "algorithmic_bytes_per_launch": nbytes, "ms_per_launch": ms_sm,
                              "launches_timed": n_sm,
                              "timing": "in situ, HIP-event pairs around the smoothing sequences of 2 steps",
                              "cold_cache": {"achieved": nbytes / (ms_cold * 1e-3) / 1e9,
                                             "frac": nbytes / (ms_cold * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "ms_per_launch": ms_cold}}, **extra),
            "assembly": {"kernel": "k3_conv_cell<FORM,1> + node gather (matrix-free convection action)",
                         "achieved": nbytes_conv / (ms_conv * 1e-3) / 1e9 if n_conv else None, "unit": "GB/s",
                         "frac": nbytes_conv / (ms_conv * 1e-3) / 1e9 / HBM_PEAK_GBS if n_conv else None,
                         "algorithmic_bytes_per_application": nbytes_conv, "ms_per_application": ms_conv,
                         "applications_timed": n_conv}}))
    ctx.close()
    _finish_dist(dist)
    if not (all_converged and finite and balance < 1e-6 and inflow_err < 1e-10):
        raise SystemExit("channel3d-bdf: invariant violated (converged %s, finite %s, mass balance %.2e, "
                         "inflow error %.2e)" % (all_converged, finite, balance, inflow_err))


def _smoother_roofline_extras(ctx, nv, ms_in_situ, ms_cold=None):
    """(kernel label, dict of extra roofline keys) of the finest-level smoothing launch of the velocity
    multigrid -- which kernel runs (nsfem_smoother_info) and, for the stencil-dictionary kernel, the
    rate a CSR stream of the same operator would need to be as fast (continuity with round 1)"""
    info = ctx.smoother_info()
    if info.get("multistep_lattice_kernel"):
        det = ctx.profile_smoother_detail()
        spl = det["steps"] / det["launches"] if det["launches"] else float("nan")
        label = ("k_cheb_lattice<%d,K,4>: %.2f Chebyshev-Jacobi smoothing steps  d = c1 d + c2 D^-1 (b - L x), x += d  of the "
                 "scalar P2 operator L (stencil dictionary: %d distinct rows, longest %d, %s) on %d interleaved components "
                 "per LAUNCH -- the iterate of a 52 x 12-node tile + halo staged once in LDS (parity-class planes), "
                 "stencil table in SGPRs; per launch the vectors are read / written once, not once per step" % (
                     nv, spl, info["stencils"], info["longest_row"],
                     "bitwise equal to the CSR values" if info["bitwise_exact"] else "equal to 2^-40", nv))
        one_step = None
        if ms_in_situ and det["launches"]:
            # what the one-step dictionary kernel of round 2 moved for the same smoothing steps: per step 1 byte per
            # row + mask + 5 vector passes (x, b, d read; d, y written)
            n = ctx.n_velocity
            per_step = n // nv + n + 5 * 8 * n
            one_step = {"note": "the same smoothing steps as one-step dictionary launches (round 2's dominant kernel "
                                "k_spmv_dict_w8<2,3>: 1 B per row + mask + 5 vector passes per step): bytes those launches "
                                "would move per launch of this kernel, and the rate that corresponds to this launch time",
                        "steps_per_launch": spl, "algorithmic_bytes_per_launch": per_step * spl,
                        "achieved": per_step * spl / (ms_in_situ * 1e-3) / 1e9,
                        "frac": per_step * spl / (ms_in_situ * 1e-3) / 1e9 / HBM_PEAK_GBS}
        extra = {"steps_per_launch": spl,
                 "bound_note": "not HBM bound: its own algorithmic bytes (x, b read, x written once per launch) are 1/3 of "
                               "what three one-step launches move; the launch time is set by the tile halos (1.8x the "
                               "algorithmic bytes are loaded), LDS reads (9-19 x 16 B per row and step) and the "
                               "two workgroups per CU the 123 VGPRs allow",
                 "one_step_equivalent": one_step,
                 "csr_equivalent": {
                     "note": "what a CSR stream of the same operator (12 B per nonzero + vectors, once per STEP) would have "
                             "to sustain to match this launch time",
                     "algorithmic_bytes_per_launch": info["csr_bytes"] * spl,
                     "frac": info["csr_bytes"] * spl / (ms_in_situ * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_in_situ else None}}
    elif info["kind"] == "stencil-dictionary":
        label = ("%s: finest-level Chebyshev-Jacobi smoothing step y = x + c1 d + c2 dinv (b - L x) on "
                 "the stencil-dictionary copy of the scalar P2 operator L (%d distinct rows, longest %d, %s), %d "
                 "interleaved components: 1 byte per row + the vectors instead of 12 bytes per nonzero" % (
                     "k_spmv_dict<3,3>" if nv == 3 else "k_spmv_dict_w8<%d,3>" % nv, info["stencils"], info["longest_row"],
                     "bitwise equal to the CSR values" if info["bitwise_exact"] else "equal to 2^-40", nv))
        extra = {"csr_equivalent": {
            "note": "what a CSR stream of the same operator (12 B per nonzero + vectors) would have to sustain to "
                    "match this launch time; comparable with the round-1 figures of k_spmv_stream / k_spmv_sell",
            "algorithmic_bytes_per_launch": info["csr_bytes"],
            "achieved": info["csr_bytes"] / (ms_in_situ * 1e-3) / 1e9 if ms_in_situ else None,
            "frac": info["csr_bytes"] / (ms_in_situ * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_in_situ else None,
            "cold_cache_frac": info["csr_bytes"] / (ms_cold * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_cold else None}}
    else:
        label = ("%s<1,%d,...>: finest-level Chebyshev-Jacobi smoothing step y = x + c1 d + c2 dinv (b - L x) on the "
                 "scalar P2 operator L, %d interleaved components" % (
                     "k_spmv_sell" if info["kind"] == "sell-64" else "k_spmv_stream_v1", nv, nv))
        extra = {}
    return label, extra


def _issue_floor(n):
    """VALU floor of the dominant kernel from the committed SQ counter passes (profiles/r04_cheb_lattice_sq_counters_n512.json:
    SQ_INSTS_VALU per launch x 4 cycles / 1024 SIMDs / 2.4 GHz); {} when no such file exists for the size"""
    f = os.path.join(ROOT, "profiles", "r04_cheb_lattice_sq_counters_n%d.json" % n)
    if not os.path.exists(f):
        return {}
    c = json.load(open(f))
    valu = c.get("SQ_INSTS_VALU")
    if not valu:
        return {}
    return {"valu_floor_us": valu * 4.0 / 1024.0 / 2.4e9 * 1e6,
            "valu_floor_source": "SQ_INSTS_VALU per finest-level launch (%s) x 4 cycles / 1024 SIMDs / 2.4 GHz; waves %s, "
                                 "VALU instructions per wave %s" % (os.path.basename(f), c.get("SQ_WAVES"),
                                                                   round(valu / c["SQ_WAVES"], 1) if c.get("SQ_WAVES") else None)}


def _trace_numbers(n, world):
    """launches / kernel time per step of THIS command from the committed rocprofv3 kernel trace of its timed steps
    (scripts/collect_profiles_r04.sh: bench.py --timed-only --trace-markers under rocprofv3 --kernel-trace, cut to the
    timed region by scripts/trace_gaps.py); not measurable inside the process"""
    f = os.path.join(ROOT, "profiles", "r04_bench_n%d_timed_steps_trace_summary.json" % n)
    if world != 1 or not os.path.exists(f):
        return {}
    t = json.load(open(f))
    return {"launches_per_step": t["launches_per_step"], "kernel_ms_per_step": t["kernel_ms_per_step"],
            "small_launch_ms_per_step": t["small_launch_ms_per_step"],
            "small_launches_per_step": t["small_launches_per_step"],
            "host_round_trips_per_step": t.get("host_round_trips_per_step"),
            "trace_source": "profiles/%s (committed kernel trace of the timed steps of this command; small = launches "
                            "under 15 us)" % os.path.basename(f)}


def _other_configs(args):
    """BASELINE configs[2..4] at their bench sizes in the default job, a few timed steps each, every one in a process
    of its own (fresh context, its own device memory): the driver's run then times them as well"""
    import subprocess
    out = []
    jobs = [("dfg-bdf", ["--workload", "dfg-bdf"]),
            ("tgv3d-ipcs", ["--workload", "tgv3d-ipcs", "--cells", "64"]),
            ("channel3d-bdf", ["--workload", "channel3d-bdf", "--cells", "48"])]
    for name, extra in jobs:
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", "10", "--warmup", "3", "--no-cpu-baseline"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            d = json.loads(line[-1]) if line else None
        except Exception as exc:                       # a failed side job must not take the headline down
            d, r = None, None
            err = repr(exc)
        if d is None:
            out.append({"workload": name, "error": (r.stderr[-300:] if r is not None else err)})
            continue
        cfg = d.get("config", {})
        out.append({"workload": cfg.get("workload", name), "ms_per_step": d.get("ms_per_step"), "value": d.get("value"),
                    "unit": d.get("unit"), "steps": d.get("steps"), "warmup": d.get("warmup"),
                    "n_dofs": cfg.get("n_dofs"),
                    "newton_its_per_step": cfg.get("newton_its_per_step"),
                    "bicgstab_its_per_step": cfg.get("bicgstab_its_per_step"),
                    "roofline_frac": (d.get("roofline") or {}).get("frac"),
                    "wall_s_incl_setup": time.perf_counter() - t0})
    return out


def _apply_truncation(ctx, args):
    parts = [float(v) for v in str(args.mg_truncation).split(",")]
    ctx.mg_set_truncation(parts[0], parts[1] if len(parts) > 1 else 0.1)
    ctx.mg_set_halo_mode(args.halo_mode)


def _serial_coarsest(n, dim=2):
    """cells across the coarsest mesh on one GPU: the first level with <= 1200 nodes (dense
    solve): 2D 512 -> 32 (1089 nodes), 336 -> 21 (484 nodes), 333 -> 21 through non-nested levels; 3D 64 -> 8 (729),
    48 -> 6 (343)"""
    while (n + 1) ** dim > 1200 and (n % 2 == 0 or n >= 5):
        n = (n + 1) // 2            # (odd sizes continue with non-nested levels, 333 -> 167 -> 84 -> 42 -> 21)
    return n


def _parse_cpu_samples(text):
    """'24:4,32:4,48:3,64:3' -> [(24, 4), (32, 4), (48, 3), (64, 3)] (cells per side : timed steps)"""
    out = []
    for item in text.split(","):
        n, _, k = item.partition(":")
        out.append((int(n), int(k) if k else 3))
    return sorted(out)


def _owned_field_difference(ua, ub, owned, dim, dist):
    """relative L2 and max-norm difference of two velocity fields over the OWNED nodes,
    all-reduced over the ranks"""
    m = np.repeat(owned, dim)
    d = (ua - ub)[m]
    sums = np.array([float(d @ d), float(ub[m] @ ub[m])])
    mx = np.array([float(np.abs(d).max()) if d.size else 0.0, float(np.abs(ub[m]).max()) if d.size else 0.0])
    if dist is not None:
        import torch
        t = torch.from_numpy(sums)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t2 = torch.from_numpy(mx)
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
    return float(np.sqrt(sums[0] / max(sums[1], 1e-300))), float(mx[0] / max(mx[1], 1e-300))


def _owned_pressure_difference(pa, pb, owned, dist):
    """the same for the pressures, compared modulo a constant (enclosed flow)"""
    a, b = pa[owned], pb[owned]
    s = np.array([a.sum(), b.sum(), float(a.size)])
    if dist is not None:
        import torch
        t = torch.from_numpy(s)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    a, b = a - s[0] / s[2], b - s[1] / s[2]
    d = a - b
    sums = np.array([float(d @ d), float(b @ b)])
    if dist is not None:
        import torch
        t = torch.from_numpy(sums)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(np.sqrt(sums[0] / max(sums[1], 1e-300)))


def cavity_ipcs_bench(args):
    """headline workload: BASELINE.json configs[1] (see the module docstring)"""
    rank, world, local_rank, dist = _init_dist(args)
    strong = args.scaling == "strong"

    # ---- local problem: strip `rank` of the global mesh (own rows + ghost row)
    #   weak   : n x (n * world) cells on [0,1] x [0,world]   (fixed work per GPU)
    #   strong : n x n cells on the unit square, cut into `world` strips (fixed total work)
    from partition import StripPartition, global_dof_counts
    n = args.n
    ny_global, height = (n, 1.0) if strong else (n * world, float(world))
    # one GPU: coarsest mesh <= 1200 nodes (dense solve, inverse computed on the device)
    # partitioned: the distributed levels stop at 64 cells across (or where the strips cannot be
    # halved any more); below that every rank runs the replicated global hierarchy without any
    # halo exchange
    part = StripPartition((0.0, 0.0), (1.0, height), n, ny_global, rank, world,
                          coarsest=args.coarsest if args.coarsest else (_serial_coarsest(n) if world == 1 else 64),
                          global_coarsest=None if world == 1 else 8, min_rows=1 if world == 1 else args.min_rows)
    dm = part.dofmap
    device = local_rank
    if os.environ.get("NSFEM_SHARE_GPU"):        # rehearsal of several ranks on a one-GPU box
        device = 0
    ctx = nat.NsfemContext(part.mesh.coords, part.mesh.cells, dm.p2_dofmap, dm.p1_dofmap,
                           dm.n_p2, dm.n_p1, device)
    if dist is not None:
        _attach_comm(ctx, dist, rank, world)
    mg_levels = None
    n2g, n1g = global_dof_counts(n, ny_global)
    if not args.no_multigrid:
        mg_levels = part.attach(ctx, args.mg_degree, args.mg_eig_ratio)
    else:
        ctx.set_partition(rank, world, part.p2_ghost, part.p1_ghost, part.p2_halo, part.p1_halo,
                          n2g, n1g)
    ctx.set_coeffs(1.0, 1.0, 1.0 / 100.0)
    ctx.set_dirichlet(nat.VELOCITY, *cavity_dirichlet(dm, height))
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    n_dofs = 2 * n2g + n1g

    _apply_truncation(ctx, args)
    ctx.set_overlap(args.overlap == "on")
    # projection step by fast diagonalisation.  One context: the P1 space is the whole rectangle lattice.  Strips: the
    # factors of the GLOBAL lattice, every rank keeps the rows of V_y of its own lattice lines (ghost lines included);
    # the solve then costs one all-reduce of (ny + 1) x (n + 1) doubles and no halo exchange (csrc/fastdiag.hip)
    fast_diag = False
    if args.poisson_solver == "fd" and not args.no_multigrid:
        import poisson_fd
        if dist is None:
            lines = poisson_fd.lattice_lines(part.mesh)
            factors = poisson_fd.factors(lines[0], lines[1], np.zeros(0, np.int64)) if lines is not None else None
            if factors is not None:
                ctx.poisson_set_fast_diag(factors)
                fast_diag = True
        elif ny_global <= 2048 and dm.n_p1 % (n + 1) == 0:
            # (taller global lattices -- weak scaling beyond 4 strips of 512 rows -- keep the multigrid-CG solve: the
            # all-reduced array grows with the number of ranks, 16.8 MB per solve at 8 x 512 rows)
            # (the strip's vertices are the lattice lines first ... of the global lattice, vertex id = j (n + 1) + i)
            xs, ys = np.linspace(0.0, 1.0, n + 1), np.linspace(0.0, height, ny_global + 1)
            first = int(part.p1_global[0]) // (n + 1)
            G = np.asarray(dm.p1_coords, dtype=np.float64).reshape(-1, n + 1, 2)
            if (G.shape[0] == dm.n_p1 // (n + 1) and np.abs(G[:, :, 0] - xs[None, :]).max() < 1e-12 and
                    np.abs(G[:, :, 1] - ys[first:first + G.shape[0], None]).max() < 1e-12 * max(1.0, height)):
                ctx.poisson_set_fast_diag(poisson_fd.factors(xs, ys, np.zeros(0, np.int64)), first_line=first)
                fast_diag = True

    def throughput_opts():
        o = ctx.default_step_opts()
        for k in (o.momentum, o.poisson, o.correction):
            k.rtol = args.krylov_rtol
        if mg_levels is not None:
            o.momentum.precond = o.poisson.precond = 1
        if fast_diag:
            o.poisson.precond = 3
        o.correction.precond = 2 if args.mass_solver == "chebyshev" else 0
        o.newton_forcing = args.newton_forcing
        o.matrix_free = args.matrix_free
        o.pressure_extrapolation = 1 if args.pressure_start == "extrapolated" else 0
        return o

    def exact_opts():
        # direct-solver accuracy in every linear solve (rtol 1e-12 of SURVEY 8d's parity runs), exact
        # Newton, Jacobi-CG mass solve, full (untruncated) velocity cycle
        o = ctx.default_step_opts()
        if mg_levels is not None:
            o.momentum.precond = o.poisson.precond = 1
        o.matrix_free = args.matrix_free
        return o

    zeros_v, zeros_p = np.zeros(ctx.n_velocity), np.zeros(ctx.n_p1)

    def reset_state():
        for slot in (nat.U0, nat.U1, nat.U2, nat.USTAR):
            ctx.set_state(slot, zeros_v)
        for slot in (nat.P, nat.P_OLD):
            ctx.set_state(slot, zeros_p)

    def one_step(i, opts):
        ctx.set_bdf((1.0, -1.0, 0.0) if i == 0 else (1.5, -2.0, 0.5), args.dt)
        info = ctx.step_ipcs(opts)
        ctx.advance(0)
        return info

    def timed_run(opts):
        """W untimed + K timed steps from the zero state; (seconds, iteration sums)"""
        reset_state()
        for i in range(args.warmup):
            one_step(i, opts)
        ctx.synchronize()
        if args.trace_markers:                 # (a k_cfl launch on either side of the timed region: scripts/trace_gaps.py)
            ctx.cfl_number(nat.U0, args.dt)
        if dist is not None:
            dist.barrier()
        ctx.comm_stats(reset=True)
        t0 = time.perf_counter()
        its = np.zeros(3)
        for i in range(args.warmup, args.warmup + args.steps):
            info = one_step(i, opts)
            its += (info.newton_iterations, info.krylov_iterations_momentum, info.krylov_iterations_poisson)
        ctx.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if args.trace_markers:
            ctx.cfl_number(nat.U0, args.dt)
        comm = {k: v / args.steps for k, v in ctx.comm_stats().items()}
        if dist is not None:
            import torch
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t[0])
        return elapsed, its / args.steps, comm

    # ---- the timed region of the contract: W warm-up + K timed steps, throughput settings
    opts = throughput_opts()
    elapsed, its, comm_per_step = timed_run(opts)
    steps_per_s = args.steps / elapsed
    u_fast, p_fast = ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD)
    if args.timed_only:
        ctx.close()
        _finish_dist(dist)
        return {"metric": "dof_updates_per_sec", "value": steps_per_s * n_dofs, "unit": "DoF-updates/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
                "scaling": "strong" if strong else "weak", "dtype": "f64", "data": "synthetic",
                "time_steps_per_sec": steps_per_s,
                "config": {"workload": "cavity-ipcs %dx%d, timed steps only" % (n, ny_global),
                           "cells": n, "n_dofs": n_dofs, "poisson_solver": "fd" if fast_diag else "mg",
                           "newton_its_per_step": float(its[0]), "bicgstab_its_per_step": float(its[1]),
                           "poisson_cg_its_per_step": float(its[2]), "comm_per_step_rank0": comm_per_step},
                "roofline": {"frac": None, "ms_per_launch": None}}

    # ---- validation of what was timed: the SAME W + K steps from the same start with
    # direct-solver accuracy (rtol 1e-12, exact Newton, Jacobi-CG mass solve, untruncated cycle);
    # the fields of the timed run must agree to north_star's nonlinear tolerance 1e-6
    ctx.mg_set_truncation(0.0, 0.1)
    elapsed_exact, its_exact, _ = timed_run(exact_opts())
    u_ref, p_ref = ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD)
    _apply_truncation(ctx, args)
    du_l2, du_max = _owned_field_difference(u_fast, u_ref, part.p2_owned, 2, dist)
    dp_l2 = _owned_pressure_difference(p_fast, p_ref, part.p1_owned, dist)
    ms_parity = 1e3 * elapsed_exact / args.steps

    # ---- dominant kernel of the timed steps: the Chebyshev-Jacobi smoothing step of the velocity
    # multigrid on its finest level (scalar P2 operator applied to both components, fused
    # epilogue).  Timed IN SITU on the context's stream -- one HIP-event pair around every run of
    # consecutive launches inside a smoothing sequence -- over 5 further steps with the throughput
    # settings (the event pairs stay out of the timed region above).  The launch's working set
    # (252 MB at n = 512) fits the 256 MiB Infinity Cache, so the in-situ figure is cache assisted;
    # `cold_cache` is the same launch timed between cache-flushing launches (HBM only).
    i_next = args.warmup + args.steps
    assembly = cold = None
    if mg_levels is not None:
        ctx.profile_smoother(True)
        ctx.profile_convection(True)
        for i in range(i_next, i_next + 5):
            one_step(i, opts)
        ms_spmv, n_launches, nbytes = ctx.profile_smoother(False)
        ms_conv, n_conv, nbytes_conv = ctx.profile_convection(False)
        if world == 1:
            ms_cold, _ = ctx.time_spmv(nat.OP_MOMENTUM_SMOOTHER, 100)
            cold = {"achieved": nbytes / (ms_cold * 1e-3) / 1e9, "frac": nbytes / (ms_cold * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "ms_per_launch": ms_cold,
                    "timing": "100 launches, each preceded by a cache-flushing 2x2-block SpMV (%.2f GB streamed); "
                              "flush time measured separately and subtracted (nsfem_time_spmv)" % (
                                  ctx.time_spmv(nat.OP_MOMENTUM_JAC, 3)[1] / 1e9)}
            ms_conv_cold, _ = ctx.time_spmv(nat.OP_CONVECTION_ACTION, 100)
        else:
            ms_conv_cold = None
        if n_conv:
            # (a negative byte count flags the element kernel ALONE: on one GPU the per-node sums of its element
            # vectors run inside the L-product launch, nsfem_profile_convection then brackets k_conv_cell only)
            cells_only = nbytes_conv < 0
            nbytes_conv = abs(nbytes_conv)
            ach = nbytes_conv / (ms_conv * 1e-3) / 1e9
            one_launch = ctx.jacobian_info()["path"] == "lattice-kernel"
            if one_launch:
                assembly = {"kernel": ("k_jac_lattice<FORM,1>: the WHOLE matrix-free action of the velocity Jacobian  y = L x + c_c "
                                       "[d conv(u)/du] x  in one launch (the per-Newton-iteration assembly of the fused step, "
                                       "ns_ipcs_solver.py:136-147): u, x of a 32 x 8-square tile staged in LDS, dictionary product "
                                       "of L from LDS, one cell per thread (7-point quadrature, node positions from the cell-type "
                                       "template: no index loads), element vectors summed per node in LDS in ascending cell "
                                       "order (bitwise equal to k_conv_cell + k_spmv_dict_w8 with node gather), y written once"),
                            "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_application": nbytes_conv,
                            "bytes_formula": "n_p2 (3 x 16 u, x, y + 1 dictionary id + 2 mask bytes) [+ n_cells x 48 vertex coordinates on non-uniform lattices: a uniform lattice takes the geometry of its two cell types from 10 scalar loads]",
                            "ms_per_application": ms_conv, "applications_timed": n_conv,
                            "bound_note": ("not HBM bound: 7 x 130 fp64 operations per cell put the VALU floor at 15 us of the "
                                           "launch; the rest is the latency chain of a wave (loads, 9 barriers, LDS phases) at "
                                           "the 4 waves per SIMD its 126 VGPRs allow (DESIGN.md 4d)"),
                            "timing": "HIP-event pair around each application inside the BiCGStab solves of 5 solver steps"}
            else:
              assembly = {"kernel": ("k_conv_cell<FORM,1>: element kernel of the matrix-free action of the convection blocks of "
                                     "the velocity Jacobian (the per-Newton-iteration assembly of the fused step; one thread "
                                     "per cell, element vectors stored node-sorted); their per-node sums run inside the "
                                     "L-product launch k_spmv_dict_w8<2,0> (fused node gather)") if cells_only else
                                    ("k_conv_cell<FORM,1> + k_res_gather: matrix-free action of the convection "
                                     "blocks of the velocity Jacobian (the per-Newton-iteration assembly of the "
                                     "fused step; one thread per cell, element vectors gathered per node in fixed order)"),
                          "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_application": nbytes_conv,
                          "bytes_formula": ("n_cells (48 coords + 24 dof ids + 2 x 96 nodal values of u, x + 96 element vector "
                                            "stored)" if cells_only else
                                            "n_cells (48 coords + 24 dof ids + 2 x 96 nodal values of u, x) + n_dofs_velocity x 16 "
                                            "(SURVEY.md 8d: vector assembly; the element buffer the two kernels hand "
                                            "over, 2 x 96 B per cell, is implementation traffic and not counted)"),
                          "ms_per_application": ms_conv, "applications_timed": n_conv,
                          "timing": "HIP-event pair around each application inside the BiCGStab solves of 5 solver steps"}
            if ms_conv_cold is not None and not cells_only and not one_launch:
                assembly["cold_cache"] = {"achieved": nbytes_conv / (ms_conv_cold * 1e-3) / 1e9,
                                          "frac": nbytes_conv / (ms_conv_cold * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "ms_per_application": ms_conv_cold}
    else:
        ms_spmv, nbytes = ctx.time_spmv(nat.OP_MOMENTUM_JAC, 200)
        n_launches = 200
    achieved = nbytes / (ms_spmv * 1e-3) / 1e9
    ms_jac, nbytes_jac = ctx.time_spmv(nat.OP_MOMENTUM_JAC, 200)
    # HBM bytes per launch of the dominant kernel: NOT measurable inside this process (hardware
    # counters need rocprofv3); the figure below is read from the committed PMC passes of this very
    # command and labelled so (`traffic_source`), null when no such profile exists for the size
    traffic = traffic_source = None
    lattice = mg_levels is not None and ctx.smoother_info().get("multistep_lattice_kernel")
    pmc4 = os.path.join(ROOT, "profiles", "r04_bench_n512_pmc_fetch_write_size.json")
    if world == 1 and n == 512 and lattice and os.path.exists(pmc4):
        # round 4: the MEAN over the finest-level launches (largest grid of the kernel symbol) of 2 x FETCH_SIZE +
        # WRITE_SIZE -- the figure that pairs with the average `algorithmic_bytes_per_launch` over the same launch shapes
        c = json.load(open(pmc4)).get("by_grid", {})
        keys = [k for k in c.get("fetch", {}) if k.startswith("void nsfem::k_cheb_lattice<2, 3, 4") and " @ grid" in k]
        if keys:
            key = max(keys, key=lambda k: int(k.rsplit(" ", 1)[1]))
            if key in c.get("write", {}):
                traffic = (2.0 * c["fetch"][key]["mean_KB"] + c["write"][key]["mean_KB"]) * 1024.0
                traffic_source = ("committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (profiles/"
                                  "r04_bench_n512_pmc_fetch_write_size.json, by_grid): MEAN over the %d finest-level launches "
                                  "of 2 x FETCH_SIZE (gfx950 wide-read correction) + WRITE_SIZE; not measured by this run; "
                                  "counts Infinity-Cache hits as well" % c["fetch"][key]["n"])
    for fname in ("r03_b_bench_n512_pmc_fetch_write_size.json", "r03_a_bench_n512_pmc_fetch_write_size.json",
                  "r02_pmc_fetch_write_size.json"):
        if traffic is not None:
            break
        pmc = os.path.join(ROOT, "profiles", fname)
        if world == 1 and n == 512 and mg_levels is not None and os.path.exists(pmc):
            # 2 x FETCH_SIZE (gfx950 wide-read correction, MI355X_MICROARCH.md section HBM) + WRITE_SIZE; a kernel
            # symbol is launched on several multigrid levels, the finest-level launches are the maxima
            c = json.load(open(pmc))
            if lattice:
                key = "void nsfem::k_cheb_lattice<2, 3, 4>"
                shape = ("; the maxima belong to the launch shape of the Chebyshev mass solve with a carried direction "
                         "(x, b, d read; x, d written = 85.1 MB algorithmic), not to the %.1f MB average of `algorithmic_"
                         "bytes_per_launch`" % (nbytes / 1e6))
            else:
                key = "void nsfem::k_spmv_dict_w8<2, 3>" if ctx.smoother_info()["kind"] == "stencil-dictionary" \
                    else "void nsfem::k_spmv_stream_v1<1, 1, 2, 3>"
                shape = ""
            if key in c["fetch"] and key in c["write"]:
                traffic = (2.0 * c["fetch"][key]["max_KB"] + c["write"][key]["max_KB"]) * 1024.0
                traffic_source = "committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command: " \
                                 "profiles/%s (not measured by this run; counts Infinity-Cache hits as well)%s" % (
                                     fname, shape)
                break
    tol_fields = 1.0e-6
    smoother_label, smoother_extra = (_smoother_roofline_extras(ctx, 2, ms_spmv, cold["ms_per_launch"] if cold else None)
                                      if mg_levels is not None else ("", {}))
    out = {
        "metric": "dof_updates_per_sec", "value": steps_per_s * n_dofs, "unit": "DoF-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "time_steps_per_sec": steps_per_s,
        "config": {"workload": "2D lid-driven cavity Re=100, %dx%d right-diagonal Taylor-Hood P2/P1 "
                               "(%d dofs), IPCS, BDF-2, dt=%g, zero initial state" % (
                                   n, ny_global, n_dofs, args.dt),
                   "n_dofs": n_dofs, "newton_tol": 1e-10, "krylov_rtol": args.krylov_rtol,
                   "newton_forcing": args.newton_forcing,
                   "velocity_jacobian": {0: "matrix-free (default)", 1: "assembled block CSR",
                                         2: "matrix-free"}[args.matrix_free],
                   "ms_per_step_with_krylov_rtol_1e-12_exact_newton": ms_parity,
                   "validation": {"what": "the %d timed + %d warm-up steps repeated from the same start with "
                                          "Krylov rtol 1e-12, exact Newton, Jacobi-CG mass solve and the "
                                          "untruncated velocity cycle; fields of the timed run vs those" % (
                                              args.steps, args.warmup),
                                  "max_rel_diff_velocity_vs_exact": du_l2,
                                  "max_rel_diff_velocity_vs_exact_maxnorm": du_max,
                                  "max_rel_diff_pressure_vs_exact": dp_l2,
                                  "tolerance": tol_fields,
                                  "newton_bicgstab_poisson_its_per_step_exact": [float(v) for v in its_exact]},
                   "poisson_solver": ("fast diagonalisation: x += V_y ((V_y^T R V_x) .* inv) V_x^T, four fp64 MFMA products "
                                      "(csrc/fastdiag.hip), one pass + residual check" +
                                      ("; strips: every rank contracts its own lattice lines, ONE all-reduce of the "
                                       "%d x %d transformed array per solve, no halo exchange" % (ny_global + 1, n + 1)
                                       if dist is not None else "") if fast_diag else
                                      "CG preconditioned by the pressure V(2,2) cycle"),
                   "preconditioner": "jacobi" if mg_levels is None else
                   "geometric multigrid, Chebyshev-Jacobi smoothing: V(0,3) momentum, V(2,2) Poisson; %d coarse P1 levels" % mg_levels,
                   "parallelism": "1 GPU" if world == 1 else
                   "%d strips of %d cell rows, RCCL halo exchange (%s mode, overlap %s) + all-reduce" % (
                       world, ny_global // world, args.halo_mode, args.overlap),
                   "newton_its_per_step": float(its[0]),
                   "bicgstab_its_per_step": float(its[1]), "poisson_cg_its_per_step": float(its[2]),
                   "comm_per_step_rank0": comm_per_step},
        "roofline": {"bound": "latency/issue" if lattice else "hbm",
                     "kernel": smoother_label if mg_levels is not None else "k_spmv_stream_v1<2,2,1,0> (momentum Jacobian)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "algorithmic_bytes_per_launch": nbytes, "ms_per_launch": ms_spmv,
                     "launches_timed": n_launches,
                     "timing": "in situ: HIP-event pairs around the runs of consecutive finest-level launches (one pair "
                               "per smoothing sequence) during 5 solver steps",
                     "cache_note": "`achieved` / `frac` are in-situ figures; when the launch's working set "
                                   "(algorithmic_bytes_per_launch) is below the 256 MiB Infinity Cache they are cache "
                                   "assisted -- `cold_cache` is the HBM-only figure of the same launch",
                     "cold_cache": cold},
        "assembly": assembly,
        "jacobian_spmv": {"kernel": "k_spmv_stream<2,2,1,0> (assembled momentum Jacobian, 2x2 block CSR; "
                                    "used by the explicit assembly seam / matrix_free=1)",
                          "achieved": nbytes_jac / (ms_jac * 1e-3) / 1e9, "unit": "GB/s",
                          "frac": nbytes_jac / (ms_jac * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "algorithmic_bytes_per_launch": nbytes_jac, "ms_per_launch": ms_jac},
    }
    out["roofline"].update(smoother_extra)
    if lattice:
        out["roofline"].update(_issue_floor(n))
    out.update(_trace_numbers(n, world))
    ctx.close()
    _finish_dist(dist)
    if world == 1 and not args.no_solver_classes:
        # the same W + K steps through the reference's solver surface: IPCSSolver.solve() /
        # advance_time() inside the InstationaryProblem loop (BASELINE.md section 3 protocol)
        out["config"].update(solver_surface_run(args, n, u_fast, p_fast))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(_parse_cpu_samples(args.cpu_samples), args.dt, n_dofs)
    if rank == 0 and world == 1 and args.other_configs and getattr(args, "thread_ranks", None) is None:
        out["other_configs"] = _other_configs(args)
    if max(du_l2, dp_l2) > tol_fields:
        raise SystemExit("bench: the timed fields differ from the exact-solver run by %.2e (velocity) / %.2e "
                         "(pressure) > %.0e" % (du_l2, dp_l2, tol_fields))
    return out


def solver_surface_run(args, n, u_abi, p_abi):
    """The timed configuration THROUGH THE REFERENCE'S SOLVER SURFACE: an InstationaryProblem
    subclass with the cavity's hook methods (demo/cavity_flow.py:21-37 set-up), IPCSSolver selected by
    set_solver_class, the throughput settings passed as `solver_settings`; W warm-up steps run inside
    solve_problem(), then K steps of the reference's loop body (source/ns_problem.py:711-727:
    update_coefficients, solver.solve(), advance_time, solver.advance_time) are timed.  Same mesh,
    same settings, same start as the C-ABI run above: the fields must agree BITWISE."""
    import contextlib
    import io
    from auxiliary_classes import EquationCoefficientHandler
    from grid_generator import HyperCubeBoundaryMarkers, hyper_cube
    from ns_ipcs_solver import IPCSSolver
    from ns_problem import InstationaryProblem, VelocityBCType

    class CavityProblem(InstationaryProblem):
        def __init__(self):
            super().__init__(None, start_time=0.0, end_time=args.dt * (args.warmup + args.steps + 8),
                             desired_start_time_step=args.dt, n_max_steps=max(args.warmup, 1))
            self._problem_name = "Cavity"
            self._output_frequency = 0
            self._postprocessing_frequency = 0
            self.compute_cfl = False
            self.set_solver_class(IPCSSolver)
            parts = [float(v) for v in str(args.mg_truncation).split(",")]
            self.solver_settings = dict(
                krylov_rtol=args.krylov_rtol, newton_forcing=args.newton_forcing,
                pressure_start=args.pressure_start if args.pressure_start == "extrapolated" else "previous",
                mass_solver=args.mass_solver, mg_truncation=(parts[0], parts[1] if len(parts) > 1 else 0.1),
                poisson_solver="fast_diagonalization" if args.poisson_solver == "fd" else "multigrid",
                matrix_free={0: None, 1: False, 2: True}[args.matrix_free])

        def setup_mesh(self):
            self._mesh, self._boundary_markers = hyper_cube(2, n)

        def set_boundary_conditions(self):
            m = HyperCubeBoundaryMarkers
            self._bcs = ((VelocityBCType.no_slip, m.left.value, None), (VelocityBCType.no_slip, m.right.value, None),
                         (VelocityBCType.no_slip, m.bottom.value, None),
                         (VelocityBCType.constant, m.top.value, (1.0, 0.0)))

        def set_equation_coefficients(self):
            self._coefficient_handler = EquationCoefficientHandler(Re=100.0)

        def set_initial_conditions(self):
            self._initial_conditions = {"velocity": (0.0, 0.0), "pressure": 0.0}

    if args.warmup == 0:
        raise SystemExit("--warmup 0 is not supported by the solver-surface timing (--no-solver-classes)")
    log = io.StringIO()
    with contextlib.redirect_stdout(log):
        problem = CavityProblem()
        t_setup = time.perf_counter()
        problem.solve_problem()                       # set-up + the W warm-up steps
        t_setup = time.perf_counter() - t_setup
        solver, ts = problem._get_solver(), problem._time_stepping
        ctx = solver._ctx
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):                   # body of the loop at source/ns_problem.py:711-727
            problem._set_next_step_size()
            ts.update_coefficients()
            solver.solve()
            ts.advance_time()
            solver.advance_time()
        ctx.synchronize()
        elapsed = time.perf_counter() - t0
        u, p = ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD)
        ctx.close()
    du = float(np.abs(u - u_abi).max()) if u.shape == u_abi.shape else None
    dp = float(np.abs((p - p.mean()) - (p_abi - p_abi.mean())).max()) if p.shape == p_abi.shape else None
    return {"ms_per_step_through_solver_classes": 1e3 * elapsed / args.steps,
            "solver_classes": {"what": "InstationaryProblem hooks + IPCSSolver.solve() / advance_time(), "
                                       "solver_settings = the throughput settings; %d warm-up steps inside "
                                       "solve_problem(), then %d timed passes of the loop body" % (args.warmup, args.steps),
                               "max_abs_velocity_difference_vs_the_c_abi_run": du,
                               "max_abs_pressure_difference_vs_the_c_abi_run": dp,
                               "difference_note": "the solver classes take the step size from DiscreteTime as the reference "
                                                  "does (t_next - t_current in floating point, e.g. 0.0010000000000000009), "
                                                  "the C-ABI loop above passes dt itself: round-off level; with the SAME "
                                                  "(alpha, k) sequence the two paths agree bit for bit "
                                                  "(tests/test_solver_classes_gpu.py)",
                               "set_up_and_warm_up_s": t_setup}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cells", dest="n", type=int, default=None,
                    help="cells per side (weak scaling: per rank; strong scaling: of the whole mesh); "
                         "default 512 = BASELINE config 2 (strong scaling: 960, 8.3 M dofs)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="--gpus N > 1: weak = every rank a --cells x --cells strip (per-GPU work fixed); "
                         "strong = ONE --cells x --cells mesh cut into N strips (total work fixed)")
    ap.add_argument("--overlap", choices=("on", "off"), default="on",
                    help="N > 1: halo exchange on a second stream under the interior rows of the SpMV")
    ap.add_argument("--dt", type=float, default=1.0e-3)
    ap.add_argument("--krylov-rtol", type=float, default=1.0e-8,
                    help="relative residual of the linear solves (Poisson, mass; Newton solves when exact)")
    ap.add_argument("--newton-forcing", type=float, default=1.0e-4,
                    help="inexact Newton: reduce each Newton linear residual only by this factor "
                         "(0 = exact Newton with --krylov-rtol)")
    ap.add_argument("--cpu-samples", default=None,
                    help="CPU baseline ladder n:steps,... (cavity cells per side : timed steps); default "
                         "24:4,32:4,48:3,64:3 (~15 s of CPU work), plus 128:1 (~13 s more, 148,739 dofs: the "
                         "extrapolation to the workload then spans 16x in size instead of 63x) on hosts with >= 64 cores")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-solver-classes", action="store_true",
                    help="skip the timing of the same steps through IPCSSolver.solve() / advance_time()")
    ap.add_argument("--no-strong", action="store_true",
                    help="--gpus N > 1: skip the strong-scaling run that the default job adds to the weak one")
    ap.add_argument("--strong-cells", type=int, default=None,
                    help="--gpus N > 1: cells per side of the strong-scaling mesh of the default job "
                         "(default 960 = 8.3 M dofs; with --cells n: n)")
    ap.add_argument("--timed-only", action="store_true",
                    help="profiling aid: only the warm-up and timed steps (no validation rerun, no in-situ / "
                         "cold-cache kernel timing, no CPU baseline); the JSON line lacks those entries")
    ap.add_argument("--no-other-configs", dest="other_configs", action="store_false",
                    help="skip the short runs of BASELINE configs[2..4] (dfg-bdf, tgv3d-ipcs 64, channel3d-bdf 48) that the "
                         "default job appends as `other_configs`")
    ap.add_argument("--trace-markers", action="store_true",
                    help="profiling aid: one k_cfl launch right before and right after the timed steps, so that a kernel "
                         "trace can be cut to the timed region (scripts/trace_gaps.py --between k_cfl)")
    ap.add_argument("--no-multigrid", action="store_true")
    ap.add_argument("--mg-degree", type=int, default=None,
                    help="Chebyshev smoother degree (default: 2 on structured meshes, 3 on the DFG mesh)")
    ap.add_argument("--mg-eig-ratio", type=float, default=None,
                    help="smoothing interval [lmax / ratio, lmax] (default: 4 structured, 16 DFG)")
    ap.add_argument("--matrix-free", type=int, default=0, choices=(0, 1, 2),
                    help="velocity Jacobian in the step driver: 0 auto, 1 assembled, 2 matrix-free")
    ap.add_argument("--mass-solver", choices=("chebyshev", "cg"), default="chebyshev",
                    help="velocity-correction mass solve: Chebyshev with a-priori bounds (no dots) or Jacobi-CG")
    ap.add_argument("--poisson-solver", choices=("fd", "mg"), default="fd",
                    help="projection step (one GPU, rectangle lattices): fd = direct solve by fast diagonalisation (four "
                         "dense products on the matrix cores, csrc/fastdiag.hip), mg = multigrid-preconditioned CG")
    ap.add_argument("--pressure-start", choices=("extrapolated", "previous"), default="extrapolated",
                    help="start vector of the projection-step CG: 2 p_n - p_(n-1) or p_n")
    ap.add_argument("--halo-mode", choices=("relaxed", "exact"), default="relaxed",
                    help="N > 1: multigrid smoothing with one halo exchange per smoothing sequence "
                         "(frozen ghosts in between) or per SpMV (the serial algorithm)")
    ap.add_argument("--mg-truncation", default="4,0.1",
                    help="R[,TOL]: truncate the velocity multigrid cycle at the first level with "
                         "c_v K_ii <= R alpha0/k M_ii, solved there by Chebyshev iteration to TOL (0: off)")
    ap.add_argument("--coarsest", type=int, default=0, help="cells across the coarsest multigrid mesh (0: default)")
    ap.add_argument("--min-rows", type=int, default=16,
                    help="N > 1 (cavity-ipcs): partitioned multigrid levels keep at least this many cell rows per rank; "
                         "coarser levels run replicated on every rank (no halo exchanges)")
    ap.add_argument("--workload", choices=("cavity-ipcs", "dfg-bdf", "cavity3d-ipcs", "cavity3d-bdf", "tgv3d-ipcs",
                                           "channel3d-bdf"),
                    default="cavity-ipcs",
                    help="cavity-ipcs = BASELINE configs[1] (headline); dfg-bdf = configs[2], 1 GPU; "
                         "cavity3d-* = 3D tetrahedral cavity (--cells cubes per side); tgv3d-ipcs = configs[3] "
                         "(triple-periodic Taylor-Green vortex); channel3d-bdf = configs[4] (3D channel Re=1000, "
                         "2:1:1 box, BDF-2 monolithic, open outlet)")
    ap.add_argument("--dfg-refine", type=int, default=5)
    ap.add_argument("--reynolds", type=float, default=1000.0, help="channel3d-bdf: Reynolds number")
    ap.add_argument("--local-ranks", type=int, default=0,
                    help="functional rehearsal of the N-rank paths on ONE GPU: N threads of this process, "
                         "in-process communicator (no RCCL; not a performance mode)")
    args = ap.parse_args()
    if args.cpu_samples is None:
        args.cpu_samples = "24:4,32:4,48:3,64:3" + (",128:1" if (os.cpu_count() or 1) >= 64 else "")
    if args.local_ranks > 1:
        import copy
        import threading
        args.gpus = args.local_ranks
        args.no_cpu_baseline = True
        shared = _ThreadRanks(args.local_ranks)
        failures = []

        def rank_main(r):
            mine = copy.copy(args)
            mine.thread_ranks, mine.thread_rank = shared, r
            shared.bind(r)
            try:
                _run_workload(mine)
            except BaseException as exc:       # a dead rank would leave the others in a barrier
                import traceback
                traceback.print_exc()
                failures.append((r, exc))
                os._exit(3)

        workers = [threading.Thread(target=rank_main, args=(r,)) for r in range(args.local_ranks)]
        for w in workers:
            w.start()
        for w in workers:
            w.join()
        nat.local_group_destroy(shared.group)
        return None
    try:
        return _run_workload(args)
    finally:
        if "torch.distributed" in sys.modules:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()


def _run_workload(args):
    if args.workload == "dfg-bdf":
        return dfg_bdf_bench(args)
    if args.workload.startswith("cavity3d"):
        args.n = args.n or 32
        return cavity3d_bench(args)
    if args.workload == "tgv3d-ipcs":
        args.n = args.n or (64 if args.scaling == "strong" else 32)
        return tgv3d_bench(args)
    if args.workload == "channel3d-bdf":
        args.n = args.n or 32
        return channel3d_bdf_bench(args)
    import copy
    rank = args.thread_rank if getattr(args, "thread_ranks", None) is not None else int(os.environ.get("RANK", "0"))
    default_job = args.n is None and not args.timed_only
    both = args.gpus > 1 and args.scaling == "weak" and not args.timed_only and not args.no_strong
    if args.strong_cells is None:
        args.strong_cells = 960 if args.n in (None, 512) else args.n

    def strong_block(st, a2):
        return {"cells": a2.n, "n_dofs": st["config"]["n_dofs"], "ms_per_step": st["ms_per_step"],
                "value": st["value"], "unit": st["unit"], "time_steps_per_sec": st["time_steps_per_sec"],
                "steps": st["steps"], "warmup": st["warmup"],
                "its_per_step_newton_bicgstab_poisson": [st["config"]["newton_its_per_step"],
                                                         st["config"]["bicgstab_its_per_step"],
                                                         st["config"]["poisson_cg_its_per_step"]],
                "comm_per_step": st["config"]["comm_per_step_rank0"]}

    if both:
        # N > 1 (round 4): the HEADLINE of the line is north_star's strong-scaling quantity -- ONE 960 x 960 mesh
        # (8.3 M dofs) cut into N strips, `"scaling": "strong"` --, the weak-scaling figure (512 x 512 cells per rank,
        # so N = 1 equals the single-GPU line) rides along as the `weak` block.  The one-GPU point of the strong
        # curve is the `strong_n1` block of the N = 1 line (same mesh on one GPU).
        a2 = copy.copy(args)
        a2.scaling, a2.n = "strong", args.strong_cells
        a2.no_cpu_baseline = a2.no_solver_classes = True
        a2.other_configs = False
        out = cavity_ipcs_bench(a2)
        a1 = copy.copy(args)
        a1.n, a1.timed_only = args.n or 512, True
        wk = cavity_ipcs_bench(a1)
        if out is not None and wk is not None:
            out["weak"] = strong_block(wk, a1)
            out["weak"]["note"] = "512 x 512 cells PER RANK (fixed work per GPU); value = all ranks' DoF-updates per second"
    else:
        args.n = args.n or (960 if args.scaling == "strong" else 512)
        out = cavity_ipcs_bench(args)
        if default_job and args.gpus == 1 and out is not None and args.scaling == "weak" and not args.no_strong:
            # the one-GPU point of the strong-scaling curve the N > 1 lines report: the 960 x 960 mesh on one GPU
            a2 = copy.copy(args)
            a2.scaling, a2.n, a2.timed_only = "strong", args.strong_cells, True
            a2.steps, a2.warmup = min(args.steps, 10), min(args.warmup, 3)
            st = cavity_ipcs_bench(a2)
            out["strong_n1"] = strong_block(st, a2)
            out["strong_n1"]["note"] = ("the mesh of the strong-scaling curve (--gpus N > 1 reports it as its headline) on "
                                        "ONE GPU: divide the N-GPU `value` by this one for the speed-up")
    if rank == 0:
        print(json.dumps(out))
    return None


if __name__ == "__main__":
    main()
