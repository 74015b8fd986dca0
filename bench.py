#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: time-steps/sec (and DoF-updates/sec) of the
2D lid-driven cavity, Re = 100, IPCS pressure-projection, Taylor-Hood P2/P1, on MI355X.

N = 1 workload = BASELINE.json configs[1]: 512 x 512 right-diagonal triangles
(2,364,419 dofs), k = 1e-3.  One "step" = one full IPCS time step (Newton diffusion step,
pressure Poisson, velocity correction, time-level shift) on device-resident state.

Prints ONE JSON line (see the driver contract) that also carries
  "roofline":     dominant kernel (block-CSR SpMV of the momentum Jacobian), algorithmic
                  bytes per launch / HIP-event time on the kernel's stream vs 8 TB/s HBM;
  "cpu_baseline": the CPU oracle configured as the reference works (full re-assembly +
                  sparse LU every Newton iteration, Poisson/mass re-factorised every step)
                  timed on a bounded sample on this host.
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

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cavity_setup(n, device):
    import _native as nat
    from fem_mesh import TaylorHoodDofMap
    from grid_generator import hyper_cube
    mesh, marks = hyper_cube(2, n)
    dm = TaylorHoodDofMap(mesh)
    ctx = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1,
                           device)
    dofs, vals = [], []
    for mid, val in ((1, (0.0, 0.0)), (2, (0.0, 0.0)), (3, (0.0, 0.0)), (4, (1.0, 0.0))):
        nodes = np.unique(dm.facet_p2_nodes(marks.facets_with_id(mid)))
        for a in range(2):
            dofs.append(2 * nodes + a)
            vals.append(np.full(nodes.size, val[a]))
    ctx.set_coeffs(1.0, 1.0, 1.0 / 100.0)
    ctx.set_dirichlet(nat.VELOCITY, np.concatenate(dofs), np.concatenate(vals))   # lid wins at corners
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    return mesh, dm, ctx, nat


def cpu_baseline(n_sample, k, steps, full_dofs):
    """Reference algorithm on the host CPU (1 process, as the reference runs): per Newton
    iteration full re-assembly + SuperLU; Poisson and mass matrices re-assembled and
    re-factorised each step (dolfin LinearVariationalSolver behaviour)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fem_oracle as fo
    from fem_mesh import TaylorHoodDofMap
    from grid_generator import hyper_cube
    mesh, marks = hyper_cube(2, n_sample)
    dm = TaylorHoodDofMap(mesh)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    last = {}
    for mid, val in ((1, (0.0, 0.0)), (2, (0.0, 0.0)), (3, (0.0, 0.0)), (4, (1.0, 0.0))):
        nodes = np.unique(dm.facet_p2_nodes(marks.facets_with_id(mid)))
        for a in range(2):
            for d in 2 * nodes + a:
                last[int(d)] = val[a]
    bd = np.array(sorted(last))
    bv = np.array([last[d] for d in bd])
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
    return {"value": sps * dm.n_dofs / full_dofs, "unit": "time-steps/s", "cores": 1, "kind": "port",
            "sample": "oracle IPCS (re-assembly + SuperLU per Newton iteration, Poisson/mass "
                      "re-factorised per step) on the n=%d cavity (%d dofs), %d steps after 1 warm-up: "
                      "%.3f steps/s there; value = that rate x %d/%d dofs (linear-in-dofs scaling, "
                      "optimistic for sparse LU)" % (n_sample, dm.n_dofs, steps, sps, dm.n_dofs, full_dofs),
            "sample_steps_per_s": sps, "sample_dofs": dm.n_dofs,
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=512, help="cells per side (512 = BASELINE config 2)")
    ap.add_argument("--dt", type=float, default=1.0e-3)
    ap.add_argument("--krylov-rtol", type=float, default=1.0e-10)
    ap.add_argument("--cpu-sample-n", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-multigrid", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("gloo", rank=rank, world_size=world)

    mesh, dm, ctx, nat = cavity_setup(args.n, local_rank)
    from multigrid import attach_hierarchy
    mg_levels = attach_hierarchy(ctx, mesh) if not args.no_multigrid else None
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = args.krylov_rtol
    if mg_levels is not None:
        opts.momentum.precond = opts.poisson.precond = 1

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
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    # every rank advanced an independent replica of the workload (no mesh partitioning yet)
    steps_per_s = world * args.steps / elapsed
    ms_spmv, nbytes = ctx.time_spmv(nat.OP_MOMENTUM_JAC, 200)
    achieved = nbytes / (ms_spmv * 1e-3) / 1e9
    out = {
        "metric": "time_steps_per_sec", "value": steps_per_s, "unit": "time-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "dof_updates_per_sec": steps_per_s * dm.n_dofs,
        "config": {"workload": "2D lid-driven cavity Re=100, %dx%d right-diagonal Taylor-Hood P2/P1 "
                               "(%d dofs), IPCS, BDF-2, dt=%g, zero initial state" % (
                                   args.n, args.n, dm.n_dofs, args.dt),
                   "n_dofs": dm.n_dofs, "newton_tol": 1e-10, "krylov_rtol": args.krylov_rtol,
                   "preconditioner": "jacobi" if mg_levels is None else "geometric multigrid V(2,2) Chebyshev, %d coarse P1 levels" % mg_levels,
                   "parallelism": "1 GPU" if world == 1 else "%d independent replicas" % world,
                   "newton_its_per_step": newton / args.steps,
                   "bicgstab_its_per_step": kry / args.steps},
        "roofline": {"bound": "hbm", "kernel": "k_spmv<2,2,1,8> (momentum Jacobian, 2x2 block CSR)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": nbytes, "ms_per_launch": ms_spmv},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample_n, args.dt, 2, dm.n_dofs)
    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
