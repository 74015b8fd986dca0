"""Scratch: partitioned IPCS with N in-process ranks (threads) on one GPU vs the single context."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import _native as nat
from gpu_common import box, cavity_bc, context, rel
from multigrid import attach_hierarchy
from partition import StripPartition

n = int(sys.argv[1]); size = int(sys.argv[2]); nsteps = int(sys.argv[3]); k = float(sys.argv[4])
use_mg = int(sys.argv[5]) if len(sys.argv) > 5 else 1
coarsest = int(sys.argv[6]) if len(sys.argv) > 6 else 4
part_coarsest = int(sys.argv[7]) if len(sys.argv) > 7 else coarsest     # partitioned levels stop here
forcing = float(sys.argv[8]) if len(sys.argv) > 8 else 0.0
relaxed = int(sys.argv[9]) if len(sys.argv) > 9 else 0

def lid_bc(dm):
    """cavity BC on whatever boundary nodes of the unit square the (local) dof map holds"""
    X = dm.p2_coords
    on = (np.abs(X[:, 0]) < 1e-12) | (np.abs(X[:, 0] - 1) < 1e-12) | (np.abs(X[:, 1]) < 1e-12) | (np.abs(X[:, 1] - 1) < 1e-12)
    nodes = np.nonzero(on)[0]
    lid = np.abs(X[nodes, 1] - 1.0) < 1e-12
    dofs = np.concatenate([2 * nodes, 2 * nodes + 1])
    vals = np.concatenate([np.where(lid, 1.0, 0.0), np.zeros(nodes.size)])
    return dofs.astype(np.int32), vals

def setup(ctx, dm):
    ctx.set_coeffs(1.0, 1.0, 0.01)
    ctx.set_dirichlet(nat.VELOCITY, *lid_bc(dm))
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))

def run(ctx, out, key):
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = 1e-12
    if use_mg:
        opts.momentum.precond = opts.poisson.precond = 1
    if forcing:
        opts.newton_forcing = forcing
        for o in (opts.momentum, opts.poisson, opts.correction):
            o.rtol = 1e-8
        opts.correction.precond = 2
    infos = []
    ctx.comm_stats(reset=True)
    for step in range(nsteps):
        ctx.set_bdf((1.0, -1.0, 0.0) if step == 0 else (1.5, -2.0, 0.5), k)
        infos.append(ctx.step_ipcs(opts))
        ctx.advance(0)
    out[key] = (ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD), infos)
    if key == 0:
        print("comm per step (rank 0):", {k: v / nsteps for k, v in ctx.comm_stats().items()})

# ---- single context reference
mesh, dm, marks = box(n, n)
ref = {}
ctx0 = context(mesh, dm)
if use_mg:
    attach_hierarchy(ctx0, mesh, coarsest=coarsest)
setup(ctx0, dm)
t0 = time.time(); run(ctx0, ref, 0); t_ref = time.time() - t0
u_ref, p_ref, inf_ref = ref[0]
print("serial: %.3fs  newton %s  kry mom %s poi %s" % (t_ref, [i.newton_iterations for i in inf_ref],
      [i.krylov_iterations_momentum for i in inf_ref], [i.krylov_iterations_poisson for i in inf_ref]))

# ---- partitioned, one thread per rank
group = nat.local_group_create(size)
parts = [StripPartition((0, 0), (1, 1), n, n, r, size, coarsest=part_coarsest,
                        global_coarsest=coarsest if part_coarsest != coarsest else None) for r in range(size)]
ctxs = []
for r, part in enumerate(parts):
    pdm = part.dofmap
    c = nat.NsfemContext(part.mesh.coords, part.mesh.cells, pdm.p2_dofmap, pdm.p1_dofmap, pdm.n_p2, pdm.n_p1)
    c.attach_local_comm(group, r)
    ctxs.append(c)
out, errs = {}, []
def worker(r):
    try:
        part = parts[r]
        if use_mg:
            part.attach(ctxs[r])
        else:
            n2g, n1g = (2 * n + 1) ** 2, (n + 1) ** 2
            ctxs[r].set_partition(r, size, part.p2_ghost, part.p1_ghost, part.p2_halo, part.p1_halo, n2g, n1g)
        ctxs[r].mg_set_halo_mode(relaxed)
        setup(ctxs[r], part.dofmap)
        run(ctxs[r], out, r)
    except Exception as e:
        errs.append((r, repr(e)))
        os._exit(3)
t0 = time.time()
threads = [threading.Thread(target=worker, args=(r,)) for r in range(size)]
[t.start() for t in threads]; [t.join() for t in threads]
print("partitioned (%d ranks): %.3fs" % (size, time.time() - t0))
u = np.zeros_like(u_ref); p = np.zeros_like(p_ref)
for r, part in enumerate(parts):
    ul, pl, infos = out[r]
    own2, own1 = part.p2_owned, part.p1_owned
    u.reshape(-1, 2)[part.p2_global[own2]] = ul.reshape(-1, 2)[own2]
    p[part.p1_global[own1]] = pl[own1]
    # ghost copies must equal the owners' values
    if r == 0:
        print("rank0: newton %s kry mom %s poi %s cor %s" % ([i.newton_iterations for i in infos],
              [i.krylov_iterations_momentum for i in infos], [i.krylov_iterations_poisson for i in infos],
              [i.krylov_iterations_correction for i in infos]))
print("rel err u %.3e  p(mod const) %.3e" % (rel(u, u_ref), rel(p - p.mean(), p_ref - p_ref.mean())))
for r, part in enumerate(parts):
    ul, pl, _ = out[r]
    print("  rank %d ghost consistency u %.2e p %.2e" % (r, np.abs(ul.reshape(-1, 2) - u.reshape(-1, 2)[part.p2_global]).max(),
          np.abs((pl - pl[part.p1_owned].mean()) - (p - p.mean())[part.p1_global] + (p - p.mean())[part.p1_global][part.p1_owned].mean() - 0).max()))
