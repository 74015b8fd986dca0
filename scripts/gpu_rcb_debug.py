"""debug: RCB partition pieces against the single context"""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "navierstokes-with-fenics_amd"))
import _native as nat
import grid_generator as gg
from fem_mesh import TaylorHoodDofMap
from partition import GraphPartition
mesh, marks = gg.dfg_channel(4, 2)
dm = TaylorHoodDofMap(mesh)
size = 2
parts = [GraphPartition(mesh, r, size, marks) for r in range(size)]
rng = np.random.default_rng(0)
ug = rng.standard_normal(2 * dm.n_p2)
pg = rng.standard_normal(dm.n_p1)
ctx0 = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1)
def arg(op, u, p):
    return u if op == nat.OP_DIV else (p if op == nat.OP_STIFF_P1 else u[::2].copy())
ref = {op: ctx0.operator_apply(op, arg(op, ug, pg)) for op in (nat.OP_MASS_P2, nat.OP_STIFF_P2, nat.OP_DIV, nat.OP_STIFF_P1)}
for part in parts:
    pdm = part.dofmap
    c = nat.NsfemContext(part.mesh.coords, part.mesh.cells, pdm.p2_dofmap, pdm.p1_dofmap, pdm.n_p2, pdm.n_p1)
    g2 = part.p2_global(dm)
    assert np.abs(pdm.p2_coords - dm.p2_coords[g2]).max() < 1e-12
    assert np.abs(pdm.p1_coords - dm.p1_coords[part.p1_global]).max() < 1e-12
    ul = ug.reshape(-1, 2)[g2].ravel()
    pl = pg[part.p1_global]
    for op in ref:
        y = c.operator_apply(op, arg(op, ul, pl))
        if op in (nat.OP_MASS_P2, nat.OP_STIFF_P2):
            d = (y - ref[op][g2])[part.p2_owned]
        else:
            d = (y - ref[op][part.p1_global])[part.p1_owned]
        print("rank", part.rank, "op", op, "owned-row error", np.abs(d).max())
    c.close()

# ---- partitioned IPCS step with / without multigrid
M = gg.DFGBoundaryMarkers
H = 4.1
def bc(dmap, mk):
    inlet = np.unique(dmap.facet_p2_nodes(mk.facets_with_id(M.inlet.value)))
    walls = np.unique(np.concatenate([dmap.facet_p2_nodes(mk.facets_with_id(m.value)).ravel() for m in (M.bottom, M.top, M.cylinder)]))
    y = dmap.p2_coords[inlet, 1]
    prof = 6.0 * y * (H - y) / H ** 2
    dofs = np.concatenate([2 * inlet, 2 * inlet + 1, 2 * walls, 2 * walls + 1])
    vals = np.concatenate([prof, np.zeros(inlet.size + 2 * walls.size)])
    _, first = np.unique(dofs[::-1], return_index=True)
    keep = dofs.size - 1 - first
    return dofs[keep].astype(np.int32), vals[keep]

def run(ctx, dmap, mk, pre_m, pre_p, tag):
    ctx.set_coeffs(1.0, 1.0, 0.05)
    ctx.set_dirichlet(nat.VELOCITY, *bc(dmap, mk))
    outlet = np.unique(dmap.facet_p1_nodes(mk.facets_with_id(M.outlet.value))).astype(np.int32)
    ctx.set_dirichlet(nat.PRESSURE, outlet, np.zeros(outlet.size))
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = 1e-10
        o.max_iter = 5000
    opts.momentum.precond, opts.poisson.precond = pre_m, pre_p
    opts.allow_nonconvergence = 1
    ctx.set_bdf((1.0, -1.0, 0.0), 0.05)
    try:
        info = ctx.step_ipcs(opts)
        print(tag, "newton", info.newton_iterations, "mom", info.krylov_iterations_momentum, "poi", info.krylov_iterations_poisson, flush=True)
    except Exception as e:
        print(tag, "FAILED", e, flush=True)
        os._exit(1)

from multigrid import attach_hierarchy
for pre_m, pre_p in ((0, 0), (1, 0), (0, 1)):
    c0 = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1)
    attach_hierarchy(c0, mesh)
    run(c0, dm, marks, pre_m, pre_p, "single %d%d" % (pre_m, pre_p))
    c0.close()
    group = nat.local_group_create(size)
    ctxs = []
    for r, part in enumerate(parts):
        pdm = part.dofmap
        c = nat.NsfemContext(part.mesh.coords, part.mesh.cells, pdm.p2_dofmap, pdm.p1_dofmap, pdm.n_p2, pdm.n_p1)
        c.attach_local_comm(group, r)
        ctxs.append(c)
    def worker(r):
        parts[r].attach(ctxs[r])
        run(ctxs[r], parts[r].dofmap, parts[r].markers, pre_m, pre_p, "rank %d %d%d" % (r, pre_m, pre_p))
    th = [threading.Thread(target=worker, args=(r,)) for r in range(size)]
    [t.start() for t in th]; [t.join() for t in th]
    for c in ctxs: c.close()
    nat.local_group_destroy(group)
