import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import _native as nat
import grid_generator as gg
from fem_mesh import TaylorHoodDofMap
from multigrid import attach_hierarchy, refinement_hierarchy, attach_schur_laplacian
m, nref, proj, scheme = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
coarse, cmarks = gg.dfg_channel(m, 0)
mesh, marks = refinement_hierarchy(coarse, cmarks, nref, project=gg._dfg_project if proj else None) if nref else (coarse, cmarks)
dm = TaylorHoodDofMap(mesh)
ctx = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1)
print("cells", mesh.num_cells(), "dofs", dm.n_dofs, "levels", attach_hierarchy(ctx, mesh))
last = {}
for mid in (1, 3, 4, 5):
    nodes = np.unique(dm.facet_p2_nodes(marks.facets_with_id(mid)))
    y = dm.p2_coords[nodes, 1]
    ux = 6.0 * y / 4.1 * (1 - y / 4.1) if mid == 1 else np.zeros_like(y)
    for d, v in zip(2 * nodes, ux): last[int(d)] = v
    for d in 2 * nodes + 1: last[int(d)] = 0.0
bd = np.array(sorted(last), dtype=np.int32); bv = np.array([last[d] for d in bd])
ctx.set_coeffs(1.0, 1.0, 0.01)
ctx.set_dirichlet(nat.VELOCITY, bd, bv)
pn = np.unique(dm.facet_p1_nodes(marks.facets_with_id(2))).astype(np.int32)
if scheme == "ipcs":
    ctx.set_dirichlet(nat.PRESSURE, pn, np.zeros(pn.size))
else:
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    if scheme == "bdfalg":
        ctx.set_dirichlet(nat.PRESSURE_PRECOND, np.zeros(0, np.int32), np.zeros(0))
        t0 = time.time(); print("singular", attach_schur_laplacian(ctx, bd), "setup %.2fs" % (time.time() - t0))
    else:
        ctx.set_dirichlet(nat.PRESSURE_PRECOND, pn, np.zeros(pn.size))
o = ctx.default_step_opts()
for k in (o.momentum, o.poisson, o.correction): k.rtol = 1e-10; k.max_iter = 300
o.momentum.precond = o.poisson.precond = 1
for step in range(2):
    ctx.set_bdf((1.0, -1.0, 0.0) if step == 0 else (1.5, -2.0, 0.5), 0.005)
    t0 = time.time()
    try:
        info = ctx.step_ipcs(o) if scheme == "ipcs" else ctx.step_bdf(o)
        print("step", step, "newton", info.newton_iterations, "kry", info.krylov_iterations_momentum, info.krylov_iterations_poisson, info.krylov_iterations_correction, "%.3fs" % (time.time() - t0))
    except Exception as e:
        print("step", step, "EXC", e); break
    ctx.advance(0 if scheme == "ipcs" else 1)
