"""Scratch: time IPCS steps on the cavity at size n (no oracle)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd")]
import numpy as np
import _native as nat
from fem_mesh import rectangle_mesh, TaylorHoodDofMap, FacetMarkers

n = int(sys.argv[1]); k = float(sys.argv[2]); nsteps = int(sys.argv[3])
rtol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-12
mg = int(sys.argv[5]) if len(sys.argv) > 5 else 0          # 0 none, 1 poisson, 2 poisson+momentum
degree = int(sys.argv[6]) if len(sys.argv) > 6 else 2
ratio = float(sys.argv[7]) if len(sys.argv) > 7 else 4.0
t0 = time.time()
m = rectangle_mesh((0, 0), (1, 1), n, n)
dm = TaylorHoodDofMap(m)
print("mesh+dofmap %.2fs" % (time.time() - t0)); t0 = time.time()
ctx = nat.NsfemContext(m.coords, m.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1)
print("create %.2fs  n_p2=%d n_p1=%d ndof=%d" % (time.time() - t0, dm.n_p2, dm.n_p1, dm.n_dofs))
marks = FacetMarkers(m)
near = lambda v: (lambda X: np.abs(X - v) < 1e-12)
marks.mark(lambda X: near(0.0)(X[:, 0]), 1); marks.mark(lambda X: near(1.0)(X[:, 0]), 2)
marks.mark(lambda X: near(0.0)(X[:, 1]), 3); marks.mark(lambda X: near(1.0)(X[:, 1]), 4)
dofs, vals = [], []
for mid, val in ((1, (0., 0.)), (2, (0., 0.)), (3, (0., 0.)), (4, (1., 0.))):
    nodes = np.unique(dm.facet_p2_nodes(marks.facets_with_id(mid)))
    for a in range(2):
        dofs.append(2 * nodes + a); vals.append(np.full(nodes.size, val[a]))
ctx.set_coeffs(1.0, 1.0, 1.0 / 100.0)
ctx.set_dirichlet(nat.VELOCITY, np.concatenate(dofs), np.concatenate(vals))
ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
if mg:
    from multigrid import attach_hierarchy
    t0 = time.time(); nl = attach_hierarchy(ctx, m, degree, ratio); print("mg levels %d setup %.2fs" % (nl, time.time() - t0))
o = ctx.default_step_opts()
if mg >= 1: o.poisson.precond = 1
if mg >= 2: o.momentum.precond = 1
for kk in (o.momentum, o.poisson, o.correction):
    kk.rtol = rtol
for step in range(nsteps):
    alpha = (1.0, -1.0, 0.0) if step == 0 else (1.5, -2.0, 0.5)
    ctx.set_bdf(alpha, k)
    t0 = time.time(); info = ctx.step_ipcs(o); tg = time.time() - t0
    print("step %d newton %d kry mom %d poi %d cor %d | %.4fs  res %s" % (
        step, info.newton_iterations, info.krylov_iterations_momentum,
        info.krylov_iterations_poisson, info.krylov_iterations_correction, tg,
        ["%.2e" % info.newton_residuals[i] for i in range(info.newton_iterations + 1)]), flush=True)
    ctx.advance(0)
for op, name in ((nat.OP_MOMENTUM_JAC, "J 2x2"), (nat.OP_MASS_P2, "M2 x2rhs"), (nat.OP_STIFF_P1, "Ap"),
                 (nat.OP_DIV, "div"), (nat.OP_GRAD, "grad")):
    ms, nb = ctx.time_spmv(op, 50)
    print("spmv %-8s %.4f ms, %.1f MB, %.1f GB/s" % (name, ms, nb / 1e6, nb / ms / 1e6))
