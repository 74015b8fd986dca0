"""Scratch: monolithic BDF steps on the GPU vs the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import _native as nat
import fem_oracle as fo
from gpu_common import box, cavity_bc, context, rel, velocity_bc
from multigrid import attach_hierarchy

case = sys.argv[1]; n = int(sys.argv[2]); k = float(sys.argv[3]); nsteps = int(sys.argv[4])
use_oracle = n <= 64
if case == "cavity":
    mesh, dm, marks = box(n, n)
    vbc = cavity_bc(dm, marks); schur = np.zeros(0, np.int32)
else:
    mesh, dm, marks = box(8 * n, n, p1=(8.0, 1.0))
    zero = lambda X: np.zeros((X.shape[0], 2))
    inlet = lambda X: np.stack([6.0 * X[:, 1] * (1.0 - X[:, 1]), 0.0 * X[:, 1]], axis=1)
    vbc = velocity_bc(dm, marks, [(1, inlet), (3, zero), (4, zero)])
    schur = np.unique(dm.facet_p1_nodes(marks.facets_with_id(2))).astype(np.int32)
ctx = context(mesh, dm)
print("levels", attach_hierarchy(ctx, mesh, coarsest=4), "ndof", dm.n_dofs)
Re = 100.0
ctx.set_coeffs(1.0, 1.0, 1.0 / Re)
ctx.set_dirichlet(nat.VELOCITY, *vbc)
ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
ctx.set_dirichlet(nat.PRESSURE_PRECOND, schur, np.zeros(schur.size))
if use_oracle:
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=1.0 / Re, body_force_term=None)
    orc = fo.BDFOracle(s, coef, pin_pressure=(case == "cavity"))
opts = ctx.default_step_opts()
opts.momentum.rtol = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-12
opts.momentum.precond = 1
opts.momentum.max_iter = 400
for step in range(nsteps):
    alpha = fo.bdf_alpha(step, 1.0)
    ctx.set_bdf(alpha, k)
    t0 = time.time(); info = ctx.step_bdf(opts); tg = time.time() - t0
    print("step %d newton %d krylov %d  %.4fs res %s" % (step, info.newton_iterations, info.krylov_iterations_momentum, tg,
          ["%.2e" % info.newton_residuals[i] for i in range(info.newton_iterations + 1)]), flush=True)
    if use_oracle:
        orc.step(alpha, k, vbc)
        u, p = ctx.get_state(nat.U0), ctx.get_state(nat.P)
        uo, po = orc.sol[0][: dm.n_velocity], orc.sol[0][dm.n_velocity:]
        print("   oracle newton %d  rel err u %.2e  p(mod const) %.2e" % (orc.newton_its[-1], rel(u, uo), rel(p - p.mean(), po - po.mean())))
        orc.advance()
    ctx.advance(1)
