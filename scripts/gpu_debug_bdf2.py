import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import _native as nat
from gpu_common import box, velocity_bc, context
from multigrid import attach_hierarchy, attach_schur_laplacian
nx, ny, coarsest = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mesh, dm, marks = box(nx, ny, p1=(8.0, 1.0))
zero = lambda X: np.zeros((X.shape[0], 2))
inlet = lambda X: np.stack([6.0 * X[:, 1] * (1.0 - X[:, 1]), 0.0 * X[:, 1]], axis=1)
vbc = velocity_bc(dm, marks, [(1, inlet), (3, zero), (4, zero)])
schur = np.unique(dm.facet_p1_nodes(marks.facets_with_id(2))).astype(np.int32)
ctx = context(mesh, dm)
print("levels", attach_hierarchy(ctx, mesh, coarsest=coarsest), "dofs", dm.n_dofs)
ctx.set_coeffs(1.0, 1.0, 0.01)
ctx.set_dirichlet(nat.VELOCITY, *vbc)
ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
if len(sys.argv) > 4 and sys.argv[4] == "alg":
    ctx.set_dirichlet(nat.PRESSURE_PRECOND, np.zeros(0, np.int32), np.zeros(0))
    print("singular", attach_schur_laplacian(ctx, vbc[0]))
else:
    ctx.set_dirichlet(nat.PRESSURE_PRECOND, schur, np.zeros(schur.size))
o = ctx.default_step_opts(); o.momentum.rtol = 1e-10; o.momentum.precond = 1; o.momentum.max_iter = 200
ctx.set_bdf((1.0, -1.0, 0.0), 0.005)
try:
    info = ctx.step_bdf(o); print("newton", info.newton_iterations, "kry", info.krylov_iterations_momentum)
except Exception as e:
    print("EXC", e)
