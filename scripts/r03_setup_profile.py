"""host set-up of the 3D channel workload (configs[4]) under cProfile: where do the seconds go?
usage: python scripts/r03_setup_profile.py [n]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd")]
import numpy as np
import _native as nat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
from partition import SlabPartition
from multigrid import attach_schur_laplacian
marks = {}
def lap(name, t0):
    marks[name] = time.perf_counter() - t0
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter(); part = SlabPartition((0.0, 0.0, 0.0), (2.0, 1.0, 1.0), 2 * n, n, n, 0, 1, coarsest=4); lap("SlabPartition", t)
mesh, dm = part.mesh, part.dofmap
t = time.perf_counter(); ctx = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1, 0); lap("nsfem_create", t)
t = time.perf_counter(); levels = part.attach(ctx, None, None); lap("part.attach", t)
X = dm.p2_coords
near = lambda v, c: np.abs(v - c) < 1e-12
inlet = np.nonzero(near(X[:, 0], 0.0))[0]
walls = np.nonzero(near(X[:, 1], 0.0) | near(X[:, 1], 1.0) | near(X[:, 2], 0.0) | near(X[:, 2], 1.0))[0]
bd = np.concatenate([3 * inlet, 3 * inlet + 1, 3 * inlet + 2, 3 * walls, 3 * walls + 1, 3 * walls + 2]).astype(np.int32)
ctx.set_coeffs(1.0, 1.0, 1e-3)
t = time.perf_counter(); ctx.set_dirichlet(nat.VELOCITY, bd, np.zeros(bd.size)); ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0)); ctx.set_dirichlet(nat.PRESSURE_PRECOND, np.zeros(0, np.int32), np.zeros(0)); lap("set_dirichlet", t)
t = time.perf_counter(); attach_schur_laplacian(ctx, np.unique(bd), part=part); lap("schur_laplacian", t)
pr.disable()
print("n =", n, "dofs", dm.n_dofs, {k: round(v, 2) for k, v in marks.items()}, "total", round(sum(marks.values()), 2))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
