"""Scratch: periodic Taylor-Green vortex (2D / 3D) through the solver classes -- Krylov iteration
counts per step with the periodic multigrid hierarchy.
usage: gpu_periodic_tg.py DIM N STEPS [mg|twolevel] [bdf|ipcs] [DT]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
os.chdir("/tmp")
from problem_specs import build_problem
from test_solver_classes_gpu import CASES
dim, n, nsteps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
scheme = sys.argv[5] if len(sys.argv) > 5 else "bdf"
dt = float(sys.argv[6]) if len(sys.argv) > 6 else 0.1
if dim == 2:
    spec = CASES["taylor_green"](n=n, dt=dt, steps=nsteps, t1=max(1.0, dt * nsteps), scheme=scheme)
else:
    spec = CASES["taylor_green_3d"](n=n, dt=dt, steps=nsteps, scheme=scheme)
prob = build_problem(spec)
if len(sys.argv) > 4 and sys.argv[4] == "twolevel":      # the old behaviour: P2 -> P1 only
    orig = prob.setup_mesh
    def setup():
        orig()
        prob._mesh.structured = None
    prob.setup_mesh = setup
t0 = time.time()
prob.solve_problem()
solver = prob._get_solver()
i = solver.last_step_info
print("dim %d n %d: levels %s dofs %d  last step info: newton %d kry %d poisson %d  total %.1fs  step times %s" % (
    dim, n, solver._mg_levels, solver._n_dofs, i.newton_iterations, i.krylov_iterations_momentum,
    i.krylov_iterations_poisson, time.time() - t0, getattr(prob, "step_wall_times", None)))
