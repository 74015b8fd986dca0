"""Scratch: stationary problems beyond the reach of the stationary Krylov solve (rotating Couette
flow Re = 1000 of tests/test_stationary_rotating_flow.py, lid-driven cavity Re = 1000) through the
solver classes, which fall back to pseudo-transient continuation."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
os.chdir("/tmp")
import numpy as np
import dlfn_compat as dlfn
dlfn.set_log_level(20)
from problem_specs import build_problem
from test_solver_classes_gpu import CASES

what, n, Re = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
t0 = time.time()
if what == "couette":
    problem = build_problem(CASES["rotating_couette"](n=n, Re=Re))
elif what == "step":
    problem = build_problem(CASES["backward_step"]())
else:
    problem = build_problem(CASES["stationary_cavity"](n=n, Re=Re))
problem.solve_problem()
solver = problem._get_solver()
i = solver.newton_info
print("%s n %d Re %g: newton its %d final |F| %.3e pseudo steps %s shift %s  %.1fs" % (
    what, n, Re, i.newton_iterations, i.newton_residuals[i.newton_iterations],
    getattr(solver, "pseudo_time_steps", 0), getattr(solver, "_preconditioner_shift", 0.0), time.time() - t0))
h = getattr(solver, "pseudo_time_history", [])
for k in range(0, len(h), max(1, len(h) // 40)):
    print("  step %4d tau %.3g |F| %.3e krylov %d" % (k, *h[k]))
