"""Scratch: periodic Taylor-Green vortex (2D / 3D) through the solver classes -- Krylov iteration
counts per step with the periodic multigrid hierarchy."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
os.chdir("/tmp")
import numpy as np
import test_reference_style_solvers as T
dim, n, nsteps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cls = T.TaylorGreenVortex if dim == 2 else T.TaylorGreenVortex3D
prob = cls.__new__(cls)
cls.__init__(prob)
prob._n_points = n
if hasattr(prob, "_n_max_steps"): prob._n_max_steps = nsteps
if len(sys.argv) > 4 and sys.argv[4] == "twolevel":      # the old behaviour: P2 -> P1 only
    orig = prob.setup_mesh
    def setup():
        orig()
        prob._mesh.structured = None
    prob.setup_mesh = setup
if len(sys.argv) > 5 and sys.argv[5] == "ipcs":
    prob.set_solver_class(T.IPCSSolver)
if len(sys.argv) > 6:
    prob._time_stepping_args = None
    dt = float(sys.argv[6])
    prob._start_time, prob._end_time = 0.0, dt * nsteps
    prob._desired_start_time_step = dt
t0 = time.time()
prob.solve_problem()
solver = prob._get_solver()
i = solver.last_step_info
print("dim %d n %d: levels %s dofs %d  last step info: newton %d kry %d poisson %d  total %.1fs  step times %s" % (
    dim, n, solver._mg_levels, solver._n_dofs, i.newton_iterations, i.krylov_iterations_momentum,
    i.krylov_iterations_poisson, time.time() - t0, getattr(prob, "step_wall_times", None)))
