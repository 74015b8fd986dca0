import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import test_reference_style_solvers as T
import ns_problem_stationary as NPS
p = T.StationaryChannelFlowProblem(3, "standard")
p.setup_mesh(); p.set_boundary_conditions(); p.set_equation_coefficients()
from ns_solver_base import StationarySolverBase
s = StationarySolverBase(p._mesh, p._boundary_markers, "standard")
s.set_equation_coefficients(p._coefficient_handler.equation_coefficients)
s.set_boundary_conditions(p._bcs)
s._setup_problem()
for picard, atol, maxit in ((True, 1e300, 1), (True, 1e-2, 10), (False, 1e-10, 50)):
    try:
        info = s._nonlinear_solve(picard, atol, maxit, True)
        print("picard", picard, "its", info.newton_iterations, "krylov", info.krylov_iterations_momentum, "conv", info.converged,
              ["%.2e" % info.newton_residuals[i] for i in range(info.newton_iterations + 1)])
    except Exception as e:
        print("EXC", repr(e))
for rtol in (1e-6, 1e-8, 1e-10, 1e-11):
    s.krylov_rtol = rtol; s.krylov_max_iter = 2000
    s._ctx.set_state(0, np.zeros(s._dofmap.n_velocity)); s._ctx.set_state(4, np.zeros(s._dofmap.n_p1))
    try:
        info = s._nonlinear_solve(True, 1e-2, 10, True)
        print("rtol", rtol, "its", info.newton_iterations, "krylov", info.krylov_iterations_momentum, ["%.2e" % info.newton_residuals[i] for i in range(info.newton_iterations + 1)])
    except Exception as e:
        print("rtol", rtol, "EXC", repr(e))
