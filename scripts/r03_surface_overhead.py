"""host overhead of one pass of the reference's time loop body through the solver classes (small mesh: the device
time is negligible) -- cProfile of 300 passes"""
import cProfile, contextlib, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd")]
from auxiliary_classes import EquationCoefficientHandler
from grid_generator import HyperCubeBoundaryMarkers, hyper_cube
from ns_ipcs_solver import IPCSSolver
from ns_problem import InstationaryProblem, VelocityBCType
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32

class CavityProblem(InstationaryProblem):
    def __init__(self):
        super().__init__(None, start_time=0.0, end_time=1e-3 * 2000, desired_start_time_step=1e-3, n_max_steps=5)
        self._problem_name = "Cavity"; self._output_frequency = 0; self._postprocessing_frequency = 0
        self.compute_cfl = False
        self.set_solver_class(IPCSSolver)
        self.solver_settings = "throughput"
    def setup_mesh(self):
        self._mesh, self._boundary_markers = hyper_cube(2, n)
    def set_boundary_conditions(self):
        m = HyperCubeBoundaryMarkers
        self._bcs = ((VelocityBCType.no_slip, m.left.value, None), (VelocityBCType.no_slip, m.right.value, None),
                     (VelocityBCType.no_slip, m.bottom.value, None), (VelocityBCType.constant, m.top.value, (1.0, 0.0)))
    def set_equation_coefficients(self):
        self._coefficient_handler = EquationCoefficientHandler(Re=100.0)
    def set_initial_conditions(self):
        self._initial_conditions = {"velocity": (0.0, 0.0), "pressure": 0.0}

log = io.StringIO()
with contextlib.redirect_stdout(log):
    problem = CavityProblem(); problem.solve_problem()
    solver, ts = problem._get_solver(), problem._time_stepping
    pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter()
    for _ in range(300):
        problem._set_next_step_size(); ts.update_coefficients(); solver.solve(); ts.advance_time(); solver.advance_time()
    solver._ctx.synchronize(); el = time.perf_counter() - t0; pr.disable()
print("ms per pass", 1e3 * el / 300)
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
