"""Stationary problem driver with Reynolds-number continuation (reference:
``StationaryProblem`` source/ns_problem.py:363-501).  Hook protocol as ``ProblemBase``; on a
failed nonlinear solve the Reynolds number is ramped up logarithmically from a solvable value
(:462-501)."""
import math

import numpy as np

import dlfn_compat as dlfn
from ns_problem import ProblemBase
from ns_solver_base import StationarySolverBase as StationarySolver


class StationaryProblem(ProblemBase):
    def __init__(self, main_dir=None, form_convective_term="standard", tol=1e-10, maxiter=50,
                 tol_picard=1e-2, maxiter_picard=10):
        super().__init__(main_dir)
        assert isinstance(form_convective_term, str)
        assert all(isinstance(i, int) and i > 0 for i in (maxiter, maxiter_picard))
        assert all(isinstance(i, float) and i > 0.0 for i in (tol_picard, tol))
        self._form_convective_term = form_convective_term
        self._tol_picard, self._maxiter_picard = tol_picard, maxiter_picard
        self._tol, self._maxiter = tol, maxiter

    def _get_solver(self):
        return self._navier_stokes_solver

    def solve_problem(self):
        self._call_hooks(("periodic", "constraints", "bcs", "force", "rotation", "coefficients"))
        if not hasattr(self, "_navier_stokes_solver"):
            self._navier_stokes_solver = StationarySolver(
                self._mesh, self._boundary_markers, self._form_convective_term, self._tol,
                self._maxiter, self._tol_picard, self._maxiter_picard)
        solver = self._navier_stokes_solver
        self._hand_over_to_solver(solver, ("coefficients", "force", "rotation", "periodic", "bcs"))
        try:
            Re = self._coefficient_handler.Re
            dlfn.info("Solving problem with Re = {0:.2f}".format(Re if Re is not None else float("nan")))
            solver.solve()
            self.postprocess_solution()
            self._write_xdmf_file()
            return
        except (RuntimeError, AssertionError) as err:
            # only a failed nonlinear / linear iteration sends the problem into the Reynolds-number
            # continuation; a missing device, a bad argument ... must stay loud
            import _native as nat
            if isinstance(err, nat.NativeError) and err.code not in (nat.ERR_NOT_CONVERGED, nat.ERR_BREAKDOWN):
                raise
        # Reynolds continuation: logarithmic ramp from Re = 10 to the target
        Re_final = self._coefficient_handler.Re
        assert Re_final is not None and Re_final > 10.0, "nonlinear solve failed"
        n_steps = int(math.ceil(math.log10(Re_final / 10.0) * 4.0)) + 1
        for Re in np.logspace(1.0, math.log10(Re_final), num=n_steps):
            self._coefficient_handler.modify_dimensionless_number("Re", float(Re))
            solver.set_equation_coefficients(self._coefficient_handler.equation_coefficients)
            dlfn.info("Solving problem with Re = {0:.2f}".format(Re))
            solver.solve()
        self.postprocess_solution()
        self._write_xdmf_file()
