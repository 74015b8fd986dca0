"""Problem drivers: the caller of the hot path.

Keeps the hook protocol and the transient time loop of the reference's
``source/ns_problem.py`` (``ProblemBase`` :17-360, ``InstationaryProblem``
:504-736; loop order :711-735) so that problem subclasses written for the reference
(``setup_mesh``, ``set_boundary_conditions``, ``set_equation_coefficients``,
``set_initial_conditions``, ``set_body_force``, ``postprocess_solution`` ...) drive
the device solvers unchanged.

Out of scope here: XDMF/HDF5 output and the DG projections
behind ``_compute_vorticity`` / ``_compute_pressure_gradient`` -- the writers are
no-ops that keep the call protocol; the CFL number, which the reference computes
every step through a DG LocalSolver and then discards (:554-603), is evaluated as a
nodal estimate on the host copy of the velocity for diagnostics only.
"""
import math
import os

import numpy as np

import dlfn_compat as dlfn
from auxiliary_classes import EquationCoefficientHandler
from bdf_time_stepping import BDFTimeStepping
from ns_solver_base import InstationarySolverBase as InstationarySolver
from ns_solver_base import PressureBCType, TractionBCType, VelocityBCType  # noqa: F401


class ProblemBase:
    _suffix = ".xdmf"

    def __init__(self, main_dir=None):
        self._main_dir = os.getcwd() if main_dir is None else main_dir
        self._results_dir = os.path.join(self._main_dir, "results")

    # -- optional hooks (defaults: nothing to set) ---------------------------------
    def set_periodic_boundary_conditions(self):
        pass

    def set_internal_constraints(self):
        pass

    def set_angular_velocity(self):
        pass

    def set_body_force(self):
        pass

    def postprocess_solution(self):
        pass

    # -- mandatory hooks -------------------------------------------------------------
    def setup_mesh(self):  # pragma: no cover
        raise NotImplementedError("You are calling a purely virtual method.")

    def set_boundary_conditions(self):  # pragma: no cover
        raise NotImplementedError("You are calling a purely virtual method.")

    def set_equation_coefficients(self):  # pragma: no cover
        raise NotImplementedError("You are calling a purely virtual method.")

    # -- field access ------------------------------------------------------------------
    def _get_solver(self):  # pragma: no cover
        raise NotImplementedError("You are calling a purely virtual method.")

    def _get_velocity(self):
        return self._get_solver().solution.split()[0]

    def _get_pressure(self):
        return self._get_solver().solution.split()[1]

    @property
    def space_dim(self):
        return self._space_dim

    # -- output protocol (I/O formats are out of scope: recorded, not written) -----------
    def _add_to_field_output(self, field):
        if not hasattr(self, "_additional_field_output"):
            self._additional_field_output = []
        self._additional_field_output.append(field)

    def _compute_vorticity(self):
        return None

    def _compute_pressure_gradient(self):
        return None

    def _write_xdmf_file(self, current_time=0.0):
        self._last_output_time = current_time
        if hasattr(self, "_additional_field_output"):
            self._additional_field_output.clear()


class InstationaryProblem(ProblemBase):
    def __init__(self, main_dir=None, start_time=0.0, end_time=1.0, form_convective_term="standard",
                 desired_start_time_step=0.1, n_max_steps=1000, tol=1e-10, maxiter=50):
        super().__init__(main_dir)
        assert isinstance(form_convective_term, str)
        assert all(isinstance(i, int) and i > 0 for i in (maxiter, n_max_steps))
        assert all(isinstance(x, float) and x >= 0.0
                   for x in (start_time, end_time, desired_start_time_step))
        self._form_convective_term = form_convective_term
        self._start_time, self._end_time = start_time, end_time
        self._desired_start_time_step = desired_start_time_step
        self._n_max_steps = n_max_steps
        self._tol, self._maxiter = tol, maxiter
        self._adaptive_time_stepping = False
        self._p_deg = 1

    def set_initial_conditions(self):  # pragma: no cover
        raise NotImplementedError("You are calling a purely virtual method.")

    def set_solver_class(self, InstationarySolverClass):
        assert issubclass(InstationarySolverClass, InstationarySolver)
        self._InstationarySolverClass = InstationarySolverClass

    def _get_solver(self):
        return self._navier_stokes_solver

    def _compute_cfl_number(self, step_size):
        """Maximum local CFL number, evaluated on the device exactly as the reference defines it
        (source/ns_problem.py:554-587): cell-local L2 projection onto DG2 of  p |u| k / h
        (p = 2, h = CellDiameter) with the degree-4 rule, max-norm of the coefficients.  One
        kernel + a 2 KB read-back (C ABI nsfem_cfl_number)."""
        import _native as nat
        cfl = self._get_solver()._ctx.cfl_number(nat.U0, step_size)
        assert math.isfinite(cfl) and cfl >= 0.0
        dlfn.info("Current CFL number = {0:6.2e}".format(cfl))
        return cfl

    def _set_next_step_size(self):
        """source/ns_problem.py:589-603: the CFL number is computed every step; the step size is
        only ever changed through the (uncovered) adaptive branch, which the reference leaves
        switched off."""
        next_step_size = self._time_stepping.get_next_step_size()
        assert next_step_size > 0.0 and math.isfinite(next_step_size)
        if getattr(self, "compute_cfl", True):
            self._last_cfl = self._compute_cfl_number(next_step_size)

    def solve_problem(self):
        assert hasattr(self, "_InstationarySolverClass")
        self.setup_mesh()
        assert self._mesh is not None
        self._space_dim = self._mesh.geometry().dim()
        self._n_cells = self._mesh.num_cells()
        self.set_periodic_boundary_conditions()
        self.set_internal_constraints()
        self.set_angular_velocity()
        self.set_boundary_conditions()
        self.set_body_force()
        self.set_equation_coefficients()
        assert isinstance(getattr(self, "_coefficient_handler", None), EquationCoefficientHandler)
        self._coefficient_handler.close()
        self.set_initial_conditions()
        if not hasattr(self, "_bcs"):
            assert hasattr(self, "_periodic_bcs")
        if hasattr(self, "_internal_constraints"):
            assert hasattr(self, "_bcs")
        assert hasattr(self, "_initial_conditions")

        self._time_stepping = BDFTimeStepping(self._start_time, self._end_time,
                                              desired_start_time_step=self._desired_start_time_step)
        if not hasattr(self, "_navier_stokes_solver"):
            self._navier_stokes_solver = self._InstationarySolverClass(
                self._mesh, self._boundary_markers, self._form_convective_term,
                self._time_stepping, self._tol, self._maxiter)
        solver = self._navier_stokes_solver
        solver.set_equation_coefficients(self._coefficient_handler.equation_coefficients)
        if hasattr(self, "_body_force"):
            solver.set_body_force(self._body_force)
        if hasattr(self, "_periodic_bcs"):
            assert hasattr(self, "_periodic_boundary_ids")
            solver.set_periodic_boundary_conditions(self._periodic_bcs, self._periodic_boundary_ids)
        if hasattr(self, "_angular_velocity"):
            solver.set_angular_velocity(self._angular_velocity)
        if hasattr(self, "_bcs"):
            if hasattr(self, "_internal_constraints"):
                solver.set_boundary_conditions(self._bcs, self._internal_constraints)
            else:
                solver.set_boundary_conditions(self._bcs)
        solver.set_initial_conditions(self._initial_conditions)
        self._write_xdmf_file(current_time=0.0)
        print("Solving problem until time = {:0.2f}".format(self._time_stepping.end_time))

        assert hasattr(self, "_postprocessing_frequency")
        assert hasattr(self, "_output_frequency")
        ts = self._time_stepping
        while not ts.is_at_end() and ts.step_number < self._n_max_steps:
            self._set_next_step_size()
            ts.update_coefficients()
            print(ts)
            solver.solve()
            if self._postprocessing_frequency > 0 and \
                    ts.step_number % self._postprocessing_frequency == 0:
                self.postprocess_solution()
            ts.advance_time()
            solver.advance_time()
            if hasattr(self, "_angular_velocity"):
                # the frame's angular velocity is moved to the NEW current time only after the
                # step, i.e. step n -> n+1 uses omega(t_n) (reference :728-731)
                solver._angular_velocity.set_time(ts.current_time)
            if self._output_frequency > 0 and ts.step_number % self._output_frequency == 0:
                self._write_xdmf_file(current_time=ts.current_time)
        print(ts)
