"""Incremental pressure-correction scheme (IPCS) on the MI355X.

Same class name, constructor and hook methods as the reference's
``source/ns_ipcs_solver.py`` (:9-247).  Per time step (reference ``_solve_time_step``
:198-208) three systems are solved on device-resident split fields:

  1. diffusion step (Newton, BiCGStab):
       1/k sum_i alpha_i (u_i, w) + c_c ((grad u*) u*, w) - c_p (p_old, div w)
       + c_v (grad u*, grad w) [+ tractions - body force] = 0            (:106-147)
  2. projection step (CG):  (grad p, grad q) = (grad p_old, grad q)
       - alpha_0/k (div u*, q)                                            (:149-171)
  3. velocity correction (CG):  (v, w) = (u*, w) - k/alpha_0 (grad(p - p_old), w)
                                                                          (:173-196)
The constant Poisson and mass matrices are integrated once on the device instead
of being re-assembled and re-factorised every step as in the reference.
"""
import _native as nat
from bdf_time_stepping import BDFTimeStepping
from fem_function import DeviceFunction, MixedFunction
from ns_solver_base import InstationarySolverBase, WeakFormConvectiveTerm

_FORM_ID = {WeakFormConvectiveTerm.standard_form: 0, WeakFormConvectiveTerm.rotational_form: 1,
            WeakFormConvectiveTerm.divergence_form: 2, WeakFormConvectiveTerm.skew_symmetric_form: 3}


class _DeviceSystem:
    """Handle of one linear/non-linear system living in the device context (takes the
    place of the reference's dolfin *VariationalSolver attributes)."""

    def __init__(self, solver, system):
        self._solver, self.system = solver, system

    def solve(self, **kw):
        self._solver._assemble_system(self.system)
        return self._solver._ctx.solve(self.system, **kw)


class IPCSSolver(InstationarySolverBase):
    _required_objects = ("_diffusion_solver", "_projection_solver", "_velocity_correction_solver")
    _scheme_id = 0

    def __init__(self, mesh, boundary_markers, form_convective_term, time_stepping, tol=1e-10,
                 max_iter=50, device=0):
        assert isinstance(time_stepping, BDFTimeStepping)
        super().__init__(mesh, boundary_markers, form_convective_term, time_stepping, tol,
                         max_iter, device=device)
        #: True: one fused C-ABI call per step; False: Newton driven from Python through
        #: the explicit ``_assemble_system`` seam (same arithmetic, same device kernels)
        self.fused_step = True
        self.last_step_info = None

    def _setup_function_spaces(self):
        if not hasattr(self, "_Wh"):
            super()._setup_function_spaces()
        slots = (nat.U0, nat.U1, nat.U2)
        self._velocities = []
        for i in range(self._time_stepping.n_levels() + 1):
            name = i * "old" + (i > 0) * "_" + "velocity"
            self._velocities.append(DeviceFunction(self, "velocity", slots[i], name))
        self._intermediate_velocity = DeviceFunction(self, "velocity", nat.USTAR,
                                                     "intermediate_velocity")
        self._pressure = DeviceFunction(self, "pressure", nat.P, "pressure")
        self._old_pressure = DeviceFunction(self, "pressure", nat.P_OLD, "old_pressure")

    def _setup_problem(self):
        if not all(hasattr(self, a) for a in ("_Wh", "_solutions", "_intermediate_velocity",
                                              "_velocities", "_pressure", "_old_pressure")):  # pragma: no cover
            self._setup_function_spaces()
        self._ctx.set_convective_form(_FORM_ID[self._form_convective_term])
        if not all(hasattr(self, a) for a in ("_next_step_size", "_alpha")):
            self._update_time_stepping_coefficients()
        self._setup_boundary_conditions()
        self._diffusion_solver = _DeviceSystem(self, nat.SYS_MOMENTUM)
        self._projection_solver = _DeviceSystem(self, nat.SYS_POISSON)
        self._velocity_correction_solver = _DeviceSystem(self, nat.SYS_CORRECTION)

    def _step_options(self):
        o = self._common_step_options(self._ctx.default_step_opts())
        o.convective_form = _FORM_ID[self._form_convective_term]
        for k in (o.momentum, o.poisson, o.correction):
            k.rtol = self.krylov_rtol
            k.max_iter = self.krylov_max_iter
        if self._mg_levels is not None:
            o.momentum.precond = o.poisson.precond = 1
        assert self.poisson_solver in ("multigrid", "fast_diagonalization")
        if self.poisson_solver == "fast_diagonalization" and self._fast_diagonalization_ready():
            o.poisson.precond = 3
        # velocity correction: Chebyshev iteration with a-priori element bounds (no dot products)
        # unless the Jacobi-CG is asked for
        assert self.mass_solver in ("chebyshev", "cg")
        o.correction.precond = 2 if self.mass_solver == "chebyshev" else 0
        return o

    def _solve_time_step(self):
        if self.fused_step:
            try:
                self.last_step_info = self._ctx.step_ipcs(self._step_options())
            except nat.NativeError as err:
                raise RuntimeError(str(err))
            return
        # ---- explicit path: dolfin NewtonSolver control, driven through the seam
        ctx, kw = self._ctx, dict(rtol=self.krylov_rtol, max_iter=self.krylov_max_iter)
        mg = dict(kw, precond=1 if self._mg_levels is not None else 0)
        self._assemble_system(nat.SYS_MOMENTUM, new_step=True)
        r0 = r = ctx.residual_norm(nat.SYS_MOMENTUM)
        residuals, it = [r], 0
        converged = r < self._tol
        while not converged and it < self._maxiter:
            ctx.solve(nat.SYS_MOMENTUM, **mg)          # J dx = b ; u* -= dx
            it += 1
            self._assemble_system(nat.SYS_MOMENTUM)
            r = ctx.residual_norm(nat.SYS_MOMENTUM)
            residuals.append(r)
            converged = (r / r0 < 10.0 * self._tol) or (r < self._tol)
        if not converged:
            raise RuntimeError("Newton solver did not converge")
        self.last_newton_residuals = residuals
        direct = self.poisson_solver == "fast_diagonalization" and self._fast_diagonalization_ready()
        self._projection_solver.solve(**(dict(kw, precond=3) if direct else mg))
        cheb = self.mass_solver == "chebyshev"
        self._velocity_correction_solver.solve(**dict(kw, precond=2 if cheb else 0))

    def set_initial_conditions(self, initial_conditions):
        super().set_initial_conditions(initial_conditions)
        assert all(hasattr(self, x) for x in ("_velocities", "_intermediate_velocity",
                                              "_pressure", "_old_pressure"))
        # split fields alias the mixed levels on the device (U0/U1, P/P_OLD): the
        # reference's four FunctionAssigner copies (:229-238) are no-ops here.

    @property
    def solution(self):
        """(velocity, pressure) at the new time level; no mixed<->split gather is
        needed because the device stores split fields (reference: :241-247)."""
        return MixedFunction(self, nat.U0, nat.P, name="solution")
