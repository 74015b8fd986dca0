"""Monolithic fully implicit BDF-2 Taylor-Hood step on the MI355X.

Same class name, constructor and hooks as the reference's
``source/ns_bdf_solver.py`` (:8-126): Newton on the mixed residual

  F = 1/k sum_i alpha_i (u_i, w) + c_c ((grad u) u, w) - c_p (p, div w)
      + c_v (grad u, grad w) - c_p (div u, q) [+ tractions - body force]   (:68-85)

with the exact Jacobian (:88) and dolfin's NewtonSolver control (:96-100).
"""
import _native as nat
from bdf_time_stepping import BDFTimeStepping
from ns_solver_base import InstationarySolverBase, WeakFormConvectiveTerm

_FORM_ID = {WeakFormConvectiveTerm.standard_form: 0, WeakFormConvectiveTerm.rotational_form: 1,
            WeakFormConvectiveTerm.divergence_form: 2, WeakFormConvectiveTerm.skew_symmetric_form: 3}


class ImplicitBDFSolver(InstationarySolverBase):
    # NB: the reference writes ("_solver") -- a string, so its _setup_problem re-runs
    # every step; a real tuple is used here.
    _required_objects = ("_solver",)
    _scheme_id = 1

    def __init__(self, mesh, boundary_markers, form_convective_term, time_stepping, tol=1e-10,
                 max_iter=50, device=0):
        assert isinstance(time_stepping, BDFTimeStepping)
        super().__init__(mesh, boundary_markers, form_convective_term, time_stepping, tol,
                         max_iter, device=device)
        self.last_step_info = None

    def _setup_problem(self):
        if not all(hasattr(self, attr) for attr in ("_Wh", "_solutions")):  # pragma: no cover
            self._setup_function_spaces()
        self._ctx.set_convective_form(_FORM_ID[self._form_convective_term])
        if not all(hasattr(self, attr) for attr in ("_next_step_size", "_alpha")):
            self._update_time_stepping_coefficients()
        self._setup_boundary_conditions()
        self._push_schur_dirichlet_set()
        self._solver = nat.SYS_MONOLITHIC

    def _push_schur_dirichlet_set(self):
        """Dirichlet set of the pressure Laplacian inside the Schur-complement preconditioner:
        P1 nodes on boundary parts without a full velocity condition (open / traction /
        component-wise boundaries) plus the true pressure Dirichlet dofs.  Only the
        preconditioner sees it; the discrete system is the reference's."""
        import numpy as np
        from ns_solver_base import VelocityBCType
        full = {bc[1] for bc in getattr(self, "_velocity_bcs", [])
                if bc[0] in (VelocityBCType.no_slip, VelocityBCType.constant, VelocityBCType.function)}
        marks, mesh, dm = self._boundary_markers, self._mesh, self._dofmap
        closed = list(full) + list(getattr(self, "_constrained_boundary_ids", ()))   # periodic parts
        open_facets = np.nonzero(mesh.facet_on_boundary & ~np.isin(marks.values, closed))[0]
        nodes = np.unique(dm.facet_p1_nodes(open_facets)) if open_facets.size else np.zeros(0, np.int64)
        if open_facets.size and getattr(self, "_mg_levels", None) is not None:
            # open boundaries: the geometric Laplacian with a strong Dirichlet condition leaves one
            # badly preconditioned mode per outflow node (iteration counts grow with the mesh);
            # the algebraic Laplacian D_f diag(M)^-1 D_f^T carries the right boundary behaviour
            from multigrid import attach_schur_laplacian
            nodes = np.asarray(self._dirichlet_bcs["pressure"][0], dtype=np.int32)
            self._ctx.set_dirichlet(nat.PRESSURE_PRECOND, nodes, np.zeros(nodes.size))
            attach_schur_laplacian(self._ctx, self._dirichlet_bcs["velocity"][0])
            return
        nodes = np.union1d(nodes, self._dirichlet_bcs["pressure"][0]).astype(np.int32)
        self._ctx.set_dirichlet(nat.PRESSURE_PRECOND, nodes, np.zeros(nodes.size))

    def _step_options(self):
        o = self._common_step_options(self._ctx.default_step_opts())
        o.convective_form = _FORM_ID[self._form_convective_term]
        o.momentum.rtol = self.krylov_rtol
        o.momentum.max_iter = self.krylov_max_iter
        o.momentum.precond = 1
        return o

    def _solve_time_step(self):
        try:
            self.last_step_info = self._ctx.step_bdf(self._step_options())
        except nat.NativeError as err:
            raise RuntimeError(str(err))
