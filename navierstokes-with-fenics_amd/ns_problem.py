"""Problem drivers: the caller of the hot path.

Keeps the hook protocol and the transient time loop of the reference's
``source/ns_problem.py`` (``ProblemBase`` :17-360, ``InstationaryProblem``
:504-736; loop order :711-735) so that problem subclasses written for the reference
(``setup_mesh``, ``set_boundary_conditions``, ``set_equation_coefficients``,
``set_initial_conditions``, ``set_body_force``, ``postprocess_solution`` ...) drive
the device solvers unchanged.

Field output goes through ``xdmf_io.XDMFFile`` (XDMF 3 with raw binary or inline heavy data;
HDF5 is not available here); vorticity and pressure gradient are evaluated exactly on the
host copy of the fields at post-processing steps; the CFL number, which the reference
computes every step through a DG LocalSolver (:554-603), is one device kernel.
"""
import math
import os
import time

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

    # -- the protocol between a problem and its solver, as two tables ----------------------------
    # Both drivers of the reference (source/ns_problem.py:623-700 instationary, :394-431 stationary) first call the
    # user's hooks and then hand what the hooks left behind to the solver; only the ORDER differs between the two,
    # so the order is the argument and the steps live here once.
    _HOOKS = {"periodic": "set_periodic_boundary_conditions", "constraints": "set_internal_constraints",
              "rotation": "set_angular_velocity", "bcs": "set_boundary_conditions", "force": "set_body_force",
              "coefficients": "set_equation_coefficients", "initial": "set_initial_conditions"}

    def _call_hooks(self, order):
        """mesh, then the hooks named in ``order``; afterwards the consistency checks of the reference"""
        self.setup_mesh()
        assert self._mesh is not None
        self._space_dim = self._mesh.geometry().dim()
        self._n_cells = self._mesh.num_cells()
        for key in order:
            getattr(self, self._HOOKS[key])()
            if key == "coefficients":
                assert isinstance(getattr(self, "_coefficient_handler", None), EquationCoefficientHandler)
                self._coefficient_handler.close()
        if not hasattr(self, "_bcs"):
            assert hasattr(self, "_periodic_bcs")
        if hasattr(self, "_internal_constraints"):
            assert hasattr(self, "_bcs")
        if "initial" in order:
            assert hasattr(self, "_initial_conditions")

    def _hand_over_to_solver(self, solver, order):
        """solver.set_*(...) for everything the hooks have set, in ``order``"""
        for key in order:
            if key == "coefficients":
                solver.set_equation_coefficients(self._coefficient_handler.equation_coefficients)
            elif key == "force" and hasattr(self, "_body_force"):
                solver.set_body_force(self._body_force)
            elif key == "periodic" and hasattr(self, "_periodic_bcs"):
                assert hasattr(self, "_periodic_boundary_ids")
                solver.set_periodic_boundary_conditions(self._periodic_bcs, self._periodic_boundary_ids)
            elif key == "rotation" and hasattr(self, "_angular_velocity"):
                solver.set_angular_velocity(self._angular_velocity)
            elif key == "bcs" and hasattr(self, "_bcs"):
                constraints = (self._internal_constraints,) if hasattr(self, "_internal_constraints") else ()
                solver.set_boundary_conditions(self._bcs, *constraints)
            elif key == "initial":
                solver.set_initial_conditions(self._initial_conditions)

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

    # -- output protocol (reference source/ns_problem.py:31-53, 55-103, 217-264) -------------
    def _add_to_field_output(self, field):
        if not hasattr(self, "_additional_field_output"):
            self._additional_field_output = []
        self._additional_field_output.append(field)

    def _get_filename(self):
        assert hasattr(self, "_coefficient_handler")
        name = getattr(self, "_problem_name", type(self).__name__)
        fname = name + self._coefficient_handler.get_file_suffix() + self._suffix
        return os.path.join(self._results_dir, fname)

    def _create_xdmf_file(self):
        from xdmf_io import XDMFFile
        fname = self._get_filename()
        assert fname.endswith(".xdmf")
        os.makedirs(self._results_dir, exist_ok=True)
        self._xdmf_file = XDMFFile(fname)
        self._xdmf_file.parameters["flush_output"] = True
        self._xdmf_file.parameters["functions_share_mesh"] = True
        self._xdmf_file.parameters["rewrite_function_mesh"] = False

    def _get_boundary_conditions_map(self, field="velocity"):
        """{bc type: (boundary ids, ...)} of the problem's boundary conditions (reference :178-203)"""
        assert isinstance(field, str) and hasattr(self, "_bcs")
        BCType = VelocityBCType if field == "velocity" else PressureBCType
        assert field == "velocity" or field.lower() == "pressure"
        bc_map = {}
        for bc in self._bcs:
            if isinstance(bc[0], BCType):
                bc_map[bc[0]] = tuple(sorted(set(bc_map.get(bc[0], ())) | {bc[1]}))
        return bc_map

    def _compute_boundary_force(self, boundary_id, symmetric_gradient_factor=0.5):
        """Surface force  int ( -p n + c_v f (grad u + grad u^T) n ) dS  over the boundary part
        ``boundary_id`` (n = outward normal of the fluid domain, c_v = viscous coefficient, f =
        ``symmetric_gradient_factor``: 1/2 reproduces the traction of the reference's
        demo/dfg_benchmark.py:54-61, 1 the Newtonian stress) -- one device kernel
        (nsfem_boundary_force) instead of a dolfin ``assemble(... * ds(id))``.  New helper; the
        reference's callers spell this out in UFL, which ``dlfn.assemble`` evaluates as well."""
        import _native as nat
        solver = self._get_solver()
        facets = self._boundary_markers.facets_with_id(boundary_id)
        facets = facets[self._mesh.facet_on_boundary[facets]]
        cells, local = self._mesh.facet_cell_local(facets)
        nu = float(solver._equation_coefficients["viscous_term"]) * float(symmetric_gradient_factor)
        force, _, _ = solver._ctx.boundary_force(cells, local, nu, 1.0, nat.U0, nat.P)
        return tuple(float(v) for v in force)

    def _compute_stream_potential(self):
        """Velocity potential phi (the reference's "stream potential", :105-176): P1 solution of
        (grad phi, grad psi) = (div u, psi) - sum over the remaining boundaries of (n . u, psi),
        phi = 0 where the velocity obeys no-slip; boundaries with a no-normal-flux condition carry
        neither term.  The Poisson solve runs on the device (nsfem_poisson_solve); the right-hand
        side is assembled on the host from the exported divergence operator and exact 2-point Gauss
        integration of the cubic boundary integrand.  2D."""
        import _native as nat
        from fem_function import HostField
        solver = self._get_solver()
        dm, mesh, marks = solver._dofmap, self._mesh, self._boundary_markers
        assert dm.dim == 2, "the velocity potential post-processing is built for 2D meshes"
        bc_map = self._get_boundary_conditions_map()
        assert VelocityBCType.no_slip in bc_map
        other = set(marks.ids(boundary_only=True)) - set(bc_map[VelocityBCType.no_slip])
        other -= set(bc_map.get(VelocityBCType.no_normal_flux, ()))
        u = self._get_velocity().vector()
        rhs = solver._ctx.operator_apply(nat.OP_DIV, u)
        un = u.reshape(-1, 2)
        g = 0.5 / np.sqrt(3.0)
        for t in (0.5 - g, 0.5 + g):                                      # Gauss points on [0, 1]
            for bid in sorted(other):
                facets = marks.facets_with_id(bid)
                facets = facets[mesh.facet_on_boundary[facets]]
                if facets.size == 0:
                    continue
                nodes2 = dm.facet_p2_nodes(facets)                        # (end, end, midpoint)
                nodes1 = dm.facet_p1_nodes(facets)
                xf = mesh.coords[mesh.facets[facets].astype(np.int64)]
                length = np.linalg.norm(xf[:, 1] - xf[:, 0], axis=1)
                normal = mesh.facet_normals(facets)
                shape2 = np.array([(1 - t) * (1 - 2 * t), t * (2 * t - 1), 4 * t * (1 - t)])
                uq = np.einsum("k,fka->fa", shape2, un[nodes2])
                flux = (uq * normal).sum(axis=1) * 0.5 * length           # weight 1/2 per Gauss point
                np.subtract.at(rhs, nodes1[:, 0], flux * (1 - t))
                np.subtract.at(rhs, nodes1[:, 1], flux * t)
        dirichlet = np.unique(np.concatenate(
            [dm.facet_p1_nodes(marks.facets_with_id(i)).ravel() for i in bc_map[VelocityBCType.no_slip]]))
        phi = solver._ctx.poisson_solve(rhs, dirichlet)
        return HostField(mesh, "velocity potential", "Node", phi[dm.p1_vertex_node])

    def write_boundary_markers(self):
        """Facet markers as a file ParaView opens (reference: :329-348 writes a .pvd through
        dolfin.File): the marked facets as a Polyline (2D) / Triangle (3D) mesh with the marker id
        as cell attribute, results/<problem>_BoundaryMarkers.xdmf."""
        assert hasattr(self, "_boundary_markers") and hasattr(self, "_mesh")
        os.makedirs(self._results_dir, exist_ok=True)
        name = getattr(self, "_problem_name", type(self).__name__) + "_BoundaryMarkers.xdmf"
        mesh, marks = self._mesh, self._boundary_markers
        keep = np.nonzero(mesh.facet_on_boundary | (marks.values != 0))[0]
        facets = mesh.facets[keep].astype(np.int32)
        dim = mesh.coords.shape[1]
        kind = 'TopologyType="Polyline" NodesPerElement="2"' if dim == 2 else 'TopologyType="Triangle"'
        rows = lambda a, fmt: "\n".join(" ".join(fmt % v for v in r) for r in np.atleast_2d(a))
        with open(os.path.join(self._results_dir, name), "w") as fh:
            fh.write('<?xml version="1.0"?>\n<Xdmf Version="3.0"><Domain><Grid Name="boundary_markers" '
                     'GridType="Uniform">\n<Topology %s NumberOfElements="%d">\n<DataItem Dimensions="%d %d" '
                     'NumberType="Int" Format="XML">\n%s\n</DataItem></Topology>\n'
                     % (kind, facets.shape[0], facets.shape[0], facets.shape[1], rows(facets, "%d")))
            fh.write('<Geometry GeometryType="%s"><DataItem Dimensions="%d %d" NumberType="Float" Precision="8" '
                     'Format="XML">\n%s\n</DataItem></Geometry>\n'
                     % ("XY" if dim == 2 else "XYZ", mesh.coords.shape[0], dim, rows(mesh.coords, "%.17g")))
            fh.write('<Attribute Name="boundary_markers" AttributeType="Scalar" Center="Cell"><DataItem '
                     'Dimensions="%d" NumberType="Int" Format="XML">\n%s\n</DataItem></Attribute>\n'
                     '</Grid></Domain></Xdmf>\n' % (keep.size, " ".join(str(int(v)) for v in marks.values[keep])))
        return os.path.join(self._results_dir, name)

    def _cell_gradients(self, nodal, dofmap, p2):
        """physical gradients of a P1 / P2 field at the vertices of every cell:
        [nc, dim + 1 vertices, components, dim]"""
        mesh = self._mesh
        dim = mesh._dim
        x = mesh.coords[mesh.cells.astype(np.int64)]
        J = np.stack([x[:, k + 1] - x[:, 0] for k in range(dim)], axis=2)
        JinvT = np.transpose(np.linalg.inv(J), (0, 2, 1))
        dl = np.concatenate([-np.ones((1, dim)), np.eye(dim)], axis=0)        # grad lambda_i (reference)
        lam_at = np.eye(dim + 1)
        vals = nodal[np.asarray(dofmap, dtype=np.int64)]
        if vals.ndim == 2:
            vals = vals[:, :, None]
        if not p2:
            dphi = np.broadcast_to(dl, (dim + 1, dim + 1, dim))
        else:
            pairs = ((1, 2), (0, 2), (0, 1)) if dim == 2 else ((2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1))
            dphi = np.zeros((dim + 1, dim + 1 + len(pairs), dim))
            for v in range(dim + 1):
                l = lam_at[v]
                for i in range(dim + 1):
                    dphi[v, i] = (4.0 * l[i] - 1.0) * dl[i]
                for e, (a, b) in enumerate(pairs):
                    dphi[v, dim + 1 + e] = 4.0 * (l[a] * dl[b] + l[b] * dl[a])
        ref = np.einsum("vkd,cka->cvad", dphi, vals)               # reference gradients
        return np.einsum("cxd,cvad->cvax", JinvT, ref)

    def _compute_vorticity(self):
        """curl of the P2 velocity (a scalar in 2D, a vector in 3D); it lies in DG1, so the
        reference's L2 projection onto DG1 (:55-83) reproduces it exactly.  Stored per cell as the
        mean of its vertex values (``vertex_values`` keeps the DG1 data)."""
        from fem_function import HostField
        dm = self._get_solver()._dofmap
        g = self._cell_gradients(self._get_velocity().nodal_values(), dm.p2_dofmap, True)    # d_x u_a
        if dm.dim == 2:
            curl = g[:, :, 1, 0] - g[:, :, 0, 1]
        else:
            curl = np.stack([g[:, :, 2, 1] - g[:, :, 1, 2], g[:, :, 0, 2] - g[:, :, 2, 0],
                             g[:, :, 1, 0] - g[:, :, 0, 1]], axis=2)
        field = HostField(self._mesh, "vorticity", "Cell", curl.mean(axis=1))
        field.vertex_values = curl
        return field

    def _compute_pressure_gradient(self):
        """grad of the P1 pressure: piecewise constant = its DG0 projection (:85-103)."""
        from fem_function import HostField
        dm = self._get_solver()._dofmap
        g = self._cell_gradients(self._get_pressure().nodal_values(), dm.p1_dofmap, False)
        return HostField(self._mesh, "pressure gradient", "Cell", g[:, 0, 0, :])

    def _write_xdmf_file(self, current_time=0.0):
        """velocity, pressure and the additional fields at ``current_time`` (:244-264)."""
        assert isinstance(current_time, float)
        if os.environ.get("NSFEM_NO_OUTPUT"):
            return
        if not hasattr(self, "_xdmf_file"):
            self._create_xdmf_file()
        solver = self._get_solver()
        components = solver.solution.split()
        for index, name in solver.sub_space_association.items():
            components[index].rename(name, "")
            self._xdmf_file.write(components[index], current_time)
        if hasattr(self, "_additional_field_output"):
            for field in self._additional_field_output:
                if field is not None:
                    self._xdmf_file.write(field, current_time)
            self._additional_field_output.clear()
        self._last_output_time = current_time


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
        self._call_hooks(("periodic", "constraints", "rotation", "bcs", "force", "coefficients", "initial"))
        self._time_stepping = BDFTimeStepping(self._start_time, self._end_time,
                                              desired_start_time_step=self._desired_start_time_step)
        if not hasattr(self, "_navier_stokes_solver"):
            self._navier_stokes_solver = self._InstationarySolverClass(
                self._mesh, self._boundary_markers, self._form_convective_term,
                self._time_stepping, self._tol, self._maxiter)
        solver = self._navier_stokes_solver
        if hasattr(self, "solver_matrix_free"):      # Jacobian mode of the device step drivers
            solver.matrix_free = self.solver_matrix_free
        # settings of the device solves (no counterpart in the reference, which calls sparse LU):
        # problem.solver_settings = {"krylov_rtol": 1e-8, "newton_forcing": 1e-4, "pressure_start":
        # "extrapolated", "mass_solver": "chebyshev", "mg_truncation": (4, 0.1)}  -- or the string
        # "throughput" for exactly the set bench.py times (InstationarySolverBase.throughput_settings)
        settings = getattr(self, "solver_settings", None)
        if settings == "throughput":
            solver.throughput_settings()
        elif settings:
            for key, value in dict(settings).items():
                assert hasattr(solver, key), "unknown solver setting %r" % (key,)
                setattr(solver, key, value)
        self._hand_over_to_solver(solver, ("coefficients", "force", "periodic", "rotation", "bcs", "initial"))
        self._write_xdmf_file(current_time=0.0)
        print("Solving problem until time = {:0.2f}".format(self._time_stepping.end_time))

        assert hasattr(self, "_postprocessing_frequency")
        assert hasattr(self, "_output_frequency")
        ts = self._time_stepping
        self.step_wall_times = []             # seconds per solver.solve() (diagnostic)
        while not ts.is_at_end() and ts.step_number < self._n_max_steps:
            self._set_next_step_size()
            ts.update_coefficients()
            print(ts)
            t_wall = time.perf_counter()
            solver.solve()
            self.step_wall_times.append(time.perf_counter() - t_wall)
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


# the reference keeps StationaryProblem in this module (source/ns_problem.py:363-501)
from ns_problem_stationary import StationaryProblem  # noqa: E402,F401
