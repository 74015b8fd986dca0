"""Solver surface of the MI355X implementation.

Mirrors the classes and call order of the reference's ``source/ns_solver_base.py``
(``SolverBase`` :59-870 and ``InstationarySolverBase`` :991-1207): the same
constructor arguments, ``set_*`` methods with the same validity checks, the same
boundary-condition tuples and enums, ``solve()`` / ``advance_time()`` /
``solution`` / ``_solve_time_step()`` hooks -- but no UFL form is built.  The
weak forms of the reference (terms :121-191, :370-399, :662-673) are integrated by
the HIP element kernels behind the ctypes C ABI (``_native.py``,
``include/nsfem.h``); ``_assemble_system()`` is the new explicit seam to them
(SURVEY.md D1).  The device library is mandatory: no CPU fallback exists.
"""
import math
from enum import Enum, auto

import numpy as np

import _native as nat
import dlfn_compat as dlfn
import fem_host
from discrete_time import DiscreteTime
from fem_function import MixedFunction
from fem_mesh import FacetMarkers, Mesh, TaylorHoodDofMap, periodic_entity_map
from multigrid import attach_hierarchy


class VelocityBCType(Enum):
    no_slip = auto()
    no_normal_flux = auto()
    no_tangential_flux = auto()
    constant = auto()
    constant_component = auto()
    function = auto()
    function_component = auto()


class PressureBCType(Enum):
    constant = auto()
    function = auto()
    mean_value = auto()


class TractionBCType(Enum):
    constant = auto()
    constant_component = auto()
    function = auto()
    function_component = auto()
    free = auto()


class WeakFormConvectiveTerm(Enum):
    standard_form = auto()
    rotational_form = auto()
    divergence_form = auto()
    skew_symmetric_form = auto()


class WeakFormViscousTerm(Enum):
    reduced_form = auto()
    traction_form = auto()


_CONVECTIVE = {"standard": WeakFormConvectiveTerm.standard_form,
               "rotational": WeakFormConvectiveTerm.rotational_form,
               "divergence": WeakFormConvectiveTerm.divergence_form,
               "skew_symmetric": WeakFormConvectiveTerm.skew_symmetric_form}
_VISCOUS = {"standard": WeakFormViscousTerm.reduced_form,
            "reduced": WeakFormViscousTerm.reduced_form,
            "traction": WeakFormViscousTerm.traction_form}
_EXPRESSION_TYPES = (dlfn.Expression, dlfn.UserExpression)
_COEFFICIENT_KEYS = ("convective_term", "coriolis_term", "euler_term", "pressure_term",
                     "viscous_term", "body_force_term")


class SolverBase:
    """P2-P1 Taylor-Hood discretisation of the incompressible Navier-Stokes
    equations; state, operators and solves live on one MI355X."""

    _sub_space_association = {0: "velocity", 1: "pressure"}
    _field_association = {value: key for key, value in _sub_space_association.items()}

    def __init__(self, mesh, boundary_markers, form_convective_term="standard",
                 form_viscous_term="reduced", device=0):
        assert isinstance(mesh, Mesh)
        assert isinstance(boundary_markers, FacetMarkers)
        assert isinstance(form_convective_term, str) and form_convective_term.lower() in _CONVECTIVE
        assert isinstance(form_viscous_term, str) and form_viscous_term.lower() in _VISCOUS
        self._mesh = mesh
        self._boundary_markers = boundary_markers
        self._space_dim = mesh.geometry().dim()
        assert boundary_markers.dim() == self._space_dim - 1
        self._n_cells = mesh.num_cells()
        self._form_convective_term = _CONVECTIVE[form_convective_term.lower()]
        self._form_viscous_term = _VISCOUS[form_viscous_term.lower()]
        self._p_deg = 1
        self._device = device

    # ------------------------------------------------------------------ setters
    def set_equation_coefficients(self, input_coefficients):
        """Coefficients of the PDE terms; may be called again to update the values
        (reference: source/ns_solver_base.py:829-855)."""
        assert isinstance(input_coefficients, dict)
        assert all(key in _COEFFICIENT_KEYS for key in input_coefficients)
        if not hasattr(self, "_equation_coefficients"):
            self._equation_coefficients = {key: None for key in _COEFFICIENT_KEYS}
            for key, value in input_coefficients.items():
                if value is not None:
                    assert isinstance(value, float) and math.isfinite(value) and value > 0.0
                self._equation_coefficients[key] = value
        else:
            for key, value in self._equation_coefficients.items():
                assert key in input_coefficients
                if value is not None:
                    assert input_coefficients[key] is not None
                    self._equation_coefficients[key] = float(input_coefficients[key])
        self._push_coefficients()

    def set_body_force(self, body_force):
        assert isinstance(body_force, _EXPRESSION_TYPES + (dlfn.Constant,))
        assert body_force.value_rank() == 1
        if isinstance(body_force, dlfn.Constant):
            assert body_force.ufl_shape[0] == self._space_dim
        self._body_force = body_force
        self._body_force.rename("body_force", "")

    def set_angular_velocity(self, angular_velocity):
        """Rotating frame of reference (reference :680-691): Coriolis term
        2 c_cor omega (e_z x u, w) and Euler term c_e omega' (e_z x x, w), both integrated on the
        device (csrc/api.hip: coriolis_gamma, momentum_begin_step)."""
        from auxiliary_classes import AngularVelocityVector
        assert isinstance(angular_velocity, AngularVelocityVector)
        assert angular_velocity.space_dim == self._space_dim
        self._angular_velocity = angular_velocity

    def _push_angular_velocity(self):
        if hasattr(self, "_angular_velocity") and hasattr(self, "_ctx"):
            av = self._angular_velocity
            zero = 0.0 if av.space_dim == 2 else (0.0, 0.0, 0.0)
            self._ctx.set_angular_velocity(av.value, zero if av.derivative is None else av.derivative)

    def set_periodic_boundary_conditions(self, constrained_domain, constrained_boundary_ids):
        assert isinstance(constrained_domain, dlfn.SubDomain)
        assert isinstance(constrained_boundary_ids, (tuple, list))
        assert all(isinstance(i, int) for i in constrained_boundary_ids)
        self._constrained_domain = constrained_domain
        self._constrained_boundary_ids = constrained_boundary_ids

    def _check_boundary_condition_format(self, bc, internal_constraint=False):
        assert isinstance(bc, (list, tuple)) and len(bc) >= 2
        assert isinstance(bc[0], (VelocityBCType, PressureBCType, TractionBCType))
        rank = 0 if isinstance(bc[0], PressureBCType) else 1
        if bc[0] is not PressureBCType.mean_value:
            assert isinstance(bc[1], int)
            known = self._boundary_markers.ids(boundary_only=not internal_constraint)
            assert bc[1] in known, "Boundary id {0} was not found in the boundary markers.".format(bc[1])
        if rank == 0:
            assert isinstance(bc[2], _EXPRESSION_TYPES + (float,)) or bc[2] is None
            if isinstance(bc[2], _EXPRESSION_TYPES):
                assert bc[2].value_rank() == 0
        elif len(bc) == 3:
            assert isinstance(bc[2], _EXPRESSION_TYPES + (tuple, list)) or bc[2] is None
            if isinstance(bc[2], _EXPRESSION_TYPES):
                assert bc[2].value_rank() == 1
            elif isinstance(bc[2], (tuple, list)):
                assert len(bc[2]) == self._space_dim and all(isinstance(x, float) for x in bc[2])
        elif len(bc) == 4:
            assert isinstance(bc[2], int) and bc[2] < self._space_dim
            assert isinstance(bc[3], _EXPRESSION_TYPES + (float,)) or bc[3] is None
            if isinstance(bc[3], _EXPRESSION_TYPES):
                assert bc[3].value_rank() == 0

    def set_boundary_conditions(self, bcs, internal_constraints=None):
        """``bcs = [(Type, boundary_id, value), (Type, boundary_id, component, value)]``
        with the validity / conflict rules of source/ns_solver_base.py:722-827."""
        assert isinstance(bcs, (list, tuple))
        for bc in bcs:
            self._check_boundary_condition_format(bc)
        groups = {VelocityBCType: [], TractionBCType: [], PressureBCType: []}
        ids = {VelocityBCType: set(), TractionBCType: set(), PressureBCType: set()}
        for bc in bcs:
            if hasattr(self, "_constrained_domain"):
                assert bc[1] not in self._constrained_boundary_ids
            groups[type(bc[0])].append(bc)
            ids[type(bc[0])].add(bc[1])
        velocity_bcs, traction_bcs, pressure_bcs = (groups[VelocityBCType], groups[TractionBCType],
                                                    groups[PressureBCType])
        if not hasattr(self, "_constrained_domain"):
            assert len(velocity_bcs) > 0
        # velocity and traction on one boundary part: different components only
        component_velocity = (VelocityBCType.no_normal_flux, VelocityBCType.no_tangential_flux,
                              VelocityBCType.constant_component, VelocityBCType.function_component)
        component_traction = (TractionBCType.constant_component, TractionBCType.function_component)
        for bndry_id in ids[VelocityBCType] & ids[TractionBCType]:
            vbc = next(bc for bc in velocity_bcs if bc[1] == bndry_id)
            tbc = next(bc for bc in traction_bcs if bc[1] == bndry_id)
            assert vbc[0] in component_velocity and tbc[0] in component_traction
            assert tbc[2] != vbc[2]
        if internal_constraints is not None:
            assert isinstance(internal_constraints, (list, tuple))
            taken = ids[VelocityBCType] | ids[TractionBCType] | ids[PressureBCType]
            for bc in internal_constraints:
                self._check_boundary_condition_format(bc, True)
                assert bc[1] not in taken
                if isinstance(bc[0], VelocityBCType):
                    velocity_bcs.append(bc)
                elif isinstance(bc[0], PressureBCType):
                    pressure_bcs.append(bc)
                else:  # pragma: no cover
                    raise NotImplementedError()
        self._velocity_bcs = velocity_bcs
        if traction_bcs:
            self._traction_bcs = traction_bcs
            self._form_viscous_term = WeakFormViscousTerm.traction_form
        if pressure_bcs:
            self._pressure_bcs = pressure_bcs

    # ------------------------------------------------------------- device setup
    def _setup_function_spaces(self):
        """Taylor-Hood dof maps + device context (reference: :501-524)."""
        periodic = None
        if hasattr(self, "_constrained_domain"):
            # periodic constraint: slave entities share the dofs of their masters (dolfin
            # FunctionSpace(..., constrained_domain=...), reference :516-518)
            periodic = periodic_entity_map(self._mesh, self._constrained_domain)
        from fem_mesh import preferred_p2_order
        self._dofmap = TaylorHoodDofMap(self._mesh, reorder=preferred_p2_order(self._mesh.geometry().dim()), periodic_map=periodic)
        dm = self._dofmap
        self._ctx = nat.NsfemContext(self._mesh.coords, self._mesh.cells, dm.p2_dofmap,
                                     dm.p1_dofmap, dm.n_p2, dm.n_p1, device=self._device)
        self._n_dofs = dm.n_dofs
        from fem_spaces import FunctionSpace
        self._Wh = FunctionSpace(dm, "mixed")
        print("Number of cells {0}, number of DoFs: {1}".format(self._n_cells, self._n_dofs))
        # geometric multigrid hierarchy for the Krylov preconditioners (structured meshes
        # coarsen; any mesh gets at least the P2 -> P1 two-level hierarchy)
        self._mg_levels = None
        if getattr(self, "use_multigrid", True):
            # (periodic spaces on structured meshes: the coarse levels are periodic as well;
            # on other meshes only the two-level
            # P2 -> P1 hierarchy is built)
            if periodic and getattr(self._mesh, "structured", None) is not None:
                self._mg_levels = attach_hierarchy(
                    self._ctx, self._mesh, periodic=(self._constrained_domain, dm.p1_vertex_node))
            else:
                self._mg_levels = attach_hierarchy(self._ctx, None if periodic else self._mesh)
        self._push_coefficients()

    # sub-space access and mixed <-> split assignment (reference :213-300, :424-476) -----------
    def _get_subspaces(self):
        """{"velocity": ..., "pressure": ...}: the collapsed sub-spaces of ``_Wh`` (created once)."""
        assert hasattr(self, "_Wh")
        if not hasattr(self, "_WhSub"):
            self._WhSub = {key: self._Wh.sub(index).collapse()
                           for key, index in self._field_association.items()}
        return self._WhSub

    def _get_subspace(self, field):
        assert field in self._field_association
        return self._get_subspaces()[field]

    def _assign_function(self, receiving_functions, assigning_functions):
        """Copy between a function on the joint space (or one of its ``split()`` parts) and
        functions on the collapsed sub-spaces, in either direction; both arguments may be a
        function or a ``{field: function}`` dictionary, exactly the combinations the reference
        accepts (source/ns_solver_base.py:213-300)."""
        assert hasattr(self, "_Wh")
        spaces = self._get_subspaces()
        is_function = lambda f: hasattr(f, "function_space") and hasattr(f, "vector")
        assert is_function(receiving_functions) or isinstance(receiving_functions, dict)
        assert is_function(assigning_functions) or isinstance(assigning_functions, dict)

        def field_of(function):
            for key, space in spaces.items():
                if function in space:
                    return key
            return None

        def joint_part(function, key):
            """the storage of field ``key`` inside a function on Wh or on Wh.sub(index)"""
            index = self._field_association[key]
            if function in self._Wh:
                return function.sub(index)
            assert function in self._Wh.sub(index)
            return function

        if isinstance(receiving_functions, dict):
            forward, pairs = True, receiving_functions
        elif isinstance(assigning_functions, dict):
            forward, pairs = False, assigning_functions
        else:
            key = field_of(receiving_functions)
            forward = key is not None
            if forward:
                pairs = {key: receiving_functions}
            else:
                key = field_of(assigning_functions)
                assert key is not None
                pairs = {key: assigning_functions}
        assert 1 <= len(pairs) <= 2
        assert all(function in spaces[key] for key, function in pairs.items())
        joint = assigning_functions if forward else receiving_functions
        assert is_function(joint)
        if len(pairs) == 2:
            assert joint in self._Wh
        for key, function in pairs.items():
            part = joint_part(joint, key)
            if forward:
                function.assign(part)
            else:
                part.assign(function)

    def _push_coefficients(self):
        if hasattr(self, "_ctx") and hasattr(self, "_equation_coefficients"):
            c = self._equation_coefficients
            self._ctx.set_coeffs(c["convective_term"], c["pressure_term"], c["viscous_term"],
                                 c["body_force_term"], c["coriolis_term"], c["euler_term"])

    # Dirichlet sets -------------------------------------------------------------
    def _velocity_dirichlet_arrays(self):
        """(dofs, values) of all velocity conditions at the current expression time;
        list order is kept so that later conditions win at shared nodes."""
        dm = self._dofmap
        dofs, vals = [], []

        def add(nodes, comp, v):
            dofs.append(self._space_dim * nodes + comp)
            vals.append(np.broadcast_to(v, nodes.shape).astype(np.float64))

        for bc in getattr(self, "_velocity_bcs", []):
            bc_type, bndry_id = bc[0], bc[1]
            nodes = np.unique(dm.facet_p2_nodes(self._boundary_markers.facets_with_id(bndry_id)))
            X = dm.p2_coords[nodes]
            if bc_type is VelocityBCType.no_slip:
                for comp in range(self._space_dim):
                    add(nodes, comp, 0.0)
            elif bc_type in (VelocityBCType.no_normal_flux, VelocityBCType.no_tangential_flux):
                normal = np.array(fem_host.boundary_normal(self._mesh, self._boundary_markers, bndry_id))
                k = int(np.abs(normal).argmax())
                assert abs(abs(normal[k]) - 1.0) < 5.0e-14, "boundary must be axis aligned"
                comps = (k,) if bc_type is VelocityBCType.no_normal_flux else tuple(
                    d for d in range(self._space_dim) if d != k)
                for comp in comps:
                    add(nodes, comp, 0.0)
            elif bc_type is VelocityBCType.constant:
                assert isinstance(bc[2], (tuple, list))
                for comp in range(self._space_dim):
                    add(nodes, comp, bc[2][comp])
            elif bc_type is VelocityBCType.constant_component:
                assert isinstance(bc[3], float)
                add(nodes, bc[2], bc[3])
            elif bc_type is VelocityBCType.function:
                v = dlfn.evaluate(bc[2], X)
                for comp in range(self._space_dim):
                    add(nodes, comp, v[:, comp])
            elif bc_type is VelocityBCType.function_component:
                add(nodes, bc[2], dlfn.evaluate(bc[3], X))
            else:  # pragma: no cover
                raise RuntimeError()
        if not dofs:
            return np.zeros(0, dtype=np.int32), np.zeros(0)
        return np.concatenate(dofs).astype(np.int32), np.concatenate(vals)

    def _pressure_dirichlet_arrays(self):
        dm = self._dofmap
        dofs, vals = [], []
        for bc in getattr(self, "_pressure_bcs", []):
            assert len(bc) == 3
            bc_type, bndry_id, value = bc
            if bc_type is PressureBCType.mean_value:
                assert bndry_id is None and isinstance(value, float)
                self._mean_pressure_value = value
                continue
            nodes = np.unique(dm.facet_p1_nodes(self._boundary_markers.facets_with_id(bndry_id)))
            if bc_type is PressureBCType.constant:
                assert isinstance(value, float)
            v = dlfn.evaluate(value, dm.p1_coords[nodes])
            dofs.append(nodes)
            vals.append(v)
        if not dofs:
            return np.zeros(0, dtype=np.int32), np.zeros(0)
        return np.concatenate(dofs).astype(np.int32), np.concatenate(vals)

    def _setup_boundary_conditions(self):
        """Builds the Dirichlet dof sets and ships them to the device
        (reference: DirichletBC lists, :526-660)."""
        assert hasattr(self, "_ctx")
        vd, vv = self._velocity_dirichlet_arrays()
        pd, pv = self._pressure_dirichlet_arrays()
        if vd.size == 0 and pd.size == 0:
            assert hasattr(self, "_constrained_domain")
        self._dirichlet_bcs = {"velocity": (vd, vv), "pressure": (pd, pv)}
        self._ctx.set_dirichlet(nat.VELOCITY, vd, vv)
        self._ctx.set_dirichlet(nat.PRESSURE, pd, pv)
        self._fast_diag_for = None           # (factors belong to a Dirichlet set: rebuilt on demand)
        self._ctx.set_viscous_form(self._form_viscous_term is WeakFormViscousTerm.traction_form)
        self._push_natural_terms()

    def _fast_diagonalization_ready(self):
        """ships the factors of the direct projection-step solver for the current pressure Dirichlet set; False where
        it does not apply (no rectangle lattice, periodic or partitioned space, conditions on parts of a side)"""
        import poisson_fd
        pd = self._dirichlet_bcs["pressure"][0]
        key = pd.tobytes()
        if getattr(self, "_fast_diag_for", None) == key:
            return self._fast_diag_ok
        self._fast_diag_for, self._fast_diag_ok = key, False
        lines = poisson_fd.lattice_lines(self._mesh)
        if lines is None or hasattr(self, "_constrained_domain") or self._dofmap.n_p1 != lines[0].size * lines[1].size:
            return False
        f = poisson_fd.factors(lines[0], lines[1], pd)
        if f is None:
            return False
        self._ctx.poisson_set_fast_diag(f)
        self._fast_diag_ok = True
        return True

    def _push_natural_terms(self):
        """Body force (nodal P2 interpolant) and boundary tractions -> device vectors
        (reference: _add_body_forces :158-171, _add_boundary_tractions :121-156)."""
        dm = self._dofmap
        if hasattr(self, "_body_force"):
            assert self._equation_coefficients["body_force_term"] is not None
            f = dlfn.evaluate(self._body_force, dm.p2_coords)
            self._ctx.set_state(nat.BODY_FORCE, np.ascontiguousarray(f).ravel())
        if hasattr(self, "_traction_bcs"):
            total = np.zeros(dm.n_velocity)
            for bc in self._traction_bcs:
                bc_type, bndry_id = bc[0], bc[1]
                if bc_type is TractionBCType.free:
                    continue
                facets = self._boundary_markers.facets_with_id(bndry_id)

                def nodal(X, bc=bc, bc_type=bc_type):
                    out = np.zeros((X.shape[0], self._space_dim))
                    if bc_type in (TractionBCType.constant, TractionBCType.function):
                        out[:] = dlfn.evaluate(bc[2], X)
                    else:
                        out[:, bc[2]] = dlfn.evaluate(bc[3], X)
                    return out
                total += fem_host.traction_vector(dm, facets, nodal)
            self._ctx.set_state(nat.TRACTION, total)

    # ----------------------------------------------------------------- accessors
    @property
    def field_association(self):
        return self._field_association

    @property
    def sub_space_association(self):
        return self._sub_space_association

    @property
    def solution(self):
        return self._solution

    def solve(self):  # pragma: no cover
        raise NotImplementedError("You are calling a purely virtual method.")


class InstationarySolverBase(SolverBase):
    """Time-dependent solver base: owns the time levels and the per-step driver
    (reference: source/ns_solver_base.py:991-1207)."""

    _scheme_id = 0     # 0: IPCS slot rotation, 1: monolithic BDF

    def __init__(self, mesh, boundary_markers, form_convective_term, time_stepping, tol=1e-10,
                 max_iter=50, device=0):
        super().__init__(mesh, boundary_markers, form_convective_term, device=device)
        assert isinstance(max_iter, int) and max_iter > 0
        assert isinstance(tol, float) and tol > 0.0
        assert isinstance(time_stepping, DiscreteTime)
        assert hasattr(time_stepping, "n_levels")
        self._time_stepping = time_stepping
        self._tol = tol
        self._maxiter = max_iter
        # Krylov options of the device solves (the reference uses sparse LU instead)
        self.krylov_rtol = 1.0e-12
        self.krylov_max_iter = 20000
        # velocity Jacobian in the fused step drivers: None = library default (matrix-free element
        # kernel), False = assembled block CSR, True = matrix-free
        self.matrix_free = None
        #: geometric multigrid preconditioning of the momentum and Poisson solves
        self.use_multigrid = True
        # ---- solver settings without a counterpart in the reference (it calls sparse LU); the
        # defaults are the parity settings (direct-solver accuracy), `throughput_settings()` switches
        # to the ones bench.py times (SURVEY.md section 8d "throughput runs")
        #: inexact Newton: every Newton linear solve reduces its residual by this factor only (never
        #: below a tenth of the nonlinear target); 0.0 = exact Newton with `krylov_rtol`.  The Newton
        #: loop always stops on the reference's criterion, evaluated on the true nonlinear residual
        self.newton_forcing = 0.0
        #: start vector of the IPCS projection-step CG: "previous" (p_n) or "extrapolated" (2 p_n - p_(n-1))
        self.pressure_start = "previous"
        #: velocity mass solve of the IPCS correction step: "chebyshev" (a-priori bounds, no dots) or "cg"
        self.mass_solver = "chebyshev"
        #: IPCS projection step: "multigrid" (CG preconditioned by the pressure V-cycle) or
        #: "fast_diagonalization" -- the direct solve of poisson_fd.py / csrc/fastdiag.hip where it applies
        #: (rectangle lattices, pressure Dirichlet conditions on whole sides; elsewhere multigrid runs)
        self.poisson_solver = "multigrid"
        #: truncated velocity multigrid cycle: None = library default (ratio 4, tolerance 0.1),
        #: 0 / False = full cycle, R or (R, tol) = truncate where c_v K_ii <= R alpha0/k M_ii
        self.mg_truncation = None

    def throughput_settings(self, krylov_rtol=1.0e-8, newton_forcing=1.0e-4):
        """The settings of bench.py's timed steps: Krylov rtol 1e-8, inexact Newton (forcing 1e-4),
        extrapolated pressure start vector, Chebyshev mass solve, truncated velocity cycle.  The
        Newton stopping criterion stays the reference's (tol, 10 tol on the true residual)."""
        self.krylov_rtol = krylov_rtol
        self.newton_forcing = newton_forcing
        self.pressure_start = "extrapolated"
        self.mass_solver = "chebyshev"
        self.poisson_solver = "fast_diagonalization"
        self.mg_truncation = (4.0, 0.1)
        return self

    def _push_multigrid_settings(self):
        """truncation of the velocity cycle -> device (whenever the attribute changed)"""
        want = self.mg_truncation
        if want == getattr(self, "_mg_truncation_pushed", "unset") or not hasattr(self, "_ctx"):
            return
        if want is None:
            ratio, tol = 4.0, 0.1
        elif want is False or want == 0:
            ratio, tol = 0.0, 0.1
        elif isinstance(want, (tuple, list)):
            ratio, tol = float(want[0]), float(want[1])
        else:
            ratio, tol = float(want), 0.1
        self._ctx.mg_set_truncation(ratio, tol)
        self._mg_truncation_pushed = want

    def _common_step_options(self, o):
        """fields of nsfem_step_opts shared by the IPCS and the monolithic driver"""
        o.newton_atol = self._tol
        o.newton_rtol = 10.0 * self._tol
        o.newton_max_iter = self._maxiter
        o.matrix_free = {None: 0, False: 1, True: 2}[getattr(self, "matrix_free", None)]
        assert self.newton_forcing >= 0.0
        o.newton_forcing = float(self.newton_forcing)
        assert self.pressure_start in ("previous", "extrapolated")
        o.pressure_extrapolation = 1 if self.pressure_start == "extrapolated" else 0
        self._push_multigrid_settings()
        return o

    # -- state ------------------------------------------------------------------
    def _setup_function_spaces(self):
        super()._setup_function_spaces()
        levels = ((nat.U0, nat.P), (nat.U1, nat.P_OLD), (nat.U2, nat.P2_OLD))
        self._solutions = []
        for i in range(self._time_stepping.n_levels() + 1):
            name = i * "old" + (i > 0) * "_" + "solution"
            self._solutions.append(MixedFunction(self, *levels[i], name=name))

    def _advance_solution(self):
        """solutions[2] <- solutions[1] <- solutions[0] on the device."""
        self._ctx.advance(self._scheme_id)

    def _setup_problem(self):  # pragma: no cover
        raise NotImplementedError("You are calling a purely virtual method.")

    def _solve_time_step(self):  # pragma: no cover
        raise NotImplementedError("You are calling a purely virtual method.")

    def _update_time_stepping_coefficients(self):
        """alpha (first derivative) and the step size k -> device
        (reference: source/ns_ipcs_solver.py:210-227, source/ns_bdf_solver.py:108-126)."""
        self._next_step_size = self._time_stepping.get_next_step_size()
        self._alpha = list(self._time_stepping.coefficients(derivative=1))
        assert len(self._alpha) == 3
        self._ctx.set_bdf(self._alpha, self._next_step_size)

    def _assemble_system(self, system, new_step=False):
        """Explicit assembly seam (new): integrates matrix and residual /
        right-hand side of ``system`` from the current device state.  The reference
        does this implicitly inside ``dolfin.*VariationalSolver.solve()``."""
        self._ctx.assemble(system, new_step)

    def _set_time(self, next_time=None, current_time=None):
        """Move every time-dependent Expression (boundary values, tractions, body
        force) to t_{n+1} and refresh the device copies of their values
        (reference: :1033-1104)."""
        next_time = self._time_stepping.next_time if next_time is None else next_time
        current_time = self._time_stepping.current_time if current_time is None else current_time
        assert isinstance(next_time, float) and isinstance(current_time, float)
        assert next_time > current_time
        touched = []

        def move(value):
            if dlfn.is_time_dependent(value):
                if "time" in value._params:
                    value.time = next_time
                else:
                    value.t = next_time
                touched.append(value)

        for name in ("_velocity_bcs", "_pressure_bcs", "_traction_bcs"):
            for bc in getattr(self, name, []):
                move(bc[-1])
        if hasattr(self, "_body_force"):
            move(self._body_force)
        if touched:
            vd, vv = self._velocity_dirichlet_arrays()
            pd, pv = self._pressure_dirichlet_arrays()
            self._dirichlet_bcs = {"velocity": (vd, vv), "pressure": (pd, pv)}
            self._ctx.set_dirichlet(nat.VELOCITY, vd, vv)
            self._ctx.set_dirichlet(nat.PRESSURE, pd, pv)
            self._push_natural_terms()

    def advance_time(self):
        self._advance_solution()

    def _project(self, condition, field):
        """L2 projection onto P2^2 / P1: host quadrature of the load vector, mass solve
        on the device (reference: dlfn.project at :1151,1168)."""
        dm = self._dofmap
        if field == "velocity":
            if isinstance(condition, (tuple, list)):
                assert len(condition) == self._space_dim and all(isinstance(x, float) for x in condition)
                return np.tile(np.asarray(condition, dtype=np.float64), dm.n_p2)
            assert condition.value_rank() == 1
            b = fem_host.load_vector(self._mesh, dm.p2_dofmap, dm.n_p2,
                                     lambda X: dlfn.evaluate(condition, X), degree=2, n_comp=self._space_dim)
            return self._ctx.mass_solve(nat.VELOCITY, b)
        if isinstance(condition, float):
            return np.full(dm.n_p1, condition)
        assert condition.value_rank() == 0
        b = fem_host.load_vector(self._mesh, dm.p1_dofmap, dm.n_p1,
                                 lambda X: dlfn.evaluate(condition, X), degree=1, n_comp=1)
        return self._ctx.mass_solve(nat.PRESSURE, b)

    def set_initial_conditions(self, initial_conditions):
        """``{"velocity": tuple | Expression, "pressure": float | Expression}``; both
        the new and the old time level are initialised (reference: :1123-1171)."""
        assert isinstance(initial_conditions, dict) and "velocity" in initial_conditions
        if not all(hasattr(self, attr) for attr in ("_Wh", "_solutions")):
            self._setup_function_spaces()
        velocity_condition = initial_conditions["velocity"]
        assert isinstance(velocity_condition, _EXPRESSION_TYPES + (tuple, list))
        u0 = self._project(velocity_condition, "velocity")
        for level in (0, 1):
            self._solutions[level].sub(0).assign(u0)
        if "pressure" in initial_conditions:
            pressure_condition = initial_conditions["pressure"]
            assert isinstance(pressure_condition, _EXPRESSION_TYPES + (float,))
            p0 = self._project(pressure_condition, "pressure")
            for level in (0, 1):
                self._solutions[level].sub(1).assign(p0)

    def solve(self):
        """One time step (reference call order, :1174-1203)."""
        if not all(hasattr(self, attr) for attr in self._required_objects):
            self._setup_problem()
        self._set_time()
        self._push_angular_velocity()
        # the reference tests a bound method here (always true): refresh every step
        self._update_time_stepping_coefficients()
        self._solve_time_step()
        if hasattr(self, "_mean_pressure_value"):
            # p <- p - (mean(p) - target); exact for the reference's P1 projection
            self._last_mean_pressure = self._ctx.shift_mean_pressure(self._mean_pressure_value)

    @property
    def solution(self):
        return self._solutions[0]


class StationarySolverBase(SolverBase):
    """Stationary Navier-Stokes: hybrid Picard -> Newton iteration on the mixed P2-P1 system
    (reference: source/ns_solver_base.py:873-988).  The residual is the monolithic one without the
    acceleration term; the Picard matrix linearises the convection as ((grad v) u_k, w)
    (:930-934), the Newton matrix is the exact Jacobian (:936).  Both are integrated and solved
    on the device (nsfem_step_bdf with alpha = 0; block-preconditioned BiCGStab instead of
    PETSc LU)."""

    def __init__(self, mesh, boundary_markers, form_convective_term, tol=1e-10, maxiter=50,
                 tol_picard=1e-2, maxiter_picard=10, device=0):
        super().__init__(mesh, boundary_markers, form_convective_term, device=device)
        assert all(isinstance(i, int) and i > 0 for i in (maxiter, maxiter_picard))
        assert all(isinstance(i, float) and i > 0.0 for i in (tol, tol_picard))
        self._tol_picard = tol_picard
        self._maxiter_picard = maxiter_picard
        self._tol = tol
        self._maxiter = maxiter
        self.krylov_rtol = 1.0e-12
        self.krylov_max_iter = 5000
        self.use_multigrid = True
        self.pseudo_transient_fallback = True
        self.pseudo_transient_max_steps = 4000

    def _setup_problem(self):
        assert hasattr(self, "_equation_coefficients")
        self._setup_function_spaces()
        if self._mg_levels is None:
            raise RuntimeError("the stationary solver needs the multigrid block preconditioner")
        self._setup_boundary_conditions()
        from ns_bdf_solver import ImplicitBDFSolver, _FORM_ID
        ImplicitBDFSolver._push_schur_dirichlet_set(self)
        self._form_id = _FORM_ID[self._form_convective_term]
        self._ctx.set_convective_form(self._form_id)
        self._ctx.set_bdf((0.0, 0.0, 0.0), 1.0)              # no acceleration term
        self._solution = MixedFunction(self, nat.U0, nat.P, name="solution")
        self._nonlinear_solver = self._picard_problem = self._newton_problem = self._ctx

    def _nonlinear_solve(self, picard, atol, maxiter, allow_nonconvergence):
        o = self._ctx.default_step_opts()
        o.newton_atol = atol
        o.newton_rtol = 1.0e-9                                # dolfin NewtonSolver default
        o.newton_max_iter = maxiter
        o.convective_form = self._form_id
        o.picard = 1 if picard else 0
        o.allow_nonconvergence = 1 if allow_nonconvergence else 0
        o.momentum.rtol = self.krylov_rtol
        o.momentum.max_iter = self.krylov_max_iter
        o.momentum.precond = 1
        # The reference solves these systems by LU.  Here: block-preconditioned BiCGStab; when it
        # fails (cell Peclet number >> 1: the V-cycle on the viscous operator no longer resembles
        # the velocity block) the preconditioner is rebuilt for (J + M / tau), tau ~ 6 h / |u|
        # ("time-step preconditioner", nsfem_set_preconditioner_shift) and the iteration resumes
        # from the current iterate; the equations are never changed.
        # Last resort: pseudo-transient continuation (_pseudo_transient_solve) -- the same
        # stationary residual driven to zero by implicit Euler steps of the hot path.
        if getattr(self, "_use_pseudo_time", False) and maxiter > 1:
            return self._pseudo_transient_solve(o, atol)
        for attempt in range(4):
            try:
                return self._ctx.step_bdf(o)
            except nat.NativeError as err:
                if "BiCGStab" not in str(err):
                    raise RuntimeError(str(err))
                if attempt == 3:
                    if not self.pseudo_transient_fallback:
                        raise RuntimeError(str(err))
                    break
                self._preconditioner_shift = self._next_preconditioner_shift()
                dlfn.info("Krylov solver failed; preconditioner shift -> {0:.3g}".format(
                    self._preconditioner_shift))
                self._ctx.set_preconditioner_shift(self._preconditioner_shift)
        self._use_pseudo_time = True
        self._preconditioner_shift = 0.0
        self._ctx.set_preconditioner_shift(0.0)
        return self._pseudo_transient_solve(o, atol)

    def _pseudo_transient_solve(self, o, atol):
        """Pseudo-transient continuation: one Newton iteration of the implicit Euler step
        (M / tau + J(u)) du = -F(u) per pseudo-time step, with F the STATIONARY residual (the old
        time level is the current iterate) and tau adapted by switched evolution relaxation
        (tau grows like 1 / |F|, so the iteration turns into Newton's method near the solution),
        capped where the block-preconditioned Krylov solver stops converging.  The velocity mass
        matrix keeps the linear systems within reach of the multigrid preconditioner at cell
        Peclet numbers where the stationary Jacobian is not; the fixed point is the solution of
        the reference's stationary problem (source/ns_solver_base.py:951-988), which the reference
        reaches by LU-based Newton iterations."""
        ctx = self._ctx
        o.picard = 0
        o.newton_max_iter = 1
        o.newton_atol = atol
        o.allow_nonconvergence = 1
        # inexact steps: the convergence rate is set by tau, and no linear system needs to be
        # solved below the tolerance of the nonlinear iteration
        o.momentum.rtol = 1.0e-4
        o.momentum.atol = 0.02 * atol
        o.momentum.max_iter = min(self.krylov_max_iter, 300)
        u = ctx.get_state(nat.U0).reshape(-1, self._space_dim)
        speed = max(float(np.sqrt((u * u).sum(axis=1)).max()), 1.0e-12)
        bc_vals = self._dirichlet_bcs["velocity"][1]
        if bc_vals.size:
            speed = max(speed, float(np.abs(bc_vals).max()))
        tau = tau0 = 8.0 * self._mesh.hmin() / speed
        tau_cap = float("inf")
        res_prev, info, failures = None, None, 0
        self.pseudo_time_history = getattr(self, "pseudo_time_history", [])
        try:
            for it in range(self.pseudo_transient_max_steps):
                ctx.advance(0)                                   # old level <- current iterate
                ctx.set_bdf((1.0, -1.0, 0.0), tau)
                try:
                    info = ctx.step_bdf(o)
                except nat.NativeError as err:
                    if "BiCGStab" not in str(err):
                        raise RuntimeError(str(err))
                    failures += 1
                    if failures > 12:
                        raise RuntimeError("pseudo-transient continuation: " + str(err))
                    ctx.set_state(nat.U0, ctx.get_state(nat.U1))  # discard the failed update
                    ctx.set_state(nat.P, ctx.get_state(nat.P_OLD))
                    tau_cap = 0.5 * tau
                    tau = 0.25 * tau
                    continue
                res = info.newton_residuals[0]                   # = |F(u)| of the stationary problem
                self.pseudo_time_history.append((tau, res, info.krylov_iterations_momentum))
                if not np.isfinite(res):
                    raise RuntimeError("pseudo-transient continuation diverged")
                if res <= atol:
                    break
                if res_prev is not None:
                    tau = min(tau * min(max(res_prev / res, 0.5), 2.0), tau_cap)
                res_prev = res
                if it % 20 == 0:
                    dlfn.info("pseudo-time step {0}: tau = {1:.3g}, residual = {2:.3e}".format(it, tau, res))
        finally:
            ctx.set_bdf((0.0, 0.0, 0.0), 1.0)
        # report like the Newton solver: zero further iterations from the final iterate
        o.newton_atol = 1.0e300
        final = ctx.step_bdf(o)
        self.pseudo_time_steps = getattr(self, "pseudo_time_steps", 0) + it + 1
        return final

    def _next_preconditioner_shift(self):
        current = getattr(self, "_preconditioner_shift", 0.0)
        if current > 0.0:
            return 4.0 * current
        u = self._ctx.get_state(nat.U0).reshape(-1, self._space_dim)
        speed = float(np.sqrt((u * u).sum(axis=1)).max())
        bc_vals = self._dirichlet_bcs["velocity"][1]
        if bc_vals.size:
            speed = max(speed, float(np.abs(bc_vals).max()))
        return 0.15 * max(speed, 1.0e-12) / self._mesh.hmin()

    def solve(self):
        if not all(hasattr(self, attr) for attr in ("_nonlinear_solver", "_picard_problem",
                                                    "_newton_problem", "_solution")):
            self._setup_problem()
        self._push_angular_velocity()
        # initial residual (zero iterations of the nonlinear loop)
        info = self._nonlinear_solve(True, 1.0e300, 1, True)
        residual = info.newton_residuals[0]
        if residual < self._tol_picard and residual > 0.0:
            order = math.floor(math.log10(residual))
            self._tol_picard = (residual / 10.0 ** order - 1.0) * 10.0 ** order
        dlfn.info("Starting Picard iteration...")
        self.picard_info = self._nonlinear_solve(True, self._tol_picard, self._maxiter_picard, False)
        dlfn.info("Starting Newton iteration...")
        self.newton_info = self._nonlinear_solve(False, self._tol, self._maxiter, True)
        n = self.newton_info.newton_iterations
        residual = self.newton_info.newton_residuals[n]
        assert residual <= self._tol, "Newton iteration did not converge."
