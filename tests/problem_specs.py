"""Table-driven flow problems for the class-level GPU tests.

Every problem of tests/test_solver_classes_gpu.py is a ``dict`` of DATA (mesh recipe, boundary
condition rows, coefficients, time window, scheme); ``build_problem`` turns one into an object
of the product's ``InstationaryProblem`` / ``StationaryProblem`` driver by filling in the hook
protocol of the problem base class (setup_mesh, set_boundary_conditions, ...).  The physical
set-ups themselves follow the configurations the reference exercises (cited per table row in the
test module); their expression as data, this builder and all assertions are this repository's.

Spec keys (all optional except ``mesh`` and ``bcs``):
  name           problem name (output directory)
  stationary     True -> StationaryProblem (Picard -> Newton), else InstationaryProblem
  mesh           ("rectangle", p0, p1, cells) | ("cube", dim, n) | ("open_cube", dim, n, openings)
                 | ("dfg", m, n_refine) | ("annulus", dim, radii, n) | ("plate", n) | ("step",)
  bcs            rows (kind, where, *args); kind: no_slip | velocity | velocity_component |
                 velocity_function | velocity_function_component | no_normal_flux |
                 pressure | pressure_function | pressure_mean | traction_component;
                 where = side name of the mesh recipe's marker table
  internal       rows like ``bcs`` applied as internal constraints
  numbers        dict(Re=..., Fr=..., Ro=...)
  start          {"velocity": tuple | expr, "pressure": float | expr}
  gravity        constant body force vector
  periodic       (axes, side names): periodic identification along the given axes of the unit box
  spin           ("constant", omega) | ("ramp", t_ramp, rate): angular velocity of the frame
  clock          dict(t0, t1, dt, steps)
  scheme         "ipcs" | "bdf"
  convection     weak form of the convective term
  output, postprocessing   frequencies; ``fields``: derived fields added in postprocess_solution;
                 ``hook``: callable(problem) run at the end of postprocess_solution
An expr is ``(strings, params, degree)`` -- C++ expression strings as dolfin.Expression takes them.
"""
import numpy as np

import dlfn_compat as dlfn
import grid_generator as gg
from auxiliary_classes import AngularVelocityVector, EquationCoefficientHandler, FunctionTime
from ns_bdf_solver import ImplicitBDFSolver
from ns_ipcs_solver import IPCSSolver
from ns_problem import InstationaryProblem, PressureBCType, TractionBCType, VelocityBCType
from ns_problem_stationary import StationaryProblem

SCHEMES = {"ipcs": IPCSSolver, "bdf": ImplicitBDFSolver}


def expr(strings, degree=2, **params):
    return (strings, params, degree)


def _expression(e):
    strings, params, degree = e
    return dlfn.Expression(strings, degree=degree, **params)


class BoxPeriodicity(dlfn.SubDomain):
    """periodic identification of opposite sides of the unit box along ``axes``: the low sides are
    the masters, a point on a high side maps to its image on the low side (first matching axis)"""

    def __init__(self, axes):
        super().__init__()
        self.axes = tuple(axes)

    def inside(self, x, on_boundary):
        return bool(on_boundary and any(dlfn.near(x[a], 0.0) for a in self.axes))

    def map(self, x_slave, x_master):
        for a in self.axes:
            if dlfn.near(x_slave[a], 1.0):
                x_master[:] = x_slave
                x_master[a] -= 1.0
                return
        x_master[:] = -10.0


class SpinUp(FunctionTime):
    """angular velocity rate * min(t, t_ramp) about the axis (constant: t_ramp = 0, value = rate)"""

    def __init__(self, t_ramp, rate):
        super().__init__(1)
        self.t_ramp, self.rate = float(t_ramp), float(rate)

    def value(self):
        return self.rate if self.t_ramp == 0.0 else self.rate * min(self._current_time, self.t_ramp)

    def derivative(self):
        return self.rate if (self.t_ramp > 0.0 and self._current_time < self.t_ramp) else 0.0


def make_mesh(recipe):
    """-> (mesh, markers, {side name: marker id})"""
    kind = recipe[0]
    box_sides = {m.name: m.value for m in gg.HyperCubeBoundaryMarkers}
    if kind == "rectangle":
        mesh, marks = gg.hyper_rectangle(*recipe[1:])
        return mesh, marks, box_sides
    if kind == "cube":
        mesh, marks = gg.hyper_cube(*recipe[1:])
        return mesh, marks, box_sides
    if kind == "open_cube":
        mesh, marks = gg.open_hyper_cube(*recipe[1:])
        return mesh, marks, box_sides
    if kind == "dfg":
        mesh, marks = gg.dfg_channel(*recipe[1:])
        return mesh, marks, {m.name: m.value for m in gg.DFGBoundaryMarkers}
    if kind == "annulus":
        mesh, marks = gg.spherical_shell(*recipe[1:])
        return mesh, marks, {"inner": gg.SphericalAnnulusBoundaryMarkers.interior_boundary.value,
                             "outer": gg.SphericalAnnulusBoundaryMarkers.exterior_boundary.value}
    if kind == "plate":
        return gg.blasius_plate(*recipe[1:])
    if kind == "step":
        return gg.backward_facing_step(*recipe[1:])
    raise ValueError(kind)


def _bc_rows(rows, sides):
    out = []
    for kind, where, *args in rows:
        mid = None if where is None else sides[where]
        if kind == "no_slip":
            out.append((VelocityBCType.no_slip, mid, None))
        elif kind == "velocity":
            out.append((VelocityBCType.constant, mid, tuple(args[0])))
        elif kind == "velocity_component":
            out.append((VelocityBCType.constant_component, mid, args[0], args[1]))
        elif kind == "velocity_function":
            out.append((VelocityBCType.function, mid, _expression(args[0])))
        elif kind == "velocity_function_component":
            out.append((VelocityBCType.function_component, mid, args[0], _expression(args[1])))
        elif kind == "no_normal_flux":
            out.append((VelocityBCType.no_normal_flux, mid, None))
        elif kind == "pressure":
            out.append((PressureBCType.constant, mid, float(args[0])))
        elif kind == "pressure_function":
            out.append((PressureBCType.function, mid, _expression(args[0])))
        elif kind == "pressure_mean":
            out.append((PressureBCType.mean_value, None, float(args[0])))
        elif kind == "traction_component":
            out.append((TractionBCType.constant_component, mid, args[0], float(args[1])))
        else:
            raise ValueError(kind)
    return tuple(out)


def _field_start(value):
    if isinstance(value, tuple) and len(value) == 3 and isinstance(value[1], dict):
        return _expression(value)
    return value


def build_problem(spec):
    """an InstationaryProblem / StationaryProblem object whose hooks read ``spec``"""
    base = StationaryProblem if spec.get("stationary") else InstationaryProblem

    class TableProblem(base):
        def __init__(self):
            if spec.get("stationary"):
                super().__init__(spec.get("main_dir"), form_convective_term=spec.get("convection", "standard"))
            else:
                c = spec["clock"]
                super().__init__(spec.get("main_dir"), start_time=c.get("t0", 0.0), end_time=c.get("t1", 1.0),
                                 desired_start_time_step=c["dt"], n_max_steps=c["steps"],
                                 form_convective_term=spec.get("convection", "standard"))
                self._output_frequency = spec.get("output", 0)
                self._postprocessing_frequency = spec.get("postprocessing", 0)
                self.set_solver_class(SCHEMES[spec.get("scheme", "bdf")])
            self._problem_name = spec.get("name", "TableProblem")
            self.spec = spec

        def setup_mesh(self):
            made = make_mesh(spec["mesh"])
            self._mesh, self._boundary_markers, self._sides = made
            if spec["mesh"][0] in ("plate", "step"):
                self._boundary_marker_map = self._sides

        def set_boundary_conditions(self):
            self._bcs = _bc_rows(spec["bcs"], self._sides)

        def set_equation_coefficients(self):
            self._coefficient_handler = EquationCoefficientHandler(**spec["numbers"])

    if "start" in spec:
        def set_initial_conditions(self):
            self._initial_conditions = {k: _field_start(v) for k, v in spec["start"].items()}
        TableProblem.set_initial_conditions = set_initial_conditions
    if "gravity" in spec:
        def set_body_force(self):
            self._body_force = dlfn.Constant(tuple(spec["gravity"]))
        TableProblem.set_body_force = set_body_force
    if "periodic" in spec:
        def set_periodic_boundary_conditions(self):
            axes, names = spec["periodic"]
            self._periodic_bcs = BoxPeriodicity(axes)
            self._periodic_boundary_ids = tuple(self._sides[n] for n in names)
        TableProblem.set_periodic_boundary_conditions = set_periodic_boundary_conditions
    if "spin" in spec:
        def set_angular_velocity(self):
            kind, *a = spec["spin"]
            fn = SpinUp(0.0, a[0]) if kind == "constant" else SpinUp(a[0], a[1])
            self._angular_velocity = AngularVelocityVector(2, function=fn)
        TableProblem.set_angular_velocity = set_angular_velocity
    if "internal" in spec:
        def set_internal_constraints(self):
            self._internal_constraints = _bc_rows(spec["internal"], self._sides)
        TableProblem.set_internal_constraints = set_internal_constraints
    if spec.get("fields") or spec.get("hook"):
        def postprocess_solution(self):
            for name in spec.get("fields", ()):
                self._add_to_field_output(getattr(self, "_compute_" + name)())
            if spec.get("hook"):
                spec["hook"](self)
        TableProblem.postprocess_solution = postprocess_solution
    return TableProblem()


def unique_dirichlet(dofs, vals):
    """later entries win on duplicate dofs (list order of DirichletBC.apply); sorted unique arrays"""
    dofs = np.asarray(dofs)
    _, first = np.unique(dofs[::-1], return_index=True)
    keep = len(dofs) - 1 - first
    return dofs[keep].astype(np.int64), np.asarray(vals)[keep]
