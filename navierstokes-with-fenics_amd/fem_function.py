"""``dolfin.Function``-like views on device-resident state.

The reference hands ``dolfin.Function`` objects around (``solver.solution``,
``.split()``, ``.vector()``; source/ns_solver_base.py:1205-1207,
source/ns_ipcs_solver.py:241-247).  Here the values live in HBM inside the
``nsfem_ctx``; these objects are handles that copy to/from the host on demand.
"""
import numpy as np

from form_language import Operand


class DeviceFunction(Operand):
    """One field (velocity: node-interleaved P2^2, or pressure: P1) in one state slot."""

    def __init__(self, solver, field, slot, name=None):
        self._solver = solver
        self.field = field            # "velocity" | "pressure"
        self.slot = slot
        self._name = name or field

    def name(self):
        return self._name

    def rename(self, name, label=""):
        self._name = name

    def vector(self):
        """Host copy of the coefficient vector."""
        return self._solver._ctx.get_state(self.slot)

    def assign(self, values):
        if isinstance(values, DeviceFunction):
            values = values.vector()
        self._solver._ctx.set_state(self.slot, np.asarray(values, dtype=np.float64))

    def dof_coordinates(self):
        dm = self._solver._dofmap
        return dm.p2_coords if self.field == "velocity" else dm.p1_coords

    def nodal_values(self):
        v = self.vector()
        return v.reshape(-1, self._solver._dofmap.dim) if self.field == "velocity" else v

    def __call__(self, point):
        """Point evaluation (host side, brute-force cell search; for tests/diagnostics)."""
        from fem_spaces import evaluate_lagrange
        return evaluate_lagrange(self._solver._dofmap, self.field, self.vector(), point)


class MixedFunction:
    """(velocity, pressure) pair standing in for a Function on the mixed space."""

    def __init__(self, solver, velocity_slot, pressure_slot, name="solution"):
        self._solver = solver
        self._name = name
        self._parts = (DeviceFunction(solver, "velocity", velocity_slot),
                       DeviceFunction(solver, "pressure", pressure_slot))

    def name(self):
        return self._name

    def split(self, deepcopy=False):
        return self._parts

    def sub(self, i):
        return self._parts[i]

    def vector(self):
        """[velocity (node-interleaved) | pressure] host copy."""
        return np.concatenate([p.vector() for p in self._parts])

    def assign(self, other):
        for mine, theirs in zip(self._parts, other.split()):
            mine.assign(theirs)


class HostField:
    """Post-processing field living on the host: ``center`` "Node" (one value per mesh vertex) or
    "Cell" (one value per cell); the reference returns dolfin DG Functions here
    (source/ns_problem.py:55-103)."""

    def __init__(self, mesh, name, center, values):
        assert center in ("Node", "Cell")
        self.mesh, self._name, self.center = mesh, name, center
        self.values = np.asarray(values, dtype=np.float64)

    def name(self):
        return self._name

    def rename(self, name, label=""):
        self._name = name


def vertex_or_cell_values(function):
    """(mesh, "Node"|"Cell", values) of a DeviceFunction / HostField for file output: like
    dolfin's XDMFFile.write, Lagrange fields are reduced to their values at the mesh vertices."""
    if isinstance(function, HostField):
        return function.mesh, function.center, function.values
    dm = function._solver._dofmap
    mesh = dm.mesh
    cells = mesh.cells.astype(np.int64)
    nv = mesh.coords.shape[0]
    node_of_vertex = np.empty(nv, dtype=np.int64)
    if function.field == "velocity":
        node_of_vertex[cells.ravel()] = np.asarray(dm.p2_dofmap)[:, :cells.shape[1]].ravel()
    else:
        node_of_vertex[cells.ravel()] = np.asarray(dm.p1_dofmap).ravel()
    return mesh, "Node", function.nodal_values()[node_of_vertex]
