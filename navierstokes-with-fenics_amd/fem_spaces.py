"""Host-side function spaces and functions on the Taylor-Hood dof map.

The reference keeps ``dolfin.Function`` objects on the mixed space ``Wh`` and on its collapsed
sub-spaces and copies between them with ``FunctionAssigner`` (``SolverBase._assign_function``,
source/ns_solver_base.py:213-300; ``_get_subspaces`` :424-476; exercised by
tests/test_function_assigner.py).  The device path never needs that gather -- velocity and
pressure stay split in HBM -- but the interface is part of the solver surface, so it exists here
on plain numpy storage:

* ``FunctionSpace(dofmap, kind)`` with kind "mixed" | "velocity" | "pressure"; ``Wh.sub(i)`` is
  the i-th sub-space AS A VIEW of the mixed space, ``Wh.sub(i).collapse()`` an independent space;
  ``function in space`` is dolfin's membership test (same space object).
* ``Function(space)``: coefficient vector laid out like the device state -- velocity
  node-interleaved (node * dim + component), pressure after it in a mixed function; ``split()``
  returns functions on ``Wh.sub(i)`` that share the parent's storage; ``f(x, y)`` evaluates.
"""
import numpy as np

from form_language import Operand

_KINDS = ("mixed", "velocity", "pressure")


def evaluate_lagrange(dofmap, field, values, point):
    """Value at ``point`` of the P2 vector (field "velocity", values [n_p2 * dim]) or P1 scalar
    (field "pressure") function; brute-force cell search (tests / diagnostics)."""
    mesh, dim = dofmap.mesh, dofmap.dim
    p = np.asarray(point, dtype=np.float64).reshape(-1)[:dim]
    x = mesh.coords[mesh.cells.astype(np.int64)]
    J = np.transpose(x[:, 1:] - x[:, :1], (0, 2, 1))                 # columns = edge vectors
    ref = np.linalg.solve(J, (p[None, :] - x[:, 0])[:, :, None])[:, :, 0]
    inside = np.nonzero((ref > -1e-12).all(axis=1) & (ref.sum(axis=1) < 1.0 + 1e-12))[0]
    if inside.size == 0:
        raise RuntimeError("point outside of the mesh")
    c = int(inside[0])
    l = np.concatenate([[1.0 - ref[c].sum()], ref[c]])
    if field == "pressure":
        return float(l @ np.asarray(values)[dofmap.p1_dofmap[c]])
    pairs = ((1, 2), (0, 2), (0, 1)) if dim == 2 else ((2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1))
    N = np.array([li * (2 * li - 1) for li in l] + [4 * l[a] * l[b] for a, b in pairs])
    return N @ np.asarray(values).reshape(-1, dim)[dofmap.p2_dofmap[c]]


class FunctionSpace:
    def __init__(self, dofmap, kind="mixed", parent=None):
        assert kind in _KINDS
        self.dofmap, self.kind, self.parent = dofmap, kind, parent
        self._subs = {}

    def mesh(self):
        return self.dofmap.mesh

    def num_sub_spaces(self):
        return 2 if self.kind == "mixed" else (self.dofmap.dim if self.kind == "velocity" else 0)

    def dim(self):
        dm = self.dofmap
        return {"mixed": dm.n_dofs, "velocity": dm.n_velocity, "pressure": dm.n_p1}[self.kind]

    def sub(self, i):
        assert self.kind == "mixed" and i in (0, 1)
        if i not in self._subs:
            self._subs[i] = FunctionSpace(self.dofmap, _KINDS[1 + i], parent=self)
        return self._subs[i]

    def collapse(self):
        return FunctionSpace(self.dofmap, self.kind)

    def offset(self):
        """first entry of this space inside its parent's coefficient vector"""
        return self.dofmap.n_velocity if (self.parent is not None and self.kind == "pressure") else 0

    def dof_coordinates(self):
        dm = self.dofmap
        if self.kind == "velocity":
            return np.repeat(dm.p2_coords, dm.dim, axis=0)
        if self.kind == "pressure":
            return dm.p1_coords
        return np.concatenate([np.repeat(dm.p2_coords, dm.dim, axis=0), dm.p1_coords])

    def __contains__(self, function):
        space = getattr(function, "function_space", None)
        return space is not None and space() is self


class Function(Operand):
    def __init__(self, space, values=None, name=None):
        assert isinstance(space, FunctionSpace)
        self._space = space
        if values is None:
            values = np.zeros(space.dim())
        assert values.shape == (space.dim(), )
        self._values = values
        self._name = name or {"mixed": "solution"}.get(space.kind, space.kind)

    def function_space(self):
        return self._space

    def name(self):
        return self._name

    def rename(self, name, label=""):
        self._name = name

    def vector(self):
        return self._values

    def assign(self, other):
        values = other.vector() if hasattr(other, "vector") else np.asarray(other, dtype=np.float64)
        assert values.shape == self._values.shape
        self._values[:] = values

    def sub(self, i):
        space = self._space.sub(i)
        start = space.offset()
        return Function(space, self._values[start: start + space.dim()], name=space.kind)

    def split(self, deepcopy=False):
        parts = tuple(self.sub(i) for i in range(2))
        if deepcopy:
            parts = tuple(Function(f.function_space().collapse(), f.vector().copy(), f.name()) for f in parts)
        return parts

    def __call__(self, *point):
        if len(point) == 1 and np.ndim(point[0]) > 0:
            point = tuple(point[0])
        dm = self._space.dofmap
        if self._space.kind == "mixed":
            nv = dm.n_velocity
            u = evaluate_lagrange(dm, "velocity", self._values[:nv], point)
            return np.concatenate([u, [evaluate_lagrange(dm, "pressure", self._values[nv:], point)]])
        return evaluate_lagrange(dm, self._space.kind, self._values, point)


def project(value, space, function=None):
    """``dolfin.project`` for the cases the solver surface needs: constants (exact) and functions
    the Lagrange space reproduces -- the nodal interpolant then IS the L2 projection.  Mixed
    spaces take a value with dim + 1 components (velocity, then pressure)."""
    from dlfn_compat import evaluate
    dm = space.dofmap
    out = function if function is not None else Function(space)
    assert out in space
    vals = out.vector()

    def fill_velocity(target, v):
        v = np.asarray(evaluate(v, dm.p2_coords), dtype=np.float64).reshape(dm.n_p2, -1)
        assert v.shape[1] == dm.dim
        target[:] = v.reshape(-1)

    if space.kind == "velocity":
        fill_velocity(vals, value)
    elif space.kind == "pressure":
        vals[:] = np.asarray(evaluate(value, dm.p1_coords), dtype=np.float64).reshape(-1)
    else:
        full = np.asarray(evaluate(value, dm.p2_coords), dtype=np.float64).reshape(dm.n_p2, -1)
        assert full.shape[1] == dm.dim + 1
        vals[: dm.n_velocity] = full[:, : dm.dim].reshape(-1)
        at_p1 = np.asarray(evaluate(value, dm.p1_coords), dtype=np.float64).reshape(dm.n_p1, -1)
        vals[dm.n_velocity:] = at_p1[:, dm.dim]
    return out
