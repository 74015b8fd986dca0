"""Shared helpers of the GPU parity tests (inputs only; no algorithm lives here)."""
import numpy as np

import _native as nat
from fem_mesh import FacetMarkers, TaylorHoodDofMap, rectangle_mesh


def have_gpu():
    try:
        import ctypes
        lib = nat.load_library()
        del lib
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


def box(nx, ny, p1=(1.0, 1.0)):
    mesh = rectangle_mesh((0.0, 0.0), p1, nx, ny)
    dm = TaylorHoodDofMap(mesh)
    marks = FacetMarkers(mesh)
    marks.mark(lambda X: np.abs(X[:, 0]) < 1e-12, 1)
    marks.mark(lambda X: np.abs(X[:, 0] - p1[0]) < 1e-12, 2)
    marks.mark(lambda X: np.abs(X[:, 1]) < 1e-12, 3)
    marks.mark(lambda X: np.abs(X[:, 1] - p1[1]) < 1e-12, 4)
    return mesh, dm, marks


def context(mesh, dm):
    return nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1)


def velocity_bc(dm, marks, spec):
    """spec: list of (marker id, fn(X) -> [n,2]); later entries win.  Returns unique dofs."""
    last = {}
    for mid, fn in spec:
        nodes = np.unique(dm.facet_p2_nodes(marks.facets_with_id(mid)))
        v = np.asarray(fn(dm.p2_coords[nodes]), dtype=np.float64)
        for a in range(2):
            for d, val in zip(2 * nodes + a, v[:, a]):
                last[int(d)] = float(val)
    d = np.array(sorted(last), dtype=np.int64)
    return d, np.array([last[i] for i in d])


def cavity_bc(dm, marks):
    zero = lambda X: np.zeros((X.shape[0], 2))
    lid = lambda X: np.tile([1.0, 0.0], (X.shape[0], 1))
    return velocity_bc(dm, marks, [(1, zero), (2, zero), (3, zero), (4, lid)])


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
