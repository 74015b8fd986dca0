"""Structured mesh factories with the reference's boundary-marker ids.

Same entry points, argument meaning and marker enumeration as the reference's
``source/grid_generator.py`` (hyper_cube :111-151, hyper_rectangle :154-208,
open_hyper_cube :211-353, HyperCubeBoundaryMarkers :36-46), producing the
dolfin-free ``fem_mesh.Mesh`` / ``FacetMarkers`` pair.  2D only: every 3D branch
of the reference solvers is "pragma: no cover" (SURVEY.md D4).  The mshr / gmsh
factories (spherical_shell, *.geo readers) need external tools that are absent.
"""
from enum import Enum, auto

import numpy as np

from fem_mesh import FacetMarkers, rectangle_mesh

_NEAR = 3.0e-16 * 1.0e3


class HyperCubeBoundaryMarkers(Enum):
    left = auto()
    right = auto()
    bottom = auto()
    top = auto()
    back = auto()
    front = auto()
    opening = auto()


HyperRectangleBoundaryMarkers = HyperCubeBoundaryMarkers


def _mark_box(mesh, lo, hi):
    markers = FacetMarkers(mesh, 0)
    ids = HyperCubeBoundaryMarkers
    tol = _NEAR * max(1.0, float(np.abs(np.array([lo, hi])).max()))
    for axis, value, marker in ((0, lo[0], ids.left), (0, hi[0], ids.right),
                                (1, lo[1], ids.bottom), (1, hi[1], ids.top)):
        markers.mark(lambda X, a=axis, v=value: np.abs(X[:, a] - v) < tol, marker.value)
    return markers


def hyper_cube(dim, n_points=10):
    """Unit square with an equidistant right-diagonal triangulation."""
    assert isinstance(dim, int) and dim in (2, 3)
    assert isinstance(n_points, int) and n_points >= 0
    if dim == 3:
        raise NotImplementedError("3D meshes are outside the 2D hot path (SURVEY.md D4)")
    mesh = rectangle_mesh((0.0, 0.0), (1.0, 1.0), n_points, n_points)
    return mesh, _mark_box(mesh, (0.0, 0.0), (1.0, 1.0))


def hyper_rectangle(first_point, second_point, n_points=10):
    """Rectangle spanned by two diagonally opposite corners."""
    assert isinstance(first_point, (tuple, list)) and isinstance(second_point, (tuple, list))
    dim = len(first_point)
    assert dim in (2, 3) and len(second_point) == dim
    assert all(isinstance(x, float) for p in (first_point, second_point) for x in p)
    assert all(b - a > 0.0 for a, b in zip(first_point, second_point))
    if isinstance(n_points, (tuple, list)):
        assert len(n_points) == dim and all(isinstance(n, int) and n > 0 for n in n_points)
    else:
        assert isinstance(n_points, int) and n_points > 0
        n_points = (n_points,) * dim
    if dim == 3:
        raise NotImplementedError("3D meshes are outside the 2D hot path (SURVEY.md D4)")
    mesh = rectangle_mesh(first_point, second_point, *n_points)
    return mesh, _mark_box(mesh, first_point, second_point)


def open_hyper_cube(dim, n_points=10, openings=None):
    """Unit square whose boundary carries ``opening`` markers on the given windows,
    ``openings = ((position, center, width), ...)`` with position in
    left/right/bottom/top."""
    if openings is None:  # pragma: no cover
        return hyper_cube(dim, n_points)
    assert isinstance(openings, (tuple, list))
    mesh, markers = hyper_cube(dim, n_points)
    ids = HyperCubeBoundaryMarkers
    side = dict(left=(0, 0.0, ids.left), right=(0, 1.0, ids.right),
                bottom=(1, 0.0, ids.bottom), top=(1, 1.0, ids.top))
    for position, center, width in openings:
        assert position in side, position
        assert isinstance(center, (tuple, list)) and len(center) == dim
        assert isinstance(width, float) and width > 0.0
        axis, value, marker = side[position]
        assert abs(center[axis] - value) < 1.0e3 * 3.0e-16, "Center point is not on the boundary"
        other = 1 - axis
        c, half = center[other], 0.5 * width

        def window(X, a=axis, v=value, o=other, c=c, half=half):
            return (np.abs(X[:, a] - v) < _NEAR) & (np.abs(X[:, o] - c) <= half)
        markers.mark(window, ids.opening.value)
    return mesh, markers
