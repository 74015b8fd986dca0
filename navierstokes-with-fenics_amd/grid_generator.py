"""Structured mesh factories with the reference's boundary-marker ids.

Same entry points, argument meaning and marker enumeration as the reference's
``source/grid_generator.py`` (hyper_cube :111-151, hyper_rectangle :154-208,
open_hyper_cube :211-353, HyperCubeBoundaryMarkers :36-46), producing the
dolfin-free ``fem_mesh.Mesh`` / ``FacetMarkers`` pair.  hyper_cube / hyper_rectangle also build
the 3D BoxMesh (Kuhn tetrahedra; the reference's 3D solver branches are "pragma: no cover",
SURVEY.md D4, here they run).  mshr / gmsh are
absent: spherical_shell (2D annulus, 3D cubed-sphere shell of tetrahedra), the DFG channel and the
backward-facing step are meshed in-repo instead.
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


class SphericalAnnulusBoundaryMarkers(Enum):
    interior_boundary = auto()
    exterior_boundary = auto()


class GeometryType(Enum):
    """geometry tags of the reference (source/grid_generator.py:11-15)"""
    spherical_annulus = auto()
    rectangle = auto()
    square = auto()
    other = auto()


class SymmetricPipeBoundaryMarkers(Enum):
    """physical ids of the gmsh pipe geometry (source/grid_generator.py:26-33)"""
    wall = 100
    symmetry = 101
    inlet = 102
    outlet = 103


def _mark_box(mesh, lo, hi):
    markers = FacetMarkers(mesh, 0)
    ids = HyperCubeBoundaryMarkers
    tol = _NEAR * max(1.0, float(np.abs(np.array([lo, hi])).max()))
    sides = [(0, lo[0], ids.left), (0, hi[0], ids.right), (1, lo[1], ids.bottom), (1, hi[1], ids.top)]
    if len(lo) == 3:                    # reference: back / front = z planes (:143-149, :200-206)
        sides += [(2, lo[2], ids.back), (2, hi[2], ids.front)]
    for axis, value, marker in sides:
        markers.mark(lambda X, a=axis, v=value: np.abs(X[:, a] - v) < tol, marker.value)
    return markers


def hyper_cube(dim, n_points=10):
    """Unit square / cube with an equidistant right-diagonal triangulation / Kuhn (BoxMesh)
    tetrahedralisation."""
    assert isinstance(dim, int) and dim in (2, 3)
    assert isinstance(n_points, int) and n_points >= 0
    if dim == 3:
        from fem_mesh import box_mesh
        mesh = box_mesh((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), n_points, n_points, n_points)
        return mesh, _mark_box(mesh, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
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
        from fem_mesh import box_mesh
        mesh = box_mesh(first_point, second_point, *n_points)
        return mesh, _mark_box(mesh, first_point, second_point)
    mesh = rectangle_mesh(first_point, second_point, *n_points)
    return mesh, _mark_box(mesh, first_point, second_point)


def open_hyper_cube(dim, n_points=10, openings=None):
    """Unit square / cube whose boundary carries ``opening`` markers on the given windows,
    ``openings = ((position, center, width), ...)`` with position in left/right (x planes),
    bottom/top (y planes), back/front (z planes, 3D); width is a float in 2D and a pair of floats
    (the window's extent along the two in-plane axes, in axis order) in 3D
    (reference: source/grid_generator.py:211-353)."""
    if openings is None:  # pragma: no cover
        return hyper_cube(dim, n_points)
    assert isinstance(openings, (tuple, list))
    assert all(isinstance(o, (tuple, list)) for o in openings)
    mesh, markers = hyper_cube(dim, n_points)
    ids = HyperCubeBoundaryMarkers
    side = dict(left=(0, 0.0, ids.left), right=(0, 1.0, ids.right),
                bottom=(1, 0.0, ids.bottom), top=(1, 1.0, ids.top))
    if dim == 3:
        side.update(back=(2, 0.0, ids.back), front=(2, 1.0, ids.front))
    for position, center, width in openings:
        assert position in ("top", "bottom", "left", "right", "front", "back")
        assert position in side, "front / back openings need dim == 3"
        assert isinstance(center, (tuple, list)) and len(center) == dim
        assert all(isinstance(x, float) for x in center)
        if isinstance(width, float):
            assert dim == 2
            width = (width, )
        else:
            assert isinstance(width, (tuple, list)) and len(width) == dim - 1
            assert all(isinstance(x, float) and x > 0.0 for x in width)
        axis, value, marker = side[position]
        assert abs(center[axis] - value) < 1.0e3 * 3.0e-16, "Center point is not on the boundary"
        others = [a for a in range(dim) if a != axis]

        def window(X, a=axis, v=value, others=others, center=center, width=width):
            ok = np.abs(X[:, a] - v) < _NEAR
            for o, w in zip(others, width):
                ok &= np.abs(X[:, o] - center[o]) <= 0.5 * w
            return ok
        markers.mark(window, ids.opening.value)
    return mesh, markers


def _extract_facet_markers(geo_filename):
    """{physical group name: id} of the ``Physical Curve`` / ``Physical Line`` statements of a
    gmsh .geo file (reference: source/grid_generator.py:356-385)."""
    import os
    assert isinstance(geo_filename, str)
    assert os.path.exists(geo_filename)
    assert geo_filename.endswith(".geo")
    facet_markers = dict()
    with open(geo_filename, "r") as file:
        for line in file:
            if "Physical Curve" not in line and "Physical Line" not in line:
                continue
            inner = line[line.index("(") + 1: line.index(")")]
            assert "," in inner
            description, number = inner.split(",")
            number = number.strip()
            assert number.isnumeric()
            description = description.strip().strip("'").strip('"')
            assert description.replace(" ", "").isalpha()
            assert description not in facet_markers
            facet_markers[description] = int(number)
    return facet_markers


# ---------------------------------------------------------------------------------------------
# DFG 2D-2 channel with a cylinder (BASELINE config 3).  The reference reads this mesh from the
# un-vendored gmsh-collection submodule through gmsh + meshio (source/grid_generator.py:406-455,
# demo/dfg_benchmark.py); neither tool nor the .geo file is available, so an unstructured
# triangle mesh of the same non-dimensional geometry is generated here: channel 22 x 4.1,
# unit-diameter cylinder centred at (2, 2).
# ---------------------------------------------------------------------------------------------
class DFGBoundaryMarkers(Enum):
    inlet = 1
    outlet = 2
    bottom = 3
    top = 4
    cylinder = 5


_DFG = dict(length=22.0, height=4.1, center=(2.0, 2.0), radius=0.5)


def _dfg_project(mesh, markers, mid):
    """move the midpoints of cylinder edges onto the circle"""
    c, r = np.array(_DFG["center"]), _DFG["radius"]
    on = markers.values == DFGBoundaryMarkers.cylinder.value
    d = mid[on] - c
    mid[on] = c + r * d / np.linalg.norm(d, axis=1)[:, None]
    return mid


def dfg_channel(m=4, n_refine=0, grading=1.3):
    """Coarse block mesh (an O-grid of ``4 m`` sectors inside the box [1,3]^2 around the
    cylinder, structured right-diagonal triangles elsewhere), refined ``n_refine`` times with
    projection of new cylinder vertices onto the circle.  The returned mesh carries the multigrid
    hierarchy (``mesh.mg_levels``).  m must be even."""
    from fem_mesh import Mesh
    from multigrid import refinement_hierarchy
    assert m % 2 == 0 and m >= 2
    L, H, (cx, cy), R = _DFG["length"], _DFG["height"], _DFG["center"], _DFG["radius"]
    s = 2.0 / m
    n_right = int(np.ceil(19.0 / (1.5 * s)))
    n_top = max(1, int(round(1.1 / s)))
    xs = np.concatenate([np.linspace(0.0, 1.0, m // 2 + 1)[:-1], np.linspace(1.0, 3.0, m + 1)[:-1],
                         np.linspace(3.0, L, n_right + 1)])
    ys = np.concatenate([np.linspace(0.0, 1.0, m // 2 + 1)[:-1], np.linspace(1.0, 3.0, m + 1)[:-1],
                         np.linspace(3.0, H, n_top + 1)])
    nxg, nyg = xs.size - 1, ys.size - 1
    b0 = m // 2                                    # first box cell index (both directions)
    node_id = -np.ones((nyg + 1, nxg + 1), dtype=np.int64)
    coords = []
    for iy in range(nyg + 1):
        for ix in range(nxg + 1):
            if b0 < ix < b0 + m and b0 < iy < b0 + m:
                continue                           # strictly inside the box: replaced by the O-grid
            node_id[iy, ix] = len(coords)
            coords.append((xs[ix], ys[iy]))
    cells = []
    for iy in range(nyg):
        for ix in range(nxg):
            if b0 <= ix < b0 + m and b0 <= iy < b0 + m:
                continue
            v0, v1 = node_id[iy, ix], node_id[iy, ix + 1]
            v2, v3 = node_id[iy + 1, ix], node_id[iy + 1, ix + 1]
            cells += [(v0, v1, v3), (v0, v2, v3)]
    # box perimeter, counter-clockwise from the lower-left corner
    per = [(b0 + i, b0) for i in range(m)] + [(b0 + m, b0 + j) for j in range(m)] + \
          [(b0 + m - i, b0 + m) for i in range(m)] + [(b0, b0 + m - j) for j in range(m)]
    per_ids = [node_id[iy, ix] for ix, iy in per]
    n_rad = m // 2 + 1
    rings = [None] * (n_rad + 1)
    rings[n_rad] = per_ids
    q = np.array([coords[i] for i in per_ids])
    ang = np.arctan2(q[:, 1] - cy, q[:, 0] - cx)
    circle = np.stack([cx + R * np.cos(ang), cy + R * np.sin(ang)], axis=1)
    for j in range(n_rad):
        t = (j / n_rad) ** grading
        pts = circle + t * (q - circle)
        rings[j] = list(range(len(coords), len(coords) + len(per_ids)))
        coords += [tuple(p) for p in pts]
    K = len(per_ids)
    for j in range(n_rad):
        for k in range(K):
            a, b = rings[j][k], rings[j][(k + 1) % K]
            c, d = rings[j + 1][k], rings[j + 1][(k + 1) % K]
            cells += [(a, b, d), (a, c, d)]
    coarse = Mesh(np.array(coords), np.array(cells, dtype=np.int32))
    marks = FacetMarkers(coarse, 0)
    ids = DFGBoundaryMarkers
    marks.mark(lambda X: np.abs(X[:, 0]) < 1e-12, ids.inlet.value)
    marks.mark(lambda X: np.abs(X[:, 0] - L) < 1e-12, ids.outlet.value)
    marks.mark(lambda X: np.abs(X[:, 1]) < 1e-12, ids.bottom.value)
    marks.mark(lambda X: np.abs(X[:, 1] - H) < 1e-12, ids.top.value)
    e = coarse.edges
    rad = np.hypot(coarse.coords[:, 0] - cx, coarse.coords[:, 1] - cy)
    on_circle = np.abs(rad - R) < 1e-9
    cyl = coarse.edge_on_boundary & on_circle[e[:, 0]] & on_circle[e[:, 1]]
    marks.values[cyl] = ids.cylinder.value
    if n_refine == 0:
        coarse.mg_levels = []
        return coarse, marks
    return refinement_hierarchy(coarse, marks, n_refine, project=_dfg_project)


# ---------------------------------------------------------------------------------------------
# 2D annulus (reference: spherical_shell, source/grid_generator.py:67-108, built with mshr,
# which is absent).  Same arguments and marker ids; the triangulation is a polar grid refined
# with projection of new boundary vertices onto the two circles, so it carries a multigrid
# hierarchy.  ``n_points`` keeps mshr's meaning: about n_points cells across the diameter.
# ---------------------------------------------------------------------------------------------
def _spherical_shell_3d(ri, ro, n_points):
    """Tetrahedral mesh of the shell ri <= |x| <= ro (the reference meshes it with mshr, which is
    absent): a cubed sphere -- the six faces of a cube, each an m x m grid, projected radially and
    extruded through n_r layers -- whose hexahedra are cut into 24 tetrahedra around their centre
    and face centres, so neighbouring hexahedra (also across the patch seams) share the same face
    triangulation.  Face centres on the two boundaries are projected onto the spheres."""
    from fem_mesh import Mesh
    h = 2.0 * ro / max(n_points, 1)
    m = max(2, int(round(0.5 * np.pi * ro / h / 2.0)))         # cells along a patch edge
    n_r = max(1, int(round((ro - ri) / h / 2.0)))
    g = np.linspace(-1.0, 1.0, m + 1)
    g = np.tan(0.25 * np.pi * g)                                # equi-angular cubed sphere
    # surface lattice points of the cube [-1,1]^3, unique ids through the integer lattice key
    key_to_id, dirs = {}, []

    def surf_id(i, j, k):
        key = (i, j, k)
        if key not in key_to_id:
            key_to_id[key] = len(dirs)
            v = np.array([g[i], g[j], g[k]])
            dirs.append(v / np.linalg.norm(v))
        return key_to_id[key]

    quads = []
    for axis in range(3):
        for side in (0, m):
            for a in range(m):
                for b in range(m):
                    corner = []
                    for da, db in ((0, 0), (1, 0), (1, 1), (0, 1)):
                        idx = [0, 0, 0]
                        idx[axis] = side
                        idx[(axis + 1) % 3] = a + da
                        idx[(axis + 2) % 3] = b + db
                        corner.append(surf_id(*idx))
                    quads.append(corner)
    dirs = np.array(dirs)
    n_s = dirs.shape[0]
    radii_l = np.linspace(ri, ro, n_r + 1)
    coords = [r * dirs for r in radii_l]                        # layer l: ids l * n_s + surface id
    coords = list(np.concatenate(coords, axis=0))
    extra = {}

    def centre(ids, radius=None):
        key = tuple(sorted(ids))
        if key not in extra:
            c = np.mean([coords[i] for i in ids], axis=0)
            if radius is not None:
                c *= radius / np.linalg.norm(c)
            extra[key] = len(coords)
            coords.append(c)
        return extra[key]

    cells = []
    for layer in range(n_r):
        lo, hi = layer * n_s, (layer + 1) * n_s
        for q in quads:
            bot = [lo + v for v in q]
            top = [hi + v for v in q]
            hexa = bot + top
            cc = centre(hexa)
            faces = [(bot, ri if layer == 0 else None), (top, ro if layer == n_r - 1 else None)]
            for e in range(4):
                f = (e + 1) % 4
                faces.append(([bot[e], bot[f], top[f], top[e]], None))
            for quad, radius in faces:
                fc = centre(quad, radius)
                for e in range(4):
                    cells.append((quad[e], quad[(e + 1) % 4], fc, cc))
    mesh = Mesh(np.array(coords), np.array(cells, dtype=np.int32))
    ids = SphericalAnnulusBoundaryMarkers
    marks = FacetMarkers(mesh, 0)
    rv = np.linalg.norm(mesh.coords, axis=1)
    bf = np.nonzero(mesh.facet_on_boundary)[0]
    inner = np.abs(rv[mesh.facets[bf, 0]] - ri) < 1e-9 * ro
    marks.values[bf[inner]] = ids.interior_boundary.value
    marks.values[bf[~inner]] = ids.exterior_boundary.value
    return mesh, marks


def spherical_shell(dim, radii, n_points=10):
    from fem_mesh import Mesh
    from multigrid import refinement_hierarchy
    assert isinstance(dim, int) and dim in (2, 3)
    assert isinstance(radii, (list, tuple)) and len(radii) == 2
    ri, ro = radii
    assert isinstance(ri, float) and ri > 0.0 and isinstance(ro, float) and ro > ri
    assert isinstance(n_points, int) and n_points >= 0
    if dim == 3:
        return _spherical_shell_3d(ri, ro, n_points)
    h = 2.0 * ro / max(n_points, 1)
    n_r = max(1, int(round((ro - ri) / h)))
    n_refine = 0
    while n_r % 2 == 0 and n_r > 2:
        n_r //= 2
        n_refine += 1
    n_t = max(8, int(np.ceil(np.pi * (ri + ro) / h / 2 ** n_refine)))
    n_t += (-n_t) % 4
    r = np.linspace(ri, ro, n_r + 1)
    t = 2.0 * np.pi * np.arange(n_t) / n_t
    coords = np.stack([np.outer(r, np.cos(t)).ravel(), np.outer(r, np.sin(t)).ravel()], axis=1)
    vid = lambda i, j: i * n_t + (j % n_t)
    cells = []
    for i in range(n_r):
        for j in range(n_t):
            a, b, c, d = vid(i, j), vid(i + 1, j), vid(i + 1, j + 1), vid(i, j + 1)
            cells += [(a, b, c), (a, c, d)]
    coarse = Mesh(coords, np.array(cells, dtype=np.int32))
    ids = SphericalAnnulusBoundaryMarkers
    marks = FacetMarkers(coarse, 0)
    rad = lambda X: np.hypot(X[:, 0], X[:, 1])
    # chords of the boundary polygons: both end points on the circle (midpoints lie inside)
    e = coarse.edges[coarse.edge_on_boundary]
    rv = rad(coarse.coords)
    on_bd = np.nonzero(coarse.edge_on_boundary)[0]
    inner = np.abs(rv[e[:, 0]] - ri) < 1e-12 * ro
    marks.values[on_bd[inner]] = ids.interior_boundary.value
    marks.values[on_bd[~inner]] = ids.exterior_boundary.value

    def project(mesh, markers, mid):
        for value, radius in ((ids.interior_boundary.value, ri), (ids.exterior_boundary.value, ro)):
            on = markers.values == value
            mid[on] *= (radius / rad(mid[on]))[:, None]
        return mid

    if n_refine == 0:
        return coarse, marks
    return refinement_hierarchy(coarse, marks, n_refine, project=project)


# ---------------------------------------------------------------------------------------------
# externally generated meshes (reference: _read_external_mesh and its three users,
# source/grid_generator.py:357-455).  The reference runs gmsh on a .geo file of the un-vendored
# gmsh-collection and converts through meshio; here the gmsh OUTPUT (<name>.msh, ASCII 2.2 or
# 4.1) is read directly when the user supplies it anywhere below the working directory.
# ---------------------------------------------------------------------------------------------
def _locate_file(basename):
    import os
    for root, _, files in os.walk(os.getcwd()):
        if basename in files:
            return os.path.join(root, basename)
    return None


def _read_external_mesh(basename):
    """-> (mesh, facet markers, {physical name: id}) from ``<basename>.msh`` or, when only the
    reference's converted files are supplied, from the XDMF pair ``<basename>.xdmf`` /
    ``<basename>_facet_markers.xdmf`` of ``grid_tools.generate_xdmf_mesh`` (the marker names
    then come from the .geo file, as in the reference: source/grid_generator.py:406-437)."""
    from mesh_io import read_msh
    assert isinstance(basename, str) and basename.endswith(".geo")
    msh_file = _locate_file(basename.replace(".geo", ".msh"))
    if msh_file is not None:
        mesh, markers, names, _ = read_msh(msh_file)
        return mesh, markers, {name: tag for name, (dim, tag) in names.items() if dim == mesh._dim - 1}
    xdmf_file = _locate_file(basename.replace(".geo", ".xdmf"))
    facet_file = _locate_file(basename.replace(".geo", "_facet_markers.xdmf"))
    if xdmf_file is None or facet_file is None:
        raise FileNotFoundError(
            "%s not found below the working directory: generate it with `gmsh -2 -format msh41 %s` "
            "(gmsh itself is not available in this environment)" % (basename.replace(".geo", ".msh"), basename))
    from grid_tools import read_xdmf_mesh
    mesh, markers, _ = read_xdmf_mesh(xdmf_file, facet_file)
    geo_file = _locate_file(basename)
    names = _extract_facet_markers(geo_file) if geo_file is not None else \
        {str(v): v for v in sorted(markers.ids(boundary_only=False)) if v != 0}
    return mesh, markers, names


class BackwardFacingStepMarkers(Enum):
    """Physical groups of the step geometry as the demo uses them
    (demo/backward_facing_step.py:21-26: "inlet", "walls"; the outlet stays natural)."""
    inlet = auto()
    outlet = auto()
    walls = auto()


def backward_facing_step_mesh(m=2, n_refine=2, inlet_length=1.0, length=8.0, step_height=0.5):
    """In-repo triangulation of the backward-facing-step channel of demo/backward_facing_step.py
    (inlet of height h = 0.5 on y in [0.5, 1] -- the demo's inlet profile fixes that -- opening
    into a channel of height 1): right-diagonal triangles on the L-shaped union of
    [0, inlet_length] x [step_height, 1] and [inlet_length, length] x [0, 1], ``m`` cells per
    step height on the coarse mesh, refined ``n_refine`` times (-> multigrid hierarchy)."""
    from fem_mesh import Mesh
    from multigrid import refinement_hierarchy
    nx1 = max(1, int(round(inlet_length / step_height * m)))
    nx2 = max(1, int(round((length - inlet_length) / step_height * m)))
    ny1 = m
    ny2 = max(1, int(round((1.0 - step_height) / step_height * m)))
    xs = np.concatenate([np.linspace(0.0, inlet_length, nx1 + 1)[:-1], np.linspace(inlet_length, length, nx2 + 1)])
    ys = np.concatenate([np.linspace(0.0, step_height, ny1 + 1)[:-1], np.linspace(step_height, 1.0, ny2 + 1)])
    node_id = -np.ones((ys.size, xs.size), dtype=np.int64)
    coords = []
    for iy in range(ys.size):
        for ix in range(xs.size):
            if ix < nx1 and iy < ny1:
                continue                               # inside the step block
            node_id[iy, ix] = len(coords)
            coords.append((xs[ix], ys[iy]))
    cells = []
    for iy in range(ys.size - 1):
        for ix in range(xs.size - 1):
            if ix < nx1 and iy < ny1:
                continue
            v0, v1 = node_id[iy, ix], node_id[iy, ix + 1]
            v2, v3 = node_id[iy + 1, ix], node_id[iy + 1, ix + 1]
            cells += [(v0, v1, v3), (v0, v2, v3)]
    coarse = Mesh(np.array(coords), np.array(cells, dtype=np.int32))
    marks = FacetMarkers(coarse, 0)
    ids = BackwardFacingStepMarkers
    marks.mark(lambda X: X[:, 0] > -1.0, ids.walls.value)             # every boundary facet ...
    marks.mark(lambda X: np.abs(X[:, 0]) < 1e-12, ids.inlet.value)    # ... except the two ends
    marks.mark(lambda X: np.abs(X[:, 0] - length) < 1e-12, ids.outlet.value)
    if n_refine == 0:
        coarse.mg_levels = []
        return coarse, marks
    return refinement_hierarchy(coarse, marks, n_refine)


def backward_facing_step(m=2, n_refine=2):
    """BackwardFacingStep.msh when supplied (reference: source/grid_generator.py, gmsh), else the
    in-repo triangulation with the marker map the .geo file defines."""
    try:
        return _read_external_mesh("BackwardFacingStep.geo")
    except FileNotFoundError:
        mesh, markers = backward_facing_step_mesh(m, n_refine)
        return mesh, markers, {marker.name: marker.value for marker in BackwardFacingStepMarkers}


class BlasiusPlateMarkers(Enum):
    """Physical groups the Blasius demo / test use (demo/blasius_flow.py:22-38): the plate is an
    INTERNAL line (no-slip imposed through ``internal_constraints``)."""
    inlet = auto()
    outlet = auto()
    bottom = auto()
    top = auto()
    plate = auto()


def blasius_plate_mesh(n=16, length=2.0, height=1.0, plate=(0.5, 1.5)):
    """In-repo stand-in for BlasiusFlowProblem.geo: the channel [0, length] x [0, height] with a
    thin plate on the centre line y = height / 2 between x = plate[0] and plate[1], marked on
    INTERIOR facets; right-diagonal triangles, ``n`` cells across the height (even, so that the
    plate lies on a lattice line).  The structured mesh carries the multigrid hierarchy."""
    assert n % 2 == 0
    nx = int(round(n * length / height))
    mesh = rectangle_mesh((0.0, 0.0), (length, height), nx, n)
    marks = FacetMarkers(mesh, 0)
    ids = BlasiusPlateMarkers
    marks.mark(lambda X: np.abs(X[:, 0]) < 1e-12, ids.inlet.value)
    marks.mark(lambda X: np.abs(X[:, 0] - length) < 1e-12, ids.outlet.value)
    marks.mark(lambda X: np.abs(X[:, 1]) < 1e-12, ids.bottom.value)
    marks.mark(lambda X: np.abs(X[:, 1] - height) < 1e-12, ids.top.value)
    on_plate = lambda X: (np.abs(X[:, 1] - 0.5 * height) < 1e-12) & (X[:, 0] > plate[0] - 1e-12) & \
        (X[:, 0] < plate[1] + 1e-12)
    marks.mark(on_plate, ids.plate.value, boundary_only=False)
    return mesh, marks


def blasius_plate(n=16):
    """BlasiusFlowProblem.msh when supplied, else the in-repo channel with an internal plate and
    the marker map the .geo file defines."""
    try:
        return _read_external_mesh("BlasiusFlowProblem.geo")
    except FileNotFoundError:
        mesh, markers = blasius_plate_mesh(n)
        return mesh, markers, {marker.name: marker.value for marker in BlasiusPlateMarkers}


def channel_with_cylinder(m=4, n_refine=2):
    """DFGBenchmark.msh when supplied, else the in-repo triangulation of the same geometry
    (``dfg_channel``) with the marker map the .geo file would define."""
    try:
        return _read_external_mesh("DFGBenchmark.geo")
    except FileNotFoundError:
        mesh, markers = dfg_channel(m, n_refine)
        names = {marker.name: marker.value for marker in DFGBoundaryMarkers}
        # the physical group names of DFGBenchmark.geo as demo/dfg_benchmark.py uses them
        names["lower wall"], names["upper wall"] = names["bottom"], names["top"]
        return mesh, markers, names
