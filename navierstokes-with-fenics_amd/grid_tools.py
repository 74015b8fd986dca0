"""Mesh conversion helper under the reference's module name (source/grid_tools.py:70-121).

``generate_xdmf_mesh(geo_file)`` keeps the reference's contract: given ``<name>.geo`` it locates
``<name>.msh`` (the reference would run gmsh when the file is missing -- there is no gmsh here,
so a missing file raises the same RuntimeError), and writes the two XDMF files the reference's
``_read_external_mesh`` consumes -- ``<name>.xdmf`` (cells + ``cell_markers``) and
``<name>_facet_markers.xdmf`` (boundary facets + ``facet_markers``), light XML data like
``meshio.write(..., data_format="XML")``.  meshio is absent; the .msh text is parsed by
``mesh_io.read_msh`` and the XDMF is written here.  ``read_xdmf_mesh`` is the way back: it loads
such a pair (ours or one produced by the reference's tool chain) into the solver's own
``Mesh`` / ``FacetMarkers`` objects; ``grid_generator._read_external_mesh`` uses it when only the
XDMF pair, not the .msh, is supplied."""
import glob
import xml.etree.ElementTree as ET
from os import path

import numpy as np

__all__ = ["generate_xdmf_mesh", "read_xdmf_mesh"]

_TOPOLOGY = {2: "Polyline", 3: "Triangle", 4: "Tetrahedron"}


def _locate_file(basename):
    """first file below (or beside) the working directory whose path contains ``basename``"""
    extension = path.splitext(basename)[1]
    candidates = []
    for pattern in ("./*", "./*/*", "./*/*/*"):
        candidates += sorted(glob.glob(pattern + extension))
    for name in candidates:
        if basename in name:
            return name
    return None


def _item(array, kind):
    a = np.asarray(array)
    dims = " ".join(str(d) for d in a.shape)
    fmt = "%d" if kind == "Int" else "%.17g"
    rows = a.reshape(a.shape[0], -1) if a.size else a.reshape(0, 1)
    text = "\n".join(" ".join(fmt % v for v in row) for row in rows)
    return '<DataItem DataType="%s" Dimensions="%s" Format="XML" Precision="8">\n%s\n</DataItem>' % (
        kind, dims, text)


def _write_grid(filename, points, cells, data_name, data):
    cells = np.asarray(cells, dtype=np.int64)
    geometry = "XY" if points.shape[1] == 2 else "XYZ"
    lines = ['<?xml version="1.0"?>', '<Xdmf Version="3.0">', "<Domain>", '<Grid Name="Grid">',
             '<Geometry GeometryType="%s">' % geometry, _item(points, "Float"), "</Geometry>",
             '<Topology TopologyType="%s" NumberOfElements="%d" NodesPerElement="%d">' % (
                 _TOPOLOGY[cells.shape[1]], cells.shape[0], cells.shape[1]),
             _item(cells, "Int"), "</Topology>",
             '<Attribute Name="%s" AttributeType="Scalar" Center="Cell">' % data_name,
             _item(np.asarray(data, dtype=np.int64), "Int"), "</Attribute>",
             "</Grid>", "</Domain>", "</Xdmf>"]
    with open(filename, "w") as fh:
        fh.write("\n".join(lines) + "\n")


def generate_xdmf_mesh(geo_file):
    """-> (xdmf_file, xdmf_facet_marker_file) next to the located .msh file."""
    from mesh_io import read_msh
    assert isinstance(geo_file, str)
    assert path.exists(geo_file)
    assert path.splitext(geo_file)[1] == ".geo"
    basename = path.basename(geo_file)
    msh_file = _locate_file(basename.replace(".geo", ".msh"))
    if msh_file is None:
        raise RuntimeError("GMSH is not installed on your machine and the msh file does not exist.")
    mesh, markers, _, cell_markers = read_msh(msh_file)
    marked = np.nonzero(markers.values != 0)[0]          # facets that carry a physical id
    xdmf_facet_marker_file = msh_file.replace(".msh", "_facet_markers.xdmf")
    _write_grid(xdmf_facet_marker_file, mesh.coords, mesh.facets[marked], "facet_markers",
                markers.values[marked])
    xdmf_file = msh_file.replace(".msh", ".xdmf")
    _write_grid(xdmf_file, mesh.coords, mesh.cells, "cell_markers", cell_markers)
    return xdmf_file, xdmf_facet_marker_file


def _load_grid(filename):
    grid = ET.parse(filename).getroot().find("Domain").find("Grid")

    def array(node, dtype):
        item = node.find("DataItem")
        if item.get("Format") != "XML":
            raise ValueError("%s: only XML light data is supported (no HDF5 library here)" % filename)
        dims = tuple(int(d) for d in item.get("Dimensions").split())
        return np.array((item.text or "").split(), dtype=dtype).reshape(dims)

    points = array(grid.find("Geometry"), np.float64)
    cells = array(grid.find("Topology"), np.int64)
    data = {a.get("Name"): array(a, np.float64).astype(np.int64) for a in grid.findall("Attribute")}
    return points, cells, data


def read_xdmf_mesh(xdmf_file, xdmf_facet_marker_file):
    """XDMF pair of ``generate_xdmf_mesh`` -> (Mesh, FacetMarkers, cell marker array)."""
    from fem_mesh import FacetMarkers, Mesh
    points, cells, cell_data = _load_grid(xdmf_file)
    dim = cells.shape[1] - 1
    mesh = Mesh(points[:, :dim].copy(), cells.astype(np.int32))
    markers = FacetMarkers(mesh, 0)
    fpoints, fcells, fdata = _load_grid(xdmf_facet_marker_file)
    values = fdata["facet_markers"].reshape(-1)
    assert fcells.shape[0] == values.size and fcells.shape[1] == dim
    # facet vertex ids refer to the point list of the facet file; both files written by the same
    # tool share it, otherwise the points are matched by coordinates
    if fpoints.shape[0] != points.shape[0] or np.abs(fpoints[:, :dim] - points[:, :dim]).max() > 0.0:
        lookup = {tuple(np.round(p, 12)): i for i, p in enumerate(points[:, :dim])}
        remap = np.array([lookup[tuple(np.round(p, 12))] for p in fpoints[:, :dim]], dtype=np.int64)
        fcells = remap[fcells]
    index = {tuple(sorted(f)): i for i, f in enumerate(mesh.facets.tolist())}
    for f, v in zip(fcells.tolist(), values.tolist()):
        markers.values[index[tuple(sorted(f))]] = v
    return mesh, markers, cell_data.get("cell_markers", np.zeros(cells.shape[0], np.int64)).reshape(-1)
