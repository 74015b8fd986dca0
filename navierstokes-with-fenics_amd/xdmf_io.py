"""XDMF field output (reference: ``dolfin.XDMFFile`` as configured in
source/ns_problem.py:39-53 and used by ``_write_xdmf_file`` :244-264).

Same call protocol -- ``XDMFFile(fname)``, ``.parameters[...]``, ``.write(function, t)`` -- and
the same on-disk model: ONE mesh shared by all functions (``functions_share_mesh``), written
once (``rewrite_function_mesh = False``), one temporal grid per output time holding every field
written at that time, light data re-flushed after every write (``flush_output``).

HDF5 is not available in this image (no h5py), so the heavy data goes either inline
(``encoding="xml"``) or into a raw little-endian side file ``<name>.bin`` referenced with
``Format="Binary"`` + ``Seek`` (default; both are standard XDMF 3 and open in ParaView).
Like dolfin's ``write`` the fields are stored as values at the mesh vertices (P2 velocity ->
its vertex dofs, P1 pressure as is); cell-wise post-processing fields (vorticity, pressure
gradient) are stored as cell attributes.  ``read_xdmf`` loads a file back for tests / restart.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np


class XDMFFile:
    def __init__(self, filename, encoding="binary"):
        assert filename.endswith(".xdmf")
        assert encoding in ("binary", "xml")
        self._filename = filename
        self._encoding = encoding
        self._bin_name = filename[:-5] + ".bin"
        self._bin_offset = 0
        self.parameters = {"flush_output": True, "functions_share_mesh": True,
                           "rewrite_function_mesh": False}
        self._mesh_items = None          # (topology xml, geometry xml) of the shared mesh
        self._steps = []                 # [(time, [attribute xml, ...])]
        if encoding == "binary" and os.path.exists(self._bin_name):
            os.remove(self._bin_name)

    # ---- heavy data ------------------------------------------------------------------------
    def _data_item(self, array, number_type):
        a = np.ascontiguousarray(array)
        dims = " ".join(str(d) for d in a.shape)
        prec = a.dtype.itemsize
        if self._encoding == "xml":
            fmt = "%d" if number_type == "Int" else "%.17g"
            rows = a.reshape(a.shape[0], -1)
            text = "\n".join(" ".join(fmt % v for v in row) for row in rows)
            return ('<DataItem Dimensions="%s" NumberType="%s" Precision="%d" Format="XML">\n%s\n'
                    '</DataItem>' % (dims, number_type, prec, text))
        with open(self._bin_name, "ab") as fh:
            fh.write(a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes())
        seek = self._bin_offset
        self._bin_offset += a.nbytes
        return ('<DataItem Dimensions="%s" NumberType="%s" Precision="%d" Format="Binary" '
                'Endian="Little" Seek="%d">%s</DataItem>'
                % (dims, number_type, prec, seek, os.path.basename(self._bin_name)))

    # ---- dolfin-like interface -----------------------------------------------------------------
    def _write_mesh(self, mesh):
        cells = np.asarray(mesh.cells, dtype=np.int32)
        coords = np.asarray(mesh.coords, dtype=np.float64)
        kind = "Triangle" if cells.shape[1] == 3 else "Tetrahedron"
        topo = ('<Topology TopologyType="%s" NumberOfElements="%d" NodesPerElement="%d">\n%s\n'
                '</Topology>' % (kind, cells.shape[0], cells.shape[1], self._data_item(cells, "Int")))
        geo = '<Geometry GeometryType="%s">\n%s\n</Geometry>' % (
            "XY" if coords.shape[1] == 2 else "XYZ", self._data_item(coords, "Float"))
        self._mesh_items = (topo, geo)

    def write(self, function, t=0.0):
        """Append ``function`` (DeviceFunction / HostField) at time ``t``."""
        from fem_function import vertex_or_cell_values
        mesh, center, values = vertex_or_cell_values(function)
        if self._mesh_items is None or self.parameters["rewrite_function_mesh"]:
            self._write_mesh(mesh)
        values = np.asarray(values, dtype=np.float64)
        if values.ndim == 2 and values.shape[1] == 2:            # ParaView wants 3-vectors
            values = np.concatenate([values, np.zeros((values.shape[0], 1))], axis=1)
        kind = "Scalar" if values.ndim == 1 else "Vector"
        attr = ('<Attribute Name="%s" AttributeType="%s" Center="%s">\n%s\n</Attribute>'
                % (function.name(), kind, center, self._data_item(values, "Float")))
        t = float(t)
        if self._steps and self._steps[-1][0] == t and self.parameters["functions_share_mesh"]:
            self._steps[-1][1].append(attr)
        else:
            self._steps.append((t, [attr]))
        if self.parameters["flush_output"]:
            self.flush()

    def flush(self):
        out = ['<?xml version="1.0"?>', '<Xdmf Version="3.0">', "<Domain>",
               '<Grid Name="TimeSeries" GridType="Collection" CollectionType="Temporal">']
        for i, (t, attrs) in enumerate(self._steps):
            out.append('<Grid Name="step_%d" GridType="Uniform">' % i)
            out.append('<Time Value="%.17g"/>' % t)
            if i == 0:
                out.extend(self._mesh_items)
            else:       # the mesh is written once and shared by reference
                out.append('<Topology Reference="XML">/Xdmf/Domain/Grid/Grid[1]/Topology</Topology>')
                out.append('<Geometry Reference="XML">/Xdmf/Domain/Grid/Grid[1]/Geometry</Geometry>')
            out.extend(attrs)
            out.append("</Grid>")
        out += ["</Grid>", "</Domain>", "</Xdmf>"]
        tmp = self._filename + ".tmp"
        with open(tmp, "w") as fh:
            fh.write("\n".join(out) + "\n")
        os.replace(tmp, self._filename)

    def close(self):
        self.flush()


def _load_item(item, directory):
    dims = tuple(int(d) for d in item.get("Dimensions").split())
    dtype = {"Int": {4: np.int32, 8: np.int64}, "Float": {4: np.float32, 8: np.float64}}[
        item.get("NumberType")][int(item.get("Precision"))]
    if item.get("Format") == "XML":
        return np.array(item.text.split(), dtype=dtype).reshape(dims)
    count = int(np.prod(dims))
    return np.fromfile(os.path.join(directory, item.text.strip()), dtype=np.dtype(dtype).newbyteorder("<"),
                       count=count, offset=int(item.get("Seek", "0"))).reshape(dims)


def read_xdmf(filename):
    """-> dict(cells, coords, times, fields={name: [array per time]}, centers={name: "Node"|"Cell"})."""
    root = ET.parse(filename).getroot()
    directory = os.path.dirname(os.path.abspath(filename))
    grids = root.find("Domain").find("Grid").findall("Grid")
    first = grids[0]
    out = {"cells": _load_item(first.find("Topology").find("DataItem"), directory),
           "coords": _load_item(first.find("Geometry").find("DataItem"), directory),
           "times": [], "fields": {}, "centers": {}}
    for g in grids:
        out["times"].append(float(g.find("Time").get("Value")))
        for a in g.findall("Attribute"):
            out["fields"].setdefault(a.get("Name"), []).append(_load_item(a.find("DataItem"), directory))
            out["centers"][a.get("Name")] = a.get("Center")
    return out
