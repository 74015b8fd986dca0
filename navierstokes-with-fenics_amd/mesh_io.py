"""Mesh ingest for externally generated meshes: gmsh ``.msh`` (ASCII, format 2.2 and 4.1).

The reference converts ``.geo -> .msh`` with the gmsh executable, reads it with meshio, writes
XDMF and re-reads it with dolfin, taking the boundary ids from the physical groups
(source/grid_generator.py:406-455, source/grid_tools.py:70-121).  Neither gmsh nor meshio is
available here; this module reads the ``.msh`` text directly into the solver's own mesh
objects: triangles -> ``fem_mesh.Mesh``, physical ids of the boundary lines -> ``FacetMarkers``
(the reference's marker convention: facet marker = physical group id).  ``write_msh`` exports a
mesh + markers as format 2.2 (round trips, hand-off to other tools).
"""
import numpy as np

from fem_mesh import FacetMarkers, Mesh

_LINE, _TRIANGLE, _POINT = 1, 2, 15          # gmsh element type ids (first order)
_NODES_PER_TYPE = {1: 2, 2: 3, 3: 4, 4: 4, 5: 8, 6: 6, 7: 5, 8: 3, 9: 6, 10: 9, 11: 10, 15: 1}


def _sections(path):
    out, name, buf = {}, None, []
    with open(path) as fh:
        for raw in fh:
            line = raw.strip()
            if not line:
                continue
            if line.startswith("$End"):
                out[name] = buf
                name, buf = None, []
            elif line.startswith("$"):
                name, buf = line[1:], []
            elif name is not None:
                buf.append(line)
    return out


def _physical_names(sec):
    names = {}
    for line in sec.get("PhysicalNames", [])[1:]:
        dim, tag, name = line.split(maxsplit=2)
        names[name.strip('"')] = (int(dim), int(tag))
    return names


def _read_v2(sec):
    lines = sec["Nodes"]
    n = int(lines[0])
    ids = np.empty(n, dtype=np.int64)
    xyz = np.empty((n, 3))
    for i, line in enumerate(lines[1:n + 1]):
        p = line.split()
        ids[i] = int(p[0])
        xyz[i] = [float(v) for v in p[1:4]]
    tris, tri_phys, segs, seg_phys = [], [], [], []
    lines = sec["Elements"]
    for line in lines[1:int(lines[0]) + 1]:
        p = [int(v) for v in line.split()]
        etype, ntags = p[1], p[2]
        phys = p[3] if ntags > 0 else 0
        nodes = p[3 + ntags:]
        if etype == _TRIANGLE:
            tris.append(nodes)
            tri_phys.append(phys)
        elif etype == _LINE:
            segs.append(nodes)
            seg_phys.append(phys)
        elif etype not in _NODES_PER_TYPE:
            raise ValueError("unknown gmsh element type %d" % etype)
        elif etype not in (_POINT,):
            raise ValueError("only first-order triangle meshes are supported (element type %d)" % etype)
    return ids, xyz, tris, tri_phys, segs, seg_phys


def _read_v4(sec):
    # entity tag -> physical tag, per dimension
    ent_phys = {0: {}, 1: {}, 2: {}, 3: {}}
    if "Entities" in sec:
        lines = sec["Entities"]
        counts = [int(v) for v in lines[0].split()]
        row = 1
        for dim, count in enumerate(counts):
            for _ in range(count):
                p = lines[row].split()
                row += 1
                tag = int(p[0])
                k = 4 if dim == 0 else 7                      # point: x y z ; others: bounding box
                nphys = int(p[k])
                ent_phys[dim][tag] = int(p[k + 1]) if nphys > 0 else 0
    lines = sec["Nodes"]
    nblocks, n = int(lines[0].split()[0]), int(lines[0].split()[1])
    ids = np.empty(n, dtype=np.int64)
    xyz = np.empty((n, 3))
    row, pos = 1, 0
    for _ in range(nblocks):
        _, _, parametric, nb = (int(v) for v in lines[row].split())
        row += 1
        for k in range(nb):
            ids[pos + k] = int(lines[row + k])
        for k in range(nb):
            xyz[pos + k] = [float(v) for v in lines[row + nb + k].split()[:3]]
        row += 2 * nb
        pos += nb
    tris, tri_phys, segs, seg_phys = [], [], [], []
    lines = sec["Elements"]
    nblocks = int(lines[0].split()[0])
    row = 1
    for _ in range(nblocks):
        dim, tag, etype, nb = (int(v) for v in lines[row].split())
        row += 1
        phys = ent_phys.get(dim, {}).get(tag, 0)
        for k in range(nb):
            nodes = [int(v) for v in lines[row + k].split()[1:]]
            if etype == _TRIANGLE:
                tris.append(nodes)
                tri_phys.append(phys)
            elif etype == _LINE:
                segs.append(nodes)
                seg_phys.append(phys)
            elif etype != _POINT:
                raise ValueError("only first-order triangle meshes are supported (element type %d)" % etype)
        row += nb
    return ids, xyz, tris, tri_phys, segs, seg_phys


def read_msh(path):
    """-> (mesh, facet markers, {physical name: (dim, id)}, cell physical ids)."""
    sec = _sections(path)
    version = float(sec["MeshFormat"][0].split()[0])
    if int(sec["MeshFormat"][0].split()[1]) != 0:
        raise ValueError("binary .msh files are not supported; export ASCII")
    ids, xyz, tris, tri_phys, segs, seg_phys = _read_v2(sec) if version < 4.0 else _read_v4(sec)
    if not tris:
        raise ValueError("no triangles in " + path)
    if np.abs(xyz[:, 2]).max() > 1e-12 * max(1.0, np.abs(xyz).max()):
        raise ValueError("only planar (z = 0) meshes are supported")
    tris = np.asarray(tris, dtype=np.int64)
    used = np.unique(tris)                                  # drop construction points (circle centres)
    lookup = np.full(int(ids.max()) + 1, -1, dtype=np.int64)
    pos_of_id = np.full(int(ids.max()) + 1, -1, dtype=np.int64)
    pos_of_id[ids] = np.arange(ids.size)
    lookup[used] = np.arange(used.size)
    coords = xyz[pos_of_id[used], :2]
    cells = lookup[tris]
    # positive orientation, as the in-repo generators produce
    a, b, c = coords[cells[:, 0]], coords[cells[:, 1]], coords[cells[:, 2]]
    neg = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0]) < 0
    cells[neg] = cells[neg][:, [0, 2, 1]]
    mesh = Mesh(coords, cells.astype(np.int32))
    marks = FacetMarkers(mesh, 0)
    if segs:
        s = lookup[np.asarray(segs, dtype=np.int64)]
        nv = mesh.num_vertices()
        key = np.minimum(s[:, 0], s[:, 1]) * nv + np.maximum(s[:, 0], s[:, 1])
        ekey = mesh.edges[:, 0].astype(np.int64) * nv + mesh.edges[:, 1]
        pos = np.searchsorted(ekey, key)
        ok = (pos < ekey.size) & (ekey[np.minimum(pos, ekey.size - 1)] == key)
        if not ok.all():
            raise ValueError("a marked line element is not an edge of the triangulation")
        marks.values[pos] = np.asarray(seg_phys, dtype=marks.values.dtype)
    return mesh, marks, _physical_names(sec), np.asarray(tri_phys, dtype=np.int32)


def write_msh(path, mesh, markers=None, physical_names=None):
    """Format 2.2 ASCII: all triangles (physical id 1 unless given by names) + marked facets."""
    with open(path, "w") as fh:
        fh.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n")
        if physical_names:
            fh.write("$PhysicalNames\n%d\n" % len(physical_names))
            for name, (dim, tag) in physical_names.items():
                fh.write('%d %d "%s"\n' % (dim, tag, name))
            fh.write("$EndPhysicalNames\n")
        fh.write("$Nodes\n%d\n" % mesh.num_vertices())
        for i, (x, y) in enumerate(mesh.coords):
            fh.write("%d %.17g %.17g 0\n" % (i + 1, x, y))
        fh.write("$EndNodes\n")
        marked = np.nonzero(markers.values != 0)[0] if markers is not None else np.zeros(0, int)
        fh.write("$Elements\n%d\n" % (marked.size + mesh.num_cells()))
        eid = 1
        for f in marked:
            a, b = mesh.edges[f]
            tag = int(markers.values[f])
            fh.write("%d 1 2 %d %d %d %d\n" % (eid, tag, tag, a + 1, b + 1))
            eid += 1
        for a, b, c in mesh.cells:
            fh.write("%d 2 2 1 1 %d %d %d\n" % (eid, a + 1, b + 1, c + 1))
            eid += 1
        fh.write("$EndElements\n")
