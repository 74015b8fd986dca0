"""Dolfin-free mesh / facet-marker / Taylor-Hood dof-map containers.

These are the objects that cross the solver boundary in the reference
(``dlfn.Mesh`` and ``dlfn.MeshFunction("size_t", mesh, dim-1)``, consumed at
source/ns_solver_base.py:78-95).  Only what the hot path needs is provided:
vertex coordinates, cell->vertex connectivity, the edge (facet) entities with
their markers, and the P2 / P1 cell dof maps handed to the C-ABI
(include/nsfem.h: nsfem_mesh_desc).

Vertex order, "right" diagonal and marker ids follow dolfin's RectangleMesh as
used by source/grid_generator.py:111-208.
"""
import numpy as np


class _Geometry:
    def __init__(self, dim):
        self._dim = dim

    def dim(self):
        return self._dim


class _Topology(_Geometry):
    pass


class Mesh:
    """Triangular mesh: ``coords`` [nv, 2] float64, ``cells`` [nc, 3] int32."""

    def __init__(self, coords, cells):
        self.coords = np.ascontiguousarray(coords, dtype=np.float64)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        assert self.coords.ndim == 2 and self.coords.shape[1] == 2, "2D simplex meshes only"
        assert self.cells.ndim == 2 and self.cells.shape[1] == 3
        nv = self.coords.shape[0]
        c = self.cells.astype(np.int64)
        # local edges in UFC order: e0 = (v1, v2), e1 = (v0, v2), e2 = (v0, v1)
        pairs = np.stack([c[:, [1, 2]], c[:, [0, 2]], c[:, [0, 1]]], axis=1)   # [nc, 3, 2]
        lo = pairs.min(axis=2)
        hi = pairs.max(axis=2)
        key = lo * nv + hi
        ukey, inv, counts = np.unique(key.ravel(), return_inverse=True, return_counts=True)
        self.cell_edges = inv.reshape(-1, 3).astype(np.int32)
        self.edges = np.stack([ukey // nv, ukey % nv], axis=1).astype(np.int32)     # [ne, 2]
        self.edge_on_boundary = counts == 1
        # one adjacent cell per edge (the only one for boundary edges)
        self.edge_cell = np.empty(self.edges.shape[0], dtype=np.int32)
        self.edge_cell[self.cell_edges.ravel()] = np.repeat(np.arange(c.shape[0], dtype=np.int32), 3)
        self._dim = 2

    # -- the slice of the dolfin.Mesh API the reference touches -----------------
    def geometry(self):
        return _Geometry(self._dim)

    def topology(self):
        return _Topology(self._dim)

    def num_cells(self):
        return int(self.cells.shape[0])

    def num_vertices(self):
        return int(self.coords.shape[0])

    def num_edges(self):
        return int(self.edges.shape[0])

    def coordinates(self):
        return self.coords

    def edge_midpoints(self):
        return 0.5 * (self.coords[self.edges[:, 0]] + self.coords[self.edges[:, 1]])

    def edge_normals(self, edge_ids):
        """Unit normals of the given edges pointing away from their adjacent cell
        (outward for boundary edges)."""
        e = self.edges[edge_ids].astype(np.int64)
        t = self.coords[e[:, 1]] - self.coords[e[:, 0]]
        n = np.stack([t[:, 1], -t[:, 0]], axis=1)
        n /= np.linalg.norm(n, axis=1)[:, None]
        centroid = self.coords[self.cells[self.edge_cell[edge_ids]].astype(np.int64)].mean(axis=1)
        mid = 0.5 * (self.coords[e[:, 0]] + self.coords[e[:, 1]])
        flip = ((centroid - mid) * n).sum(axis=1) > 0.0
        n[flip] *= -1.0
        return n

    def hmin(self):
        e = self.coords[self.edges[:, 1]] - self.coords[self.edges[:, 0]]
        return float(np.sqrt((e * e).sum(axis=1)).min())


class FacetMarkers:
    """``MeshFunction("size_t", mesh, dim - 1)`` stand-in: one id per edge."""

    def __init__(self, mesh, value=0):
        self.mesh = mesh
        self.values = np.full(mesh.num_edges(), value, dtype=np.int64)

    def dim(self):
        return 1

    def set_all(self, value):
        self.values[:] = value

    def array(self):
        return self.values

    def mark(self, predicate, value, boundary_only=True):
        """Mark every facet whose two vertices and midpoint satisfy
        ``predicate(x) -> bool array`` (dolfin SubDomain.mark semantics with
        check_midpoint=True); ``on_boundary`` is honoured via ``boundary_only``."""
        m = self.mesh
        ok = predicate(m.coords[m.edges[:, 0]]) & predicate(m.coords[m.edges[:, 1]]) \
            & predicate(m.edge_midpoints())
        if boundary_only:
            ok &= m.edge_on_boundary
        self.values[ok] = value

    def ids(self, boundary_only=True):
        v = self.values[self.mesh.edge_on_boundary] if boundary_only else self.values
        return set(int(i) for i in np.unique(v))

    def facets_with_id(self, marker_id):
        return np.nonzero(self.values == marker_id)[0]


class TaylorHoodDofMap:
    """Scalar P2 and P1 node numbering + cell dof maps.

    Local P2 order (v0, v1, v2, e(v1v2), e(v0v2), e(v0v1)).  Scalar P2 nodes are
    renumbered lexicographically by (y, x) so that vertex and edge nodes of one
    neighbourhood sit close together in memory (x-gather locality of the SpMV
    kernels; on the structured meshes of grid_generator this is the natural
    (2nx+1) x (2ny+1) lattice order).  Velocity dofs are node-interleaved
    (2 * node + component); mixed vectors are [velocity | pressure].

    The reference's dolfin numbering is graph-reordered and not reproducible
    (SURVEY.md R4): fields are compared by coordinates, never by index.
    """

    def __init__(self, mesh, reorder=True, periodic_map=None):
        self.mesh = mesh
        nv, ne = mesh.num_vertices(), mesh.num_edges()
        xy = np.concatenate([mesh.coords, mesh.edge_midpoints()], axis=0)      # entity order
        n_ent = nv + ne
        if reorder:
            scale = 1.0 / max(mesh.hmin(), 1e-300)
            q = np.round(xy * (4.0 * scale)).astype(np.int64)      # robust lexicographic key
            order = np.lexsort((q[:, 0], q[:, 1]))
            ent_to_node = np.empty(n_ent, dtype=np.int64)
            ent_to_node[order] = np.arange(n_ent)
        else:
            ent_to_node = np.arange(n_ent, dtype=np.int64)
        p1_node = np.arange(nv, dtype=np.int64)
        if periodic_map is not None:
            # periodic_map: (p2_master_of_entity [n_ent], p1_master_of_vertex [nv]); slaves
            # take the master's node, then the ids are compacted.
            m2, m1 = periodic_map
            ent_to_node = _compact(ent_to_node[m2])
            p1_node = _compact(p1_node[m1])
        self.vertex_node = ent_to_node[:nv]
        self.edge_node = ent_to_node[nv:]
        self.n_p2 = int(ent_to_node.max()) + 1
        self.n_p1 = int(p1_node.max()) + 1
        c = mesh.cells.astype(np.int64)
        self.p2_dofmap = np.ascontiguousarray(
            np.concatenate([self.vertex_node[c], self.edge_node[mesh.cell_edges]], axis=1),
            dtype=np.int32)
        self.p1_vertex_node = p1_node
        self.p1_dofmap = np.ascontiguousarray(p1_node[c], dtype=np.int32)
        # node coordinates
        self.p2_coords = np.zeros((self.n_p2, 2))
        self.p2_coords[ent_to_node] = xy          # slaves written first or last: same set per node
        if periodic_map is not None:
            masters = np.nonzero(periodic_map[0] == np.arange(n_ent))[0]
            self.p2_coords[ent_to_node[masters]] = xy[masters]
        self.p1_coords = np.zeros((self.n_p1, 2))
        self.p1_coords[p1_node] = mesh.coords
        if periodic_map is not None:
            mv = np.nonzero(periodic_map[1] == np.arange(nv))[0]
            self.p1_coords[p1_node[mv]] = mesh.coords[mv]
        self.n_velocity = 2 * self.n_p2
        self.n_dofs = self.n_velocity + self.n_p1

    # -- Dirichlet dof selection (topological, as dolfin's default) -------------
    def facet_p2_nodes(self, facet_ids):
        """[nf, 3] scalar P2 node ids (end, end, midpoint) of the given edges."""
        e = self.mesh.edges[facet_ids].astype(np.int64)
        return np.stack([self.vertex_node[e[:, 0]], self.vertex_node[e[:, 1]],
                         self.edge_node[facet_ids]], axis=1)

    def facet_p1_nodes(self, facet_ids):
        e = self.mesh.edges[facet_ids].astype(np.int64)
        return self.p1_vertex_node[e]


def _compact(ids):
    u, inv = np.unique(ids, return_inverse=True)
    return inv.astype(np.int64)


def rectangle_mesh(p0, p1, nx, ny):
    """dolfin.RectangleMesh(p0, p1, nx, ny) with the default "right" diagonal:
    vertex id = iy * (nx + 1) + ix, quad -> (v0, v1, v3), (v0, v2, v3)."""
    x = np.linspace(p0[0], p1[0], nx + 1)
    y = np.linspace(p0[1], p1[1], ny + 1)
    X, Y = np.meshgrid(x, y, indexing="xy")
    coords = np.stack([X.ravel(), Y.ravel()], axis=1)
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    v0 = (iy * (nx + 1) + ix).ravel()
    v1 = v0 + 1
    v2 = v0 + (nx + 1)
    v3 = v2 + 1
    cells = np.empty((2 * nx * ny, 3), dtype=np.int32)
    cells[0::2] = np.stack([v0, v1, v3], axis=1)
    cells[1::2] = np.stack([v0, v2, v3], axis=1)
    mesh = Mesh(coords, cells)
    mesh.structured = (tuple(p0), tuple(p1), int(nx), int(ny))   # enables the multigrid hierarchy
    return mesh


def periodic_entity_map(mesh, domain):
    """Master entity of every vertex and edge under a dolfin-style periodic ``SubDomain``
    (``inside(x, on_boundary)`` marks the master part of the boundary, ``map(x_slave, x_master)``
    sends a slave point to its master; reference usage: tests/test_transient_solvers.py:20-46,
    source/ns_solver_base.py:516-518).  Returns (entity_master [nv + ne], vertex_master [nv]):
    indices into the entity list (vertices, then edges)."""
    nv, ne = mesh.num_vertices(), mesh.num_edges()
    pts = np.concatenate([mesh.coords, mesh.edge_midpoints()], axis=0)
    on_bnd = np.zeros(nv + ne, dtype=bool)
    bedges = np.nonzero(mesh.edge_on_boundary)[0]
    on_bnd[nv + bedges] = True
    on_bnd[mesh.edges[bedges].ravel()] = True
    scale = 1.0 / max(mesh.hmin(), 1e-300)

    def key(p):
        return (int(round(p[0] * 8.0 * scale)), int(round(p[1] * 8.0 * scale)))

    is_master = np.zeros(nv + ne, dtype=bool)
    for i in np.nonzero(on_bnd)[0]:
        is_master[i] = bool(domain.inside(pts[i], True))
    lookup_v = {key(pts[i]): i for i in range(nv) if is_master[i]}
    lookup_e = {key(pts[i]): i for i in range(nv, nv + ne) if is_master[i]}
    master = np.arange(nv + ne, dtype=np.int64)
    # every boundary entity (masters included: a corner such as (0, 1) lies on the master edge
    # x = 0 but is itself the image of (0, 0) under the y-shift) is sent through ``map``; chains
    # are followed to their root so that all periodic images of a point share one dof.  [The
    # reference's PeriodicDomain does not treat the corner explicitly; how dolfin resolves that
    # chain cannot be checked here -- the mathematically periodic space is built.]
    for i in np.nonzero(on_bnd)[0]:
        target = np.array([np.nan, np.nan])
        domain.map(pts[i].copy(), target)
        if not np.all(np.isfinite(target)):
            continue
        j = (lookup_v if i < nv else lookup_e).get(key(target))
        if j is not None and j != i:
            master[i] = j
    for i in np.nonzero(on_bnd)[0]:
        root, hops = master[i], 0
        while master[root] != root and hops < 8:
            root, hops = master[root], hops + 1
        master[i] = root
    return master, master[:nv].copy()
