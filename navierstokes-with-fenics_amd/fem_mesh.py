"""Dolfin-free mesh / facet-marker / Taylor-Hood dof-map containers.

These are the objects that cross the solver boundary in the reference
(``dlfn.Mesh`` and ``dlfn.MeshFunction("size_t", mesh, dim-1)``, consumed at
source/ns_solver_base.py:78-95).  Only what the hot path needs is provided:
vertex coordinates, cell->vertex connectivity, the edge (facet) entities with
their markers, and the P2 / P1 cell dof maps handed to the C-ABI
(include/nsfem.h: nsfem_mesh_desc).

Vertex order, "right" diagonal / Kuhn split and marker ids follow dolfin's RectangleMesh and
BoxMesh as used by source/grid_generator.py:111-208.
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
    """Simplex mesh: triangles (``coords`` [nv, 2], ``cells`` [nc, 3]) or tetrahedra
    (``coords`` [nv, 3], ``cells`` [nc, 4]).  Entities in UFC local order:
    triangle edges e0 = (v1, v2), e1 = (v0, v2), e2 = (v0, v1);
    tetrahedron edges e0 = (v2, v3), e1 = (v1, v3), e2 = (v1, v2), e3 = (v0, v3), e4 = (v0, v2),
    e5 = (v0, v1) and faces f_i opposite v_i.  ``facets`` are the codimension-1 entities (edges
    in 2D, faces in 3D) that carry the boundary markers."""

    _TRI_EDGES = ((1, 2), (0, 2), (0, 1))
    _TET_EDGES = ((2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1))
    _TET_FACES = ((1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2))

    def __init__(self, coords, cells):
        self.coords = np.ascontiguousarray(coords, dtype=np.float64)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        assert self.coords.ndim == 2 and self.coords.shape[1] in (2, 3), "2D / 3D simplex meshes only"
        self._dim = dim = int(self.coords.shape[1])
        assert self.cells.ndim == 2 and self.cells.shape[1] == dim + 1
        # the edge and facet entities are built on first use (__getattr__): the coarse levels of a multigrid hierarchy
        # and the strips / slabs of a partition never ask for them (13 M-dof channel: 1.9 of 2.9 s of mesh set-up)

    _EDGE_ATTRS = ("cell_edges", "edges", "edge_on_boundary", "edge_cell")
    _FACET_ATTRS = ("facets", "cell_facets", "facet_on_boundary", "facet_cell", "facet_edges")

    def __getattr__(self, name):
        # (only reached when the attribute does not exist yet)
        if name in Mesh._EDGE_ATTRS or (name in Mesh._FACET_ATTRS and self.__dict__.get("_dim") == 2):
            self._build_edges()
        elif name in Mesh._FACET_ATTRS:
            self._build_facets()
        if name in self.__dict__:
            return self.__dict__[name]
        raise AttributeError(name)

    def _build_edges(self):
        if "edges" in self.__dict__ and "cell_edges" in self.__dict__:
            return
        dim = self._dim
        box = self.__dict__.get("_box_shape")
        if dim == 3 and box is not None:       # box_mesh: the same arrays in closed form, no sort
            edges, cell_edges = _box_edges(box[0], box[1], box[2], self.cells)
            self.edges, self.cell_edges = edges, cell_edges
            return
        nv = self.coords.shape[0]
        c = self.cells.astype(np.int64)
        nc = c.shape[0]
        local_edges = self._TRI_EDGES if dim == 2 else self._TET_EDGES
        pairs = np.stack([c[:, list(e)] for e in local_edges], axis=1)          # [nc, n_le, 2]
        key = pairs.min(axis=2) * nv + pairs.max(axis=2)
        ukey, inv, counts = np.unique(key.ravel(), return_inverse=True, return_counts=True)
        self.cell_edges = inv.reshape(nc, len(local_edges)).astype(np.int32)
        self.edges = np.stack([ukey // nv, ukey % nv], axis=1).astype(np.int32)     # [ne, 2]
        if dim == 2:
            self.edge_on_boundary = counts == 1
            # one adjacent cell per edge (the only one for boundary edges)
            self.edge_cell = np.empty(self.edges.shape[0], dtype=np.int32)
            self.edge_cell[self.cell_edges.ravel()] = np.repeat(np.arange(nc, dtype=np.int32), 3)
            self.facets = self.edges
            self.cell_facets = self.cell_edges
            self.facet_on_boundary = self.edge_on_boundary
            self.facet_cell = self.edge_cell

    def _build_facets(self):
        if "facets" in self.__dict__:
            return
        self._build_edges()
        nv = self.coords.shape[0]
        c = self.cells.astype(np.int64)
        nc = c.shape[0]
        tri = np.sort(np.stack([c[:, list(f)] for f in self._TET_FACES], axis=1), axis=2)   # [nc, 4, 3]
        fkey = (tri[:, :, 0] * nv + tri[:, :, 1]) * nv + tri[:, :, 2]
        ufk, finv, fcounts = np.unique(fkey.ravel(), return_inverse=True, return_counts=True)
        self.cell_facets = finv.reshape(nc, 4).astype(np.int32)
        self.facets = np.stack([ufk // (nv * nv), (ufk // nv) % nv, ufk % nv], axis=1).astype(np.int32)
        self.facet_on_boundary = fcounts == 1
        self.facet_cell = np.empty(self.facets.shape[0], dtype=np.int32)
        self.facet_cell[self.cell_facets.ravel()] = np.repeat(np.arange(nc, dtype=np.int32), 4)
        # the three edges of every face: (b, c), (a, c), (a, b) for the sorted face (a, b, c)
        f = self.facets.astype(np.int64)
        ekey = self.edges[:, 0].astype(np.int64) * nv + self.edges[:, 1]
        self.facet_edges = np.stack(
            [np.searchsorted(ekey, f[:, i] * nv + f[:, j]) for i, j in ((1, 2), (0, 2), (0, 1))],
            axis=1).astype(np.int32)

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

    def num_facets(self):
        return int(self.facets.shape[0])

    def facet_midpoints(self):
        return self.coords[self.facets.astype(np.int64)].mean(axis=1)

    def facet_normals(self, facet_ids):
        """Unit normals of the given facets pointing away from their adjacent cell (outward for
        boundary facets)."""
        if self._dim == 2:
            return self.edge_normals(facet_ids)
        f = self.facets[facet_ids].astype(np.int64)
        a, b, c = self.coords[f[:, 0]], self.coords[f[:, 1]], self.coords[f[:, 2]]
        n = np.cross(b - a, c - a)
        n /= np.linalg.norm(n, axis=1)[:, None]
        centroid = self.coords[self.cells[self.facet_cell[facet_ids]].astype(np.int64)].mean(axis=1)
        flip = ((centroid - (a + b + c) / 3.0) * n).sum(axis=1) > 0.0
        n[flip] *= -1.0
        return n

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

    def facet_cell_local(self, facet_ids):
        """(adjacent cell, UFC local facet number = index of the opposite vertex) of the given
        facets -- the arrays nsfem_boundary_force takes (include/nsfem.h)"""
        facet_ids = np.asarray(facet_ids, dtype=np.int64)
        cells = self.facet_cell[facet_ids].astype(np.int64)
        local = np.argmax(self.cell_facets[cells] == facet_ids[:, None], axis=1)
        assert np.all(self.cell_facets[cells, local] == facet_ids)
        return cells.astype(np.int32), local.astype(np.int32)

    def hmin(self):
        e = self.coords[self.edges[:, 1]] - self.coords[self.edges[:, 0]]
        return float(np.sqrt((e * e).sum(axis=1)).min())


class FacetMarkers:
    """``MeshFunction("size_t", mesh, dim - 1)`` stand-in: one id per facet (edge / face)."""

    def __init__(self, mesh, value=0):
        self.mesh = mesh
        self.values = np.full(mesh.num_facets(), value, dtype=np.int64)

    def dim(self):
        return self.mesh._dim - 1

    def set_all(self, value):
        self.values[:] = value

    def array(self):
        return self.values

    def mark(self, predicate, value, boundary_only=True):
        """Mark every facet whose vertices and midpoint satisfy
        ``predicate(x) -> bool array`` (dolfin SubDomain.mark semantics with
        check_midpoint=True); ``on_boundary`` is honoured via ``boundary_only``."""
        m = self.mesh
        ok = predicate(m.facet_midpoints())
        for k in range(m.facets.shape[1]):
            ok &= predicate(m.coords[m.facets[:, k]])
        if boundary_only:
            ok &= m.facet_on_boundary
        self.values[ok] = value

    def ids(self, boundary_only=True):
        v = self.values[self.mesh.facet_on_boundary] if boundary_only else self.values
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

    @staticmethod
    def _lattice_lex(mesh, xy, n_ent):
        """lexicographic node ids (x fastest, last axis slowest) of the entities of a structured mesh from their
        half-lattice positions, or None when the mesh is not a fully occupied half lattice"""
        lattice = getattr(mesh, "structured", None)
        if lattice is None:
            return None
        dim = mesh._dim
        lo = np.asarray(lattice[0], dtype=np.float64)
        hi = np.asarray(lattice[1], dtype=np.float64)
        cells = np.asarray(lattice[2:2 + dim], dtype=np.int64)
        width = 2 * cells + 1
        if int(np.prod(width)) != n_ent:
            return None
        f = (xy - lo) * (2.0 * cells / (hi - lo))
        q = np.round(f).astype(np.int64)
        if np.abs(f - q).max() > 1e-6 or (q < 0).any() or (q >= width).any():
            return None
        ids = q[:, dim - 1]
        for a in range(dim - 2, -1, -1):
            ids = ids * width[a] + q[:, a]
        if np.bincount(ids, minlength=n_ent).max() != 1:
            return None
        return ids

    def __init__(self, mesh, reorder=True, periodic_map=None, class_key=None):
        """reorder: True / "lex" lexicographic lattice order (strip / slab partitions rely on it:
        halos are contiguous ranges); "parity" = structured meshes only (``mesh.structured``):
        nodes grouped by the parity class of their half-lattice index (vertices, then the edge
        types), lexicographic inside a class -- consecutive rows of every P2 operator then have
        equal length and their k-th columns are consecutive ids, which is what the SELL-64 SpMV
        kernel (csrc/linalg.hip: k_spmv_sell) needs for a fully coalesced x gather; falls back to
        "lex" on general meshes; False keeps the entity order.
        class_key (lexicographic order only): one integer per entity (vertices, then edges), the
        slowest sort key -- unstructured partitions number interior nodes first, then the nodes
        next to ghosts, then the ghosts, so that the rows which can run under a halo exchange form
        one contiguous range."""
        self.mesh = mesh
        nv, ne = mesh.num_vertices(), mesh.num_edges()
        xy = np.concatenate([mesh.coords, mesh.edge_midpoints()], axis=0)      # entity order
        n_ent = nv + ne
        dim = self.dim = mesh._dim
        lattice = getattr(mesh, "structured", None) if reorder == "parity" else None
        self.ordering = "parity" if lattice is not None else ("lex" if reorder else "entity")
        if lattice is not None:
            lo = np.asarray(lattice[0], dtype=np.float64)
            hi = np.asarray(lattice[1], dtype=np.float64)
            cells = np.asarray(lattice[2:], dtype=np.float64)
            q = np.round((xy - lo) * (2.0 * cells / (hi - lo))).astype(np.int64)   # half-lattice index
            cls = np.zeros(n_ent, dtype=np.int64)
            for a in range(dim):
                cls |= (q[:, a] & 1) << a
            # slabs of `block` half-lattice planes along the slowest axis, the classes inside a slab:
            # a contiguous row range is a spatial slab again (each XCD's L2 then holds the few
            # planes of x its rows touch), yet 64 consecutive rows still belong to one class
            block = self.parity_block = int(getattr(mesh, "parity_block", 4 if dim == 3 else 16))
            slab = q[:, dim - 1] // block
            # 3D, wide cross-sections: the class passes over a slab touch ~8 lattice planes of x; when those
            # exceed an XCD's 4 MiB L2 (257 x 129 nodes x 24 B = 0.8 MB per plane: channel n = 64) every pass
            # re-fetches them (PMC: 1.9 x the algorithmic bytes).  Blocks of `block_y` lattice lines inside a slab
            # bound the working set; 0 = off.  NSFEM_PARITY_BLOCK_Y overrides the automatic choice.
            import os
            block_y = 0
            if dim == 3:
                plane_bytes = float(2 * cells[0] + 1) * float(2 * cells[1] + 1) * 24.0
                env = os.environ.get("NSFEM_PARITY_BLOCK_Y")
                if env is not None:
                    block_y = int(env)
                elif 8.0 * plane_bytes > 3.0e6:
                    block_y = 32
            self.parity_block_y = block_y
            # one composite integer key (slab slowest, then the y block, the class, the lattice position with x
            # fastest) and ONE sort instead of a five-key lexsort: the keys are distinct, the order is the same
            width = (2 * cells + 1).astype(np.int64)
            key = slab
            if block_y > 0:
                key = key * (int(width[1]) // block_y + 1) + q[:, 1] // block_y
            key = key * (1 << dim) + cls
            for a_ in range(dim - 1, -1, -1):
                key = key * int(width[a_]) + q[:, a_]
            order = np.argsort(key, kind="stable")
            ent_to_node = np.empty(n_ent, dtype=np.int64)
            ent_to_node[order] = np.arange(n_ent)
        elif reorder and getattr(mesh, "structured", None) is None and class_key is None and \
                _unstructured_order() in ("morton", "rcm"):
            # unstructured meshes, locality experiments (NSFEM_P2_ORDER_UNSTRUCTURED = morton | rcm; default lex):
            # Z-order curve of the node coordinates, or reverse Cuthill-McKee of the P2 node graph
            self.ordering = _unstructured_order()
            if self.ordering == "morton":
                lo, hi = xy.min(axis=0), xy.max(axis=0)
                q = np.clip(((xy - lo) / np.maximum(hi - lo, 1e-300) * (2 ** 20 - 1)).astype(np.int64), 0, 2 ** 20 - 1)
                code = np.zeros(n_ent, dtype=np.int64)
                for bit in range(20):
                    for a in range(dim):
                        code |= ((q[:, a] >> bit) & 1) << (dim * bit + a)
                order = np.argsort(code, kind="stable")
            else:
                import scipy.sparse as sp
                from scipy.sparse.csgraph import reverse_cuthill_mckee
                ents = np.concatenate([mesh.cells.astype(np.int64), nv + mesh.cell_edges.astype(np.int64)], axis=1)
                k = ents.shape[1]
                rows = np.repeat(ents, k, axis=1).ravel()
                cols = np.tile(ents, (1, k)).ravel()
                graph = sp.csr_matrix((np.ones(rows.size, dtype=np.int8), (rows, cols)), shape=(n_ent, n_ent))
                order = np.asarray(reverse_cuthill_mckee(graph, symmetric_mode=True), dtype=np.int64)
            ent_to_node = np.empty(n_ent, dtype=np.int64)
            ent_to_node[order] = np.arange(n_ent)
        elif reorder and class_key is None and self._lattice_lex(mesh, xy, n_ent) is not None:
            # structured meshes: every point of the half lattice carries exactly one P2 node, its lexicographic
            # rank is its lattice position -- no sort (13 M-dof channel: 0.6 s of set-up)
            ent_to_node = self._lattice_lex(mesh, xy, n_ent)
        elif reorder:
            scale = 1.0 / max(mesh.hmin(), 1e-300)
            q = np.round(xy * (4.0 * scale)).astype(np.int64)      # robust lexicographic key
            keys = tuple(q[:, k] for k in range(dim))                # x fastest, last axis slowest
            if class_key is not None:
                keys = keys + (np.asarray(class_key, dtype=np.int64),)
            order = np.lexsort(keys)
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
        self.p2_coords = np.zeros((self.n_p2, dim))
        self.p2_coords[ent_to_node] = xy          # slaves written first or last: same set per node
        if periodic_map is not None:
            masters = np.nonzero(periodic_map[0] == np.arange(n_ent))[0]
            self.p2_coords[ent_to_node[masters]] = xy[masters]
        self.p1_coords = np.zeros((self.n_p1, dim))
        self.p1_coords[p1_node] = mesh.coords
        if periodic_map is not None:
            mv = np.nonzero(periodic_map[1] == np.arange(nv))[0]
            self.p1_coords[p1_node[mv]] = mesh.coords[mv]
        self.n_velocity = dim * self.n_p2
        self.n_dofs = self.n_velocity + self.n_p1

    # -- Dirichlet dof selection (topological, as dolfin's default) -------------
    def facet_p2_nodes(self, facet_ids):
        """scalar P2 node ids of the given facets: 2D [nf, 3] (end, end, midpoint); 3D [nf, 6]
        (three vertices, then the midpoints of the edges opposite to them within the face)."""
        f = self.mesh.facets[facet_ids].astype(np.int64)
        if self.dim == 2:
            return np.stack([self.vertex_node[f[:, 0]], self.vertex_node[f[:, 1]],
                             self.edge_node[facet_ids]], axis=1)
        return np.concatenate([self.vertex_node[f], self.edge_node[self.mesh.facet_edges[facet_ids]]],
                              axis=1)

    def facet_p1_nodes(self, facet_ids):
        f = self.mesh.facets[facet_ids].astype(np.int64)
        return self.p1_vertex_node[f]


def _unstructured_order():
    import os
    return os.environ.get("NSFEM_P2_ORDER_UNSTRUCTURED", "lex")


def preferred_p2_order(dim):
    """numbering of the scalar P2 nodes on structured single-GPU meshes: lexicographic in 2D,
    parity classes in 3D (pairs with the SELL-64 SpMV kernel: measured 6 % faster smoothing
    launches and 7 % faster time steps than lexicographic + CSR-stream on tetrahedra, 15 % slower
    on triangles); the environment variable NSFEM_P2_ORDER = lex | parity overrides it"""
    import os
    choice = os.environ.get("NSFEM_P2_ORDER", "parity" if dim == 3 else "lex")
    return "parity" if choice == "parity" else True


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


def box_mesh(p0, p1, nx, ny, nz):
    """dolfin.BoxMesh(p0, p1, nx, ny, nz): vertex id = (iz (ny+1) + iy)(nx+1) + ix, six
    tetrahedra per cube sharing the diagonal (0,0,0)-(1,1,1) (Kuhn / Freudenthal split), so the
    mesh of a uniformly refined grid is the refinement of the coarse mesh (nested P1 spaces)."""
    x = np.linspace(p0[0], p1[0], nx + 1)
    y = np.linspace(p0[1], p1[1], ny + 1)
    z = np.linspace(p0[2], p1[2], nz + 1)
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    iz, iy, ix = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    sx, sy, sz = 1, nx + 1, (nx + 1) * (ny + 1)
    v0 = (iz * sz + iy * sy + ix).ravel()
    v = [v0 + bx * sx + by * sy + bz * sz for bz in (0, 1) for by in (0, 1) for bx in (0, 1)]
    # v[k]: bit 0 = x, bit 1 = y, bit 2 = z
    tets = ((0, 1, 3, 7), (0, 1, 7, 5), (0, 5, 7, 4), (0, 3, 2, 7), (0, 6, 4, 7), (0, 2, 6, 7))
    assert coords.shape[0] < 2 ** 31
    cells = np.empty((6 * nx * ny * nz, 4), dtype=np.int32)
    # positive orientation: every cube is a translate of the first one, so the sign of a tetrahedron's volume
    # depends on its type only -- decided on the first cube
    first = coords[[bx * sx + by * sy + bz * sz for bz in (0, 1) for by in (0, 1) for bx in (0, 1)]]
    for k, t in enumerate(tets):
        a = first[list(t)]
        det = np.dot(a[1] - a[0], np.cross(a[2] - a[0], a[3] - a[0]))
        order = t if det > 0 else (t[0], t[2], t[1], t[3])
        for j, i in enumerate(order):
            cells[k::6, j] = v[i]
    mesh = Mesh(coords, cells)
    mesh.structured = (tuple(p0), tuple(p1), int(nx), int(ny), int(nz))
    mesh._box_shape = (int(nx), int(ny), int(nz))        # (edges in closed form on first use: Mesh._build_edges)
    return mesh


def _box_edges(nx, ny, nz, cells):
    """(edges, cell_edges) of box_mesh in closed form, the arrays Mesh._build_edges gets from np.unique: edges sorted
    by (lower vertex, upper vertex).  From a vertex seven edges lead to higher vertices -- along x, y, the xy diagonal,
    z, the xz and yz diagonals and the body diagonal, in ascending order of the other end -- as far as the box
    allows; the edge's number is the count of such edges at lower vertices plus its rank at its own."""
    sy, sz = nx + 1, (nx + 1) * (ny + 1)
    nvert = sz * (nz + 1)
    dirs = ((1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 1))
    off = np.array([dx + dy * sy + dz * sz for dx, dy, dz in dirs], dtype=np.int64)
    vid = np.arange(nvert, dtype=np.int64)
    ix, iy, iz = vid % sy, (vid // sy) % (ny + 1), vid // sz
    valid = np.stack([(ix + dx <= nx) & (iy + dy <= ny) & (iz + dz <= nz) for dx, dy, dz in dirs], axis=1)
    rank = np.cumsum(valid, axis=1, dtype=np.int32) - valid                     # valid directions before this one
    count = valid.sum(axis=1, dtype=np.int64)
    first = np.concatenate([[0], np.cumsum(count)[:-1]])
    vv, dd = np.nonzero(valid)
    edges = np.stack([vv, vv + off[dd]], axis=1).astype(np.int32)
    cell_edges = np.empty((cells.shape[0], 6), dtype=np.int32)
    c = cells.astype(np.int64)
    for k, (a, b) in enumerate(Mesh._TET_EDGES):
        lo, hi = np.minimum(c[:, a], c[:, b]), np.maximum(c[:, a], c[:, b])
        d = np.searchsorted(off, hi - lo)
        assert np.array_equal(off[d], hi - lo)
        cell_edges[:, k] = first[lo] + rank[lo, d]
    return edges, cell_edges


def periodic_entity_map(mesh, domain):
    """Master entity of every vertex and edge under a dolfin-style periodic ``SubDomain``
    (``inside(x, on_boundary)`` marks the master part of the boundary, ``map(x_slave, x_master)``
    sends a slave point to its master; reference usage: tests/test_transient_solvers.py:20-46,
    source/ns_solver_base.py:516-518).  Returns (entity_master [nv + ne], vertex_master [nv]):
    indices into the entity list (vertices, then edges)."""
    nv, ne = mesh.num_vertices(), mesh.num_edges()
    pts = np.concatenate([mesh.coords, mesh.edge_midpoints()], axis=0)
    on_bnd = np.zeros(nv + ne, dtype=bool)
    if mesh._dim == 2:
        bedges = np.nonzero(mesh.edge_on_boundary)[0]
    else:                                       # edges of the boundary faces
        bedges = np.unique(mesh.facet_edges[mesh.facet_on_boundary])
    on_bnd[nv + bedges] = True
    on_bnd[mesh.edges[bedges].ravel()] = True
    scale = 1.0 / max(mesh.hmin(), 1e-300)

    def key(p):
        return tuple(int(round(v * 8.0 * scale)) for v in p)

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
        target = np.full(pts.shape[1], np.nan)
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
