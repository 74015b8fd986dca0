"""Element-partition domain decomposition of structured meshes over the GPUs of one node.

New functionality (the reference runs serially): the mesh is cut into
strips of cell rows, one per rank.  Rank r owns the dofs of its cell rows except its bottom
lattice line (owned by rank r-1) and additionally assembles ONE ghost row of cells above its
strip, so that every matrix row of an owned dof is complete without any assembly
communication.  Because all node numberings are lexicographic, every halo is a contiguous
range of the local vectors:

    send up   : the top owned line            -> bottom ghost line of rank r+1
    send down : the first owned line(s)       -> top ghost line(s) of rank r-1
                (two P2 lattice lines, one P1 line)

Dot products run over owned dofs only (ghost entries of the Krylov vectors are kept zero by
the ghost row mask of the SpMV kernels); the per-block partial sums are all-reduced.
The same partition is applied to every multigrid level (own rows / 2^l + one coarse ghost
row); the coarsest level is solved redundantly on every rank from an all-reduced right-hand
side.
"""
import numpy as np

from fem_mesh import Mesh, TaylorHoodDofMap
from multigrid import structured_prolongation

GHOST = 2     # row-mask value of ghost dofs (1 = Dirichlet)


class StripLevel:
    """One P1 level of one rank: local mesh (own rows + ghost row), ghost flags, halo ranges."""

    def __init__(self, p0, p1, nx, ny, row0, own_rows, ghost_rows):
        hx = (p1[0] - p0[0])
        x = np.linspace(p0[0], p1[0], nx + 1)
        y_all = np.linspace(p0[1], p1[1], ny + 1)
        rows = own_rows + ghost_rows
        y = y_all[row0: row0 + rows + 1]
        X, Y = np.meshgrid(x, y, indexing="xy")
        coords = np.stack([X.ravel(), Y.ravel()], axis=1)
        ix, iy = np.meshgrid(np.arange(nx), np.arange(rows), indexing="xy")
        v0 = (iy * (nx + 1) + ix).ravel()
        v1, v2 = v0 + 1, v0 + (nx + 1)
        v3 = v2 + 1
        cells = np.empty((2 * nx * rows, 3), dtype=np.int32)
        cells[0::2] = np.stack([v0, v1, v3], axis=1)
        cells[1::2] = np.stack([v0, v2, v3], axis=1)
        self.mesh = Mesh(coords, cells)
        self.nx, self.rows, self.own_rows, self.row0 = nx, rows, own_rows, row0
        self.has_below, self.has_above = row0 > 0, ghost_rows > 0
        w1 = nx + 1
        self.w1 = w1
        self.n_p1 = w1 * (rows + 1)
        ghost = np.zeros(self.n_p1, dtype=np.uint8)
        if self.has_below:
            ghost[:w1] = GHOST
        if self.has_above:
            ghost[w1 * (own_rows + 1):] = GHOST
        self.p1_ghost = ghost
        # halo ranges in P1 node units: (offset, count)
        self.p1_halo = dict(
            send_up=(w1 * own_rows, w1) if self.has_above else (0, 0),
            recv_above=(w1 * (own_rows + 1), w1) if self.has_above else (0, 0),
            send_down=(w1, w1) if self.has_below else (0, 0),
            recv_below=(0, w1) if self.has_below else (0, 0))
        del hx

    def global_p1_offset(self):
        """global node id of local P1 node 0 on this level (lexicographic lines)."""
        return self.row0 * self.w1


class StripPartition:
    """Everything rank ``rank`` of ``size`` needs: local fine mesh + dof map, ghost masks,
    halo ranges, local multigrid levels with prolongations, and the replicated coarsest mesh."""

    def __init__(self, p0, p1, nx, ny, rank, size, coarsest=8, global_coarsest=None, min_rows=1):
        assert ny % size == 0, "cell rows must divide evenly over the ranks"
        own = ny // size
        self.rank, self.size = rank, size
        self.p0, self.p1, self.nx, self.ny = tuple(p0), tuple(p1), nx, ny
        g = 1 if rank < size - 1 else 0
        self.fine = StripLevel(p0, p1, nx, ny, rank * own, own, g)
        self.mesh = self.fine.mesh
        if size == 1:      # no halos to keep contiguous: the numbering is free (NSFEM_P2_ORDER)
            from fem_mesh import preferred_p2_order
            self.mesh.structured = (tuple(p0), tuple(p1), nx, ny)
        self.dofmap = TaylorHoodDofMap(self.mesh, reorder=preferred_p2_order(2) if size == 1 else True)
        dm = self.dofmap
        # P2 lattice lines: 2*rows + 1 lines of w2 nodes, lexicographic (dof map reorders so)
        w2 = 2 * nx + 1
        self.w2 = w2
        lines = 2 * self.fine.rows + 1
        assert dm.n_p2 == w2 * lines
        ghost2 = np.zeros(dm.n_p2, dtype=np.uint8)
        if self.fine.has_below:
            ghost2[:w2] = GHOST
        if self.fine.has_above:
            ghost2[w2 * (2 * own + 1):] = GHOST
        self.p2_ghost = ghost2
        self.p2_halo = dict(
            send_up=(w2 * 2 * own, w2) if self.fine.has_above else (0, 0),
            recv_above=(w2 * (2 * own + 1), 2 * w2) if self.fine.has_above else (0, 0),
            send_down=(w2, 2 * w2) if self.fine.has_below else (0, 0),
            recv_below=(0, w2) if self.fine.has_below else (0, 0))
        self.p1_ghost = self.fine.p1_ghost
        self.p1_halo = self.fine.p1_halo
        # global numbering of the local nodes (for gathering results / tests)
        self.p2_global = rank * own * 2 * w2 + np.arange(dm.n_p2)
        self.p1_global = rank * own * (nx + 1) + np.arange(dm.n_p1)
        self.p2_owned = ghost2 == 0
        self.p1_owned = self.p1_ghost == 0
        # multigrid levels: coarsen while every rank keeps >= 2 own rows and the global mesh
        # keeps >= coarsest cells per direction
        self.levels = []          # (StripLevel, prolongation csr to the finer level)
        lx, ly, lown, fine_level = nx, ny, own, self.fine
        # (``min_rows``: strips thinner than that are all halo -- every product of such a level is a
        # latency-bound message -- so the partitioned levels end there and the replicated hierarchy takes over)
        # (one rank: a level with an odd number of cells is followed by the NON-NESTED mesh of ceil(n / 2) cells,
        # multigrid.interpolation_prolongation -- 333 -> 167 -> 84 -> 42 -> 21; strips of several ranks need nested
        # levels for their halo lines and stop there)
        while True:
            even = lx % 2 == 0 and lown % 2 == 0
            if not even and not (size == 1 and min(lx, ly) >= 5):
                break
            cx, cy, cown = (lx + 1) // 2, (ly + 1) // 2, (lown + 1) // 2
            if cown < max(1, min_rows) or min(cx, cy) < coarsest:
                break
            lev = StripLevel(p0, p1, cx, cy, rank * cown, cown, g)
            if even:
                # prolongation for the (cx, lev.rows) -> (lx, 2 * lev.rows) refinement, truncated
                # to the vertex rows the finer local mesh actually has
                rowptr, col, val = structured_prolongation(lx, 2 * lev.rows)
                n_fine = (lx + 1) * (fine_level.rows + 1)
                rowptr = rowptr[: n_fine + 1].copy()
                nnz = rowptr[-1]
                col, val = col[:nnz].copy(), val[:nnz].copy()
            else:
                from multigrid import interpolation_prolongation
                rowptr, col, val = interpolation_prolongation(lx, ly, cx, cy)
            self.levels.append((lev, (rowptr, col, val)))
            lx, ly, lown, fine_level = cx, cy, cown, lev
        # replicated coarsest problem: the global mesh of the last level
        last = self.levels[-1][0] if self.levels else self.fine
        self.coarse_global_shape = (last.nx, ly)
        self.coarse_global_offset = last.global_p1_offset()
        # replicated hierarchy below the global coarse mesh (no halo exchanges on its levels):
        # lets the partitioned levels stop early (``coarsest`` large) without a huge dense solve
        self.global_tail = []
        if global_coarsest is not None:
            from multigrid import structured_hierarchy
            self.global_tail = structured_hierarchy(self.p0, self.p1, last.nx, ly,
                                                    coarsest=global_coarsest, allow_non_nested=False)

    def attach(self, ctx, degree=None, eig_ratio=None):
        """Ship the partition, the local multigrid levels and the replicated global coarsest
        mesh to a device context created on ``self.mesh`` (a communicator must already be
        attached when size > 1)."""
        n2g, n1g = global_dof_counts(self.nx, self.ny)
        ctx.set_partition(self.rank, self.size, self.p2_ghost, self.p1_ghost, self.p2_halo,
                          self.p1_halo, n2g, n1g)
        ctx.mg_prolongations = []                      # kept for attach_schur_laplacian(ctx, ..., part=self)
        for lev, (rowptr, col, val) in self.levels:
            ctx.mg_add_level(lev.mesh.coords, lev.mesh.cells, rowptr, col, val,
                             ghost=lev.p1_ghost, halo=lev.p1_halo)
            ctx.mg_prolongations.append((lev.n_p1, (rowptr, col, val)))
        cx, cy = self.coarse_global_shape
        from fem_mesh import rectangle_mesh
        cg = rectangle_mesh(self.p0, self.p1, cx, cy)
        ctx.mg_set_global_coarse(cg.coords, cg.cells, self.coarse_global_offset)
        for mesh, (rowptr, col, val) in self.global_tail:
            ctx.mg_add_global_level(mesh.coords, mesh.cells, rowptr, col, val)
        ctx.mg_finalize(2 if degree is None else degree, 4.0 if eig_ratio is None else eig_ratio)
        return len(self.levels)


def global_dof_counts(nx, ny, nz=None):
    if nz is None:
        return (2 * nx + 1) * (2 * ny + 1), (nx + 1) * (ny + 1)
    return (2 * nx + 1) * (2 * ny + 1) * (2 * nz + 1), (nx + 1) * (ny + 1) * (nz + 1)


# ---------------------------------------------------------------------------------------------
# 3D: slabs of cube layers along z (the slowest lexicographic index, so every halo is again one
# contiguous range: whole lattice planes)
# ---------------------------------------------------------------------------------------------
class SlabLevel:
    """One P1 level of one rank of a Kuhn box mesh: own cube layers + ghost layer above."""

    def __init__(self, p0, p1, nx, ny, nz, layer0, own_layers, ghost_layers):
        from fem_mesh import box_mesh
        layers = own_layers + ghost_layers
        hz = (p1[2] - p0[2]) / nz
        lo = (p0[0], p0[1], p0[2] + layer0 * hz)
        hi = (p1[0], p1[1], p0[2] + (layer0 + layers) * hz)
        self.mesh = box_mesh(lo, hi, nx, ny, layers)
        # exact global coordinates (avoid round-off differences between ranks)
        z_all = np.linspace(p0[2], p1[2], nz + 1)
        w1 = (nx + 1) * (ny + 1)
        self.mesh.coords[:, 2] = np.repeat(z_all[layer0: layer0 + layers + 1], w1)
        self.nx, self.ny, self.rows, self.own_rows, self.row0 = nx, ny, layers, own_layers, layer0
        self.has_below, self.has_above = layer0 > 0, ghost_layers > 0
        self.w1 = w1
        self.n_p1 = w1 * (layers + 1)
        ghost = np.zeros(self.n_p1, dtype=np.uint8)
        if self.has_below:
            ghost[:w1] = GHOST
        if self.has_above:
            ghost[w1 * (own_layers + 1):] = GHOST
        self.p1_ghost = ghost
        self.p1_halo = dict(
            send_up=(w1 * own_layers, w1) if self.has_above else (0, 0),
            recv_above=(w1 * (own_layers + 1), w1) if self.has_above else (0, 0),
            send_down=(w1, w1) if self.has_below else (0, 0),
            recv_below=(0, w1) if self.has_below else (0, 0))

    def global_p1_offset(self):
        return self.row0 * self.w1


class SlabPartition:
    """3D counterpart of ``StripPartition``: rank ``rank`` of ``size`` owns nz / size cube layers
    of the (nx, ny, nz) Kuhn box mesh (dofs of its layers except the bottom lattice plane) and
    assembles one ghost layer above.  Same attributes, same ``attach``."""

    def __init__(self, p0, p1, nx, ny, nz, rank, size, coarsest=8, global_coarsest=None):
        from multigrid import structured_hierarchy, structured_prolongation_3d
        assert nz % size == 0, "cube layers must divide evenly over the ranks"
        own = nz // size
        self.rank, self.size = rank, size
        self.p0, self.p1, self.nx, self.ny, self.nz = tuple(p0), tuple(p1), nx, ny, nz
        g = 1 if rank < size - 1 else 0
        self.fine = SlabLevel(p0, p1, nx, ny, nz, rank * own, own, g)
        self.mesh = self.fine.mesh
        if size == 1:      # no halos to keep contiguous: the numbering is free (NSFEM_P2_ORDER)
            from fem_mesh import preferred_p2_order
            self.mesh.structured = (tuple(p0), tuple(p1), nx, ny, nz)
        self.dofmap = dm = TaylorHoodDofMap(self.mesh, reorder=preferred_p2_order(3) if size == 1 else True)
        w2 = (2 * nx + 1) * (2 * ny + 1)
        self.w2 = w2
        planes = 2 * self.fine.rows + 1
        assert dm.n_p2 == w2 * planes
        ghost2 = np.zeros(dm.n_p2, dtype=np.uint8)
        if self.fine.has_below:
            ghost2[:w2] = GHOST
        if self.fine.has_above:
            ghost2[w2 * (2 * own + 1):] = GHOST
        self.p2_ghost = ghost2
        self.p2_halo = dict(
            send_up=(w2 * 2 * own, w2) if self.fine.has_above else (0, 0),
            recv_above=(w2 * (2 * own + 1), 2 * w2) if self.fine.has_above else (0, 0),
            send_down=(w2, 2 * w2) if self.fine.has_below else (0, 0),
            recv_below=(0, w2) if self.fine.has_below else (0, 0))
        self.p1_ghost = self.fine.p1_ghost
        self.p1_halo = self.fine.p1_halo
        self.p2_global = rank * own * 2 * w2 + np.arange(dm.n_p2)
        self.p1_global = rank * own * self.fine.w1 + np.arange(dm.n_p1)
        self.p2_owned = ghost2 == 0
        self.p1_owned = self.p1_ghost == 0
        self.levels = []
        lx, ly, lz, lown, fine_level = nx, ny, nz, own, self.fine
        # (one rank: odd levels continue through non-nested Kuhn meshes, multigrid.interpolation_prolongation_3d)
        while True:
            even = lx % 2 == 0 and ly % 2 == 0 and lown % 2 == 0
            if not even and not (size == 1 and min(lx, ly, lz) >= 5):
                break
            cx, cy, cz, cown = (lx + 1) // 2, (ly + 1) // 2, (lz + 1) // 2, (lown + 1) // 2
            if cown < 1 or min(cx, cy, cz) < coarsest:
                break
            lev = SlabLevel(p0, p1, cx, cy, cz, rank * cown, cown, g)
            if even:
                rowptr, col, val = structured_prolongation_3d(lx, ly, 2 * lev.rows)
                n_fine = (lx + 1) * (ly + 1) * (fine_level.rows + 1)
                rowptr = rowptr[: n_fine + 1].copy()
                nnz = rowptr[-1]
                col, val = col[:nnz].copy(), val[:nnz].copy()
            else:
                from multigrid import interpolation_prolongation_3d
                rowptr, col, val = interpolation_prolongation_3d(lx, ly, lz, cx, cy, cz)
            self.levels.append((lev, (rowptr, col, val)))
            lx, ly, lz, lown, fine_level = cx, cy, cz, cown, lev
        last = self.levels[-1][0] if self.levels else self.fine
        self.coarse_global_shape = (last.nx, last.ny, lz)
        self.coarse_global_offset = last.global_p1_offset()
        self.global_tail = []
        if global_coarsest is not None:
            self.global_tail = structured_hierarchy(self.p0, self.p1, last.nx, last.ny, lz,
                                                    coarsest=global_coarsest, allow_non_nested=False)

    def attach(self, ctx, degree=None, eig_ratio=None):
        from fem_mesh import box_mesh
        n2g, n1g = global_dof_counts(self.nx, self.ny, self.nz)
        ctx.set_partition(self.rank, self.size, self.p2_ghost, self.p1_ghost, self.p2_halo,
                          self.p1_halo, n2g, n1g)
        ctx.mg_prolongations = []                      # kept for attach_schur_laplacian(ctx, ..., part=self)
        for lev, (rowptr, col, val) in self.levels:
            ctx.mg_add_level(lev.mesh.coords, lev.mesh.cells, rowptr, col, val,
                             ghost=lev.p1_ghost, halo=lev.p1_halo)
            ctx.mg_prolongations.append((lev.n_p1, (rowptr, col, val)))
        cg = box_mesh(self.p0, self.p1, *self.coarse_global_shape)
        ctx.mg_set_global_coarse(cg.coords, cg.cells, self.coarse_global_offset)
        for mesh, (rowptr, col, val) in self.global_tail:
            ctx.mg_add_global_level(mesh.coords, mesh.cells, rowptr, col, val)
        ctx.mg_finalize(2 if degree is None else degree, 4.0 if eig_ratio is None else eig_ratio)
        return len(self.levels)


# ---------------------------------------------------------------------------------------------
# triple-periodic boxes (BASELINE configs[3]: 3D Taylor-Green vortex across the GPUs of a node):
# x and y are periodic INSIDE every slab (slave vertices / edges share their master's dof, as in
# TaylorHoodDofMap's periodic maps), z is periodic ACROSS the ranks: the halo exchange wraps
# around (rank size-1 <-> rank 0), every rank has a ghost plane below and a ghost layer above.
# ---------------------------------------------------------------------------------------------
def _xy_periodic_masters(mesh, p0, p1):
    """(entity master [nv + ne], vertex master [nv]) identifying x = p1[0] with x = p0[0] and
    y = p1[1] with y = p0[1] (z untouched), by lattice keys of vertices and edge midpoints."""
    nv = mesh.num_vertices()
    pts = np.concatenate([mesh.coords, mesh.edge_midpoints()], axis=0)
    scale = 8.0 / max(mesh.hmin(), 1e-300)
    L = np.array([p1[0] - p0[0], p1[1] - p0[1], 0.0])
    wrapped = pts.copy()
    for a in range(2):
        on_far = np.abs(pts[:, a] - p1[a]) < 1e-9 * max(1.0, abs(L[a]))
        wrapped[on_far, a] -= L[a]
    key = lambda X: [tuple(r) for r in np.round((X - np.asarray(p0)) * scale).astype(np.int64)]
    moved = np.nonzero(np.abs(wrapped - pts).sum(axis=1) > 0.0)[0]
    master = np.arange(pts.shape[0], dtype=np.int64)
    if moved.size:
        kv = {k: i for i, k in enumerate(key(pts[:nv]))}
        ke = {k: nv + i for i, k in enumerate(key(pts[nv:]))}
        for i, k in zip(moved.tolist(), key(wrapped[moved])):
            master[i] = (kv if i < nv else ke)[k]
    return master, master[:nv].copy()


class PeriodicSlabLevel(SlabLevel):
    """P1 level of a triple-periodic slab: ghost plane below AND ghost layer above on every rank;
    dof = (plane, iy mod ny, ix mod nx), so every lattice plane holds nx * ny dofs."""

    def __init__(self, p0, p1, nx, ny, nz, layer0, own_layers):
        super().__init__(p0, p1, nx, ny, nz + 1, layer0, own_layers, 1)     # (mesh may stick out by one layer)
        # the box above was built on a domain one layer taller so that the last rank's ghost layer
        # exists; restore the true spacing / coordinates
        hz = (p1[2] - p0[2]) / nz
        wv = (nx + 1) * (ny + 1)
        self.mesh.coords[:, 2] = np.repeat(p0[2] + hz * (layer0 + np.arange(own_layers + 2)), wv)
        self.has_below = self.has_above = True
        ix, iy, iz = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), np.arange(own_layers + 2), indexing="ij")
        dof = (iz * ny + (iy % ny)) * nx + (ix % nx)
        self.vertex_dof = dof.transpose(2, 1, 0).ravel().astype(np.int64)    # vertex id = (iz, iy, ix)
        self.w1 = nx * ny
        self.n_p1 = self.w1 * (own_layers + 2)
        ghost = np.zeros(self.n_p1, dtype=np.uint8)
        ghost[: self.w1] = GHOST
        ghost[self.w1 * (own_layers + 1):] = GHOST
        self.p1_ghost = ghost
        w1 = self.w1
        self.p1_halo = dict(send_up=(w1 * own_layers, w1), recv_above=(w1 * (own_layers + 1), w1),
                            send_down=(w1, w1), recv_below=(0, w1))
        self.dofmap = self.vertex_dof[self.mesh.cells.astype(np.int64)].astype(np.int32)


class PeriodicSlabPartition:
    """Rank ``rank`` of ``size`` of the triple-periodic (nx, ny, nz) Kuhn box: owns nz / size cube
    layers (the dofs of its layers except the bottom lattice plane, which belongs to the rank
    below -- for rank 0 that is rank size-1: the plane z = p0 IS the plane z = p1)."""

    periodic = True

    def __init__(self, p0, p1, nx, ny, nz, rank, size, coarsest=4, global_coarsest=None):
        from multigrid import structured_prolongation_3d
        import scipy.sparse as sp
        assert nz % size == 0 and size >= 2, "needs at least two ranks (one rank: use a periodic dof map)"
        own = nz // size
        self.rank, self.size = rank, size
        self.p0, self.p1, self.nx, self.ny, self.nz = tuple(p0), tuple(p1), nx, ny, nz
        self.fine = PeriodicSlabLevel(p0, p1, nx, ny, nz, rank * own, own)
        self.mesh = self.fine.mesh
        self.dofmap = dm = TaylorHoodDofMap(self.mesh, periodic_map=_xy_periodic_masters(self.mesh, p0, (p1[0], p1[1], 0.0)))
        assert np.array_equal(dm.p1_vertex_node, self.fine.vertex_dof)
        w2 = 4 * nx * ny
        self.w2 = w2
        planes = 2 * (own + 1) + 1
        assert dm.n_p2 == w2 * planes and dm.n_p1 == self.fine.n_p1
        z2 = dm.p2_coords[:, 2]
        assert np.all(np.diff(z2) >= -1e-12)                      # plane-contiguous numbering
        ghost2 = np.zeros(dm.n_p2, dtype=np.uint8)
        ghost2[:w2] = GHOST
        ghost2[w2 * (2 * own + 1):] = GHOST
        self.p2_ghost = ghost2
        self.p2_halo = dict(send_up=(w2 * 2 * own, w2), recv_above=(w2 * (2 * own + 1), 2 * w2),
                            send_down=(w2, 2 * w2), recv_below=(0, w2))
        self.p1_ghost, self.p1_halo = self.fine.p1_ghost, self.fine.p1_halo
        n2g, n1g = w2 * 2 * nz, nx * ny * nz
        self.n_p2_global, self.n_p1_global = n2g, n1g
        self.p2_global = (rank * own * 2 * w2 + np.arange(dm.n_p2)) % n2g
        self.p1_global = (rank * own * self.fine.w1 + np.arange(dm.n_p1)) % n1g
        self.p2_owned, self.p1_owned = ghost2 == 0, self.p1_ghost == 0
        self.levels = []
        lx, ly, lz, lown, fine_level = nx, ny, nz, own, self.fine
        while lx % 2 == 0 and ly % 2 == 0 and lown % 2 == 0 and lown // 2 >= 1 \
                and min(lx, ly, lz) // 2 >= coarsest:
            cx, cy, cz, cown = lx // 2, ly // 2, lz // 2, lown // 2
            lev = PeriodicSlabLevel(p0, p1, cx, cy, cz, rank * cown, cown)
            # vertex prolongation from the coarse local box (cown + 1 layers) to its refinement
            # (2 cown + 2 layers), cut to the vertex planes the finer local mesh has (lown + 2)
            rowptr, col, val = structured_prolongation_3d(lx, ly, 2 * (cown + 1))
            n_fine_v = (lx + 1) * (ly + 1) * (lown + 2)
            P = sp.csr_matrix((val, col, rowptr))[:n_fine_v]
            f_dof, c_dof = fine_level.vertex_dof, lev.vertex_dof
            rep = np.full(int(f_dof.max()) + 1, -1, dtype=np.int64)
            rep[f_dof[::-1]] = np.arange(f_dof.size - 1, -1, -1)
            E = sp.csr_matrix((np.ones(c_dof.size), (np.arange(c_dof.size), c_dof)),
                              shape=(c_dof.size, int(c_dof.max()) + 1))
            Pp = (P[rep] @ E).tocsr()
            Pp.sum_duplicates()
            Pp.sort_indices()
            self.levels.append((lev, (Pp.indptr.astype(np.int32), Pp.indices.astype(np.int32), Pp.data.copy())))
            lx, ly, lz, lown, fine_level = cx, cy, cz, cown, lev
        last = self.levels[-1][0] if self.levels else self.fine
        self.coarse_global_shape = (last.nx, last.ny, lz)
        self.coarse_global_offset = last.row0 * last.w1
        self.global_coarsest = global_coarsest

    def _global_periodic_levels(self):
        """the replicated triple-periodic hierarchy below the partitioned levels: global coarse mesh
        (+ coarser levels down to ``global_coarsest``) with their periodic dof maps"""
        from fem_mesh import box_mesh
        from multigrid import periodic_levels, structured_hierarchy
        cx, cy, cz = self.coarse_global_shape
        mesh = box_mesh(self.p0, self.p1, cx, cy, cz)
        ix, iy, iz = np.meshgrid(np.arange(cx + 1), np.arange(cy + 1), np.arange(cz + 1), indexing="ij")
        vdof = (((iz % cz) * cy + (iy % cy)) * cx + (ix % cx)).transpose(2, 1, 0).ravel().astype(np.int64)
        tail = []
        if self.global_coarsest is not None:
            tail = structured_hierarchy(self.p0, self.p1, cx, cy, cz, coarsest=self.global_coarsest, allow_non_nested=False)
        return mesh, vdof, tail

    def attach(self, ctx, degree=None, eig_ratio=None):
        import scipy.sparse as sp
        ctx.set_partition(self.rank, self.size, self.p2_ghost, self.p1_ghost, self.p2_halo,
                          self.p1_halo, self.n_p2_global, self.n_p1_global, periodic=True)
        for lev, (rowptr, col, val) in self.levels:
            ctx.mg_add_level(lev.mesh.coords, lev.mesh.cells, rowptr, col, val,
                             ghost=lev.p1_ghost, halo=lev.p1_halo, dofmap=lev.dofmap)
        mesh, vdof, tail = self._global_periodic_levels()
        ctx.mg_set_global_coarse(mesh.coords, mesh.cells, self.coarse_global_offset,
                                 dofmap=vdof[mesh.cells.astype(np.int64)])
        f_dof = vdof
        for cmesh, (rowptr, col, val) in tail:            # replicated periodic levels below it
            shape = cmesh.structured[2:]
            ix, iy, iz = np.meshgrid(*[np.arange(k + 1) for k in shape], indexing="ij")
            c_dof = (((iz % shape[2]) * shape[1] + (iy % shape[1])) * shape[0] + (ix % shape[0])) \
                .transpose(2, 1, 0).ravel().astype(np.int64)
            P = sp.csr_matrix((val, col, rowptr), shape=(f_dof.size, c_dof.size))
            rep = np.full(int(f_dof.max()) + 1, -1, dtype=np.int64)
            rep[f_dof[::-1]] = np.arange(f_dof.size - 1, -1, -1)
            E = sp.csr_matrix((np.ones(c_dof.size), (np.arange(c_dof.size), c_dof)),
                              shape=(c_dof.size, int(c_dof.max()) + 1))
            Pp = (P[rep] @ E).tocsr()
            Pp.sum_duplicates()
            Pp.sort_indices()
            ctx.mg_add_global_level(cmesh.coords, cmesh.cells, Pp.indptr.astype(np.int32),
                                    Pp.indices.astype(np.int32), Pp.data.copy(),
                                    dofmap=c_dof[cmesh.cells.astype(np.int64)])
            f_dof = c_dof
        ctx.mg_finalize(2 if degree is None else degree, 4.0 if eig_ratio is None else eig_ratio)
        return len(self.levels)


# ---------------------------------------------------------------------------------------------
# doubly periodic rectangles (the reference's Taylor-Green convergence study in 2D): x periodic
# inside every strip, y across the ranks -- the 2D counterpart of PeriodicSlabPartition
# ---------------------------------------------------------------------------------------------
class PeriodicStripLevel(StripLevel):
    def __init__(self, p0, p1, nx, ny, row0, own_rows):
        hy = (p1[1] - p0[1]) / ny
        super().__init__(p0, (p1[0], p1[1] + hy), nx, ny + 1, row0, own_rows, 1)
        self.mesh.coords[:, 1] = np.repeat(p0[1] + hy * (row0 + np.arange(own_rows + 2)), nx + 1)
        self.has_below = self.has_above = True
        ix, iy = np.meshgrid(np.arange(nx + 1), np.arange(own_rows + 2), indexing="xy")
        self.vertex_dof = (iy * nx + (ix % nx)).ravel().astype(np.int64)          # vertex id = iy (nx+1) + ix
        w1 = self.w1 = nx
        self.n_p1 = w1 * (own_rows + 2)
        ghost = np.zeros(self.n_p1, dtype=np.uint8)
        ghost[:w1] = GHOST
        ghost[w1 * (own_rows + 1):] = GHOST
        self.p1_ghost = ghost
        self.p1_halo = dict(send_up=(w1 * own_rows, w1), recv_above=(w1 * (own_rows + 1), w1),
                            send_down=(w1, w1), recv_below=(0, w1))
        self.dofmap = self.vertex_dof[self.mesh.cells.astype(np.int64)].astype(np.int32)


class PeriodicStripPartition:
    """Rank ``rank`` of ``size`` of the doubly periodic (nx, ny) right-diagonal rectangle mesh."""

    periodic = True

    def __init__(self, p0, p1, nx, ny, rank, size, coarsest=4, global_coarsest=None):
        import scipy.sparse as sp
        assert ny % size == 0 and size >= 2
        own = ny // size
        self.rank, self.size = rank, size
        self.p0, self.p1, self.nx, self.ny = tuple(p0), tuple(p1), nx, ny
        self.fine = PeriodicStripLevel(p0, p1, nx, ny, rank * own, own)
        self.mesh = self.fine.mesh
        masters = _xy_periodic_masters(self.mesh, (p0[0], p0[1]), (p1[0], np.inf))   # x only
        self.dofmap = dm = TaylorHoodDofMap(self.mesh, periodic_map=masters)
        assert np.array_equal(dm.p1_vertex_node, self.fine.vertex_dof)
        w2 = self.w2 = 2 * nx
        lines = 2 * (own + 1) + 1
        assert dm.n_p2 == w2 * lines and np.all(np.diff(dm.p2_coords[:, 1]) >= -1e-12)
        ghost2 = np.zeros(dm.n_p2, dtype=np.uint8)
        ghost2[:w2] = GHOST
        ghost2[w2 * (2 * own + 1):] = GHOST
        self.p2_ghost = ghost2
        self.p2_halo = dict(send_up=(w2 * 2 * own, w2), recv_above=(w2 * (2 * own + 1), 2 * w2),
                            send_down=(w2, 2 * w2), recv_below=(0, w2))
        self.p1_ghost, self.p1_halo = self.fine.p1_ghost, self.fine.p1_halo
        self.n_p2_global, self.n_p1_global = w2 * 2 * ny, nx * ny
        self.p2_global = (rank * own * 2 * w2 + np.arange(dm.n_p2)) % self.n_p2_global
        self.p1_global = (rank * own * nx + np.arange(dm.n_p1)) % self.n_p1_global
        self.p2_owned, self.p1_owned = ghost2 == 0, self.p1_ghost == 0
        self.levels = []
        lx, ly, lown, fine_level = nx, ny, own, self.fine
        while lx % 2 == 0 and lown % 2 == 0 and lown // 2 >= 1 and min(lx, ly) // 2 >= coarsest:
            cx, cy, cown = lx // 2, ly // 2, lown // 2
            lev = PeriodicStripLevel(p0, p1, cx, cy, rank * cown, cown)
            rowptr, col, val = structured_prolongation(lx, 2 * (cown + 1))
            P = sp.csr_matrix((val, col, rowptr))[: (lx + 1) * (lown + 2)]
            self.levels.append((lev, _constrain_prolongation(P, fine_level.vertex_dof, lev.vertex_dof)))
            lx, ly, lown, fine_level = cx, cy, cown, lev
        last = self.levels[-1][0] if self.levels else self.fine
        self.coarse_global_shape = (last.nx, ly)
        self.coarse_global_offset = last.row0 * last.w1
        self.global_coarsest = global_coarsest

    def attach(self, ctx, degree=None, eig_ratio=None):
        from fem_mesh import rectangle_mesh
        from multigrid import structured_hierarchy
        ctx.set_partition(self.rank, self.size, self.p2_ghost, self.p1_ghost, self.p2_halo,
                          self.p1_halo, self.n_p2_global, self.n_p1_global, periodic=True)
        for lev, (rowptr, col, val) in self.levels:
            ctx.mg_add_level(lev.mesh.coords, lev.mesh.cells, rowptr, col, val,
                             ghost=lev.p1_ghost, halo=lev.p1_halo, dofmap=lev.dofmap)
        cx, cy = self.coarse_global_shape

        def vdof_of(shape):
            ix, iy = np.meshgrid(np.arange(shape[0] + 1), np.arange(shape[1] + 1), indexing="xy")
            return ((iy % shape[1]) * shape[0] + (ix % shape[0])).ravel().astype(np.int64)

        mesh = rectangle_mesh(self.p0, self.p1, cx, cy)
        f_dof = vdof_of((cx, cy))
        ctx.mg_set_global_coarse(mesh.coords, mesh.cells, self.coarse_global_offset,
                                 dofmap=f_dof[mesh.cells.astype(np.int64)])
        tail = [] if self.global_coarsest is None else \
            structured_hierarchy(self.p0, self.p1, cx, cy, coarsest=self.global_coarsest, allow_non_nested=False)
        import scipy.sparse as sp
        for cmesh, (rowptr, col, val) in tail:
            c_dof = vdof_of(cmesh.structured[2:])
            P = sp.csr_matrix((val, col, rowptr), shape=(f_dof.size, c_dof.size))
            rp, ci, cv = _constrain_prolongation(P, f_dof, c_dof)
            ctx.mg_add_global_level(cmesh.coords, cmesh.cells, rp, ci, cv,
                                    dofmap=c_dof[cmesh.cells.astype(np.int64)])
            f_dof = c_dof
        ctx.mg_finalize(2 if degree is None else degree, 4.0 if eig_ratio is None else eig_ratio)
        return len(self.levels)


def _constrain_prolongation(P, f_dof, c_dof):
    """vertex prolongation P -> prolongation between constrained (periodic) dof sets: rows of one
    representative vertex per finer dof, columns of all vertices of a coarse dof added up"""
    import scipy.sparse as sp
    rep = np.full(int(f_dof.max()) + 1, -1, dtype=np.int64)
    rep[f_dof[::-1]] = np.arange(f_dof.size - 1, -1, -1)
    E = sp.csr_matrix((np.ones(c_dof.size), (np.arange(c_dof.size), c_dof)),
                      shape=(c_dof.size, int(c_dof.max()) + 1))
    Pp = (P[rep] @ E).tocsr()
    Pp.sum_duplicates()
    Pp.sort_indices()
    return Pp.indptr.astype(np.int32), Pp.indices.astype(np.int32), Pp.data.copy()


# ---------------------------------------------------------------------------------------------
# unstructured meshes (BASELINE configs[2]: DFG channel): recursive coordinate bisection of the
# COARSEST mesh of a refinement hierarchy, inherited by the children -- the partitions of all
# levels are nested by construction.  Halos are index lists (nsfem_set_halo_lists), any number of
# neighbours per rank.
# ---------------------------------------------------------------------------------------------
def recursive_bisection(points, n_parts):
    """owner [n] in 0 .. n_parts-1 by recursive coordinate bisection: split along the longest
    axis of the bounding box at the count-weighted median; n_parts need not be a power of two."""
    points = np.asarray(points, dtype=np.float64)
    owner = np.zeros(points.shape[0], dtype=np.int64)

    def split(idx, first, count):
        if count == 1:
            owner[idx] = first
            return
        left = count // 2
        pts = points[idx]
        axis = int(np.argmax(pts.max(axis=0) - pts.min(axis=0)))
        order = np.argsort(pts[:, axis], kind="stable")
        cut = int(round(idx.size * left / count))
        split(idx[order[:cut]], first, left)
        split(idx[order[cut:]], first + left, count - left)

    split(np.arange(points.shape[0]), 0, int(n_parts))
    return owner


class GraphLevel:
    """one level of one rank: local sub-mesh (global vertex order kept), ghost flags, halo lists"""


def _halo_lists(rank, size, global_ids, owner_of, local_on, local_index):
    """index lists of one node family.  global_ids: ascending global ids of MY local nodes;
    owner_of[g]: owning rank; local_on[q][g]: node g is local on rank q; local_index: my local
    index of every entry of global_ids.  Both sides of a pair order by global id."""
    own = owner_of[global_ids]
    nbr, sp, si, rp, ri = [], [0], [], [0], []
    for q in range(size):
        if q == rank:
            continue
        recv = local_index[own == q]
        send = local_index[(own == rank) & local_on[q][global_ids]]
        if recv.size == 0 and send.size == 0:
            continue
        nbr.append(q)
        si.append(send)
        ri.append(recv)
        sp.append(sp[-1] + send.size)
        rp.append(rp[-1] + recv.size)
    cat = lambda parts: np.concatenate(parts).astype(np.int32) if parts else np.zeros(0, np.int32)
    return dict(neighbour=np.asarray(nbr, dtype=np.int32), send_ptr=np.asarray(sp, dtype=np.int64),
                send_idx=cat(si), recv_ptr=np.asarray(rp, dtype=np.int64), recv_idx=cat(ri))


class GraphPartition:
    """Rank ``rank`` of ``size`` of an unstructured triangle mesh with a refinement hierarchy
    (``fine_mesh.mg_levels`` of multigrid.refinement_hierarchy; children of cell c are 4c .. 4c+3).

    * cells: the coarsest cells are split by recursive coordinate bisection of their centroids,
      every finer cell belongs to its parent's rank;
    * nodes (vertices, and edge midpoints on the finest level): owned by the LOWEST rank among the
      cells that contain them;
    * local cells of a level: every cell that contains a node the rank owns (its owned rows are
      then complete) plus, on coarser levels, the parents of all local cells of the next finer
      level (every local fine node then finds its whole prolongation row on the rank);
    * the coarsest mesh is replicated (global coarse solve), local coarsest nodes map into it by
      an index list.
    Same attributes as the strip / slab partitions (mesh, dofmap, p*_ghost, p*_owned, levels)."""

    periodic = False

    def __init__(self, fine_mesh, rank, size, markers=None):
        import scipy.sparse as sp
        from fem_mesh import FacetMarkers, Mesh
        assert fine_mesh._dim == 2, "refinement hierarchies exist for triangle meshes"
        hierarchy = getattr(fine_mesh, "mg_levels", [])
        meshes = [fine_mesh] + [m for m, _ in hierarchy]
        prolongs = [P for _, P in hierarchy]
        nlev = len(meshes)
        self.rank, self.size = rank, size
        self.global_meshes = meshes
        cell_owner = [None] * nlev
        cell_owner[-1] = recursive_bisection(meshes[-1].coords[meshes[-1].cells.astype(np.int64)].mean(axis=1), size)
        for l in range(nlev - 2, -1, -1):
            cell_owner[l] = np.repeat(cell_owner[l + 1], 4)
            assert cell_owner[l].size == meshes[l].num_cells(), "not a red-refinement hierarchy"
        self.cell_owner = cell_owner
        # node owners and, for every rank, the local cells / nodes of every level
        vown, vloc, cloc = [], [], []
        for l, m in enumerate(meshes):
            c = m.cells.astype(np.int64)
            vo = np.full(m.num_vertices(), size, dtype=np.int64)
            np.minimum.at(vo, c.ravel(), np.repeat(cell_owner[l], 3))
            vown.append(vo)
        ce = fine_mesh.cell_edges.astype(np.int64)
        eown = np.full(fine_mesh.num_edges(), size, dtype=np.int64)
        np.minimum.at(eown, ce.ravel(), np.repeat(cell_owner[0], 3))
        for l, m in enumerate(meshes):
            c = m.cells.astype(np.int64)
            loc = np.zeros((size, m.num_cells()), dtype=bool)
            for q in range(size):
                loc[q] = (vown[l][c] == q).any(axis=1)
                if l == 0:
                    loc[q] |= (eown[ce] == q).any(axis=1)
                else:
                    loc[q] |= cloc[l - 1][q].reshape(-1, 4).any(axis=1)
            cloc.append(loc)
            vl = np.zeros((size, m.num_vertices()), dtype=bool)
            for q in range(size):
                vl[q][c[loc[q]].ravel()] = True
            vloc.append(vl)
        eloc = np.zeros((size, fine_mesh.num_edges()), dtype=bool)
        for q in range(size):
            eloc[q][ce[cloc[0][q]].ravel()] = True
        # ---- my local meshes.  Local vertex order: interior vertices first, then the owned ones
        # next to a ghost, then the ghosts (global order inside a class): the rows that reference no
        # ghost column -- those that can run under a halo exchange -- are one contiguous range
        self.level_vertices = []
        made = []
        for l, m in enumerate(meshes):
            lv = np.nonzero(vloc[l][rank])[0]                      # ascending global ids
            lc = np.nonzero(cloc[l][rank])[0]
            gc = m.cells[lc].astype(np.int64)
            ghost_g = vown[l] != rank
            cls = np.zeros(m.num_vertices(), dtype=np.int64)
            cls[gc[ghost_g[gc].any(axis=1)].ravel()] = 1
            cls[ghost_g] = 2
            order = np.lexsort((lv, cls[lv]))
            lv_local = lv[order]                                   # global id of local vertex i
            g2l = np.full(m.num_vertices(), -1, dtype=np.int64)
            g2l[lv_local] = np.arange(lv.size)
            local = Mesh(m.coords[lv_local], g2l[gc].astype(np.int32))
            lev = GraphLevel()
            lev.mesh, lev.vertices, lev.cells, lev.g2l = local, lv_local, lc, g2l
            lev.n_p1 = lv.size
            lev.p1_ghost = (vown[l][lv_local] != rank).astype(np.uint8)
            lev.p1_lists = _halo_lists(rank, size, lv, vown[l], vloc[l], g2l[lv])
            made.append(lev)
            self.level_vertices.append(lv_local)
        self.fine = fine = made[0]
        self.mesh = fine.mesh
        # global edge id of every local edge
        nvg = fine_mesh.num_vertices()
        ge_key = fine_mesh.edges[:, 0].astype(np.int64) * nvg + fine_mesh.edges[:, 1]    # sorted (np.unique)
        le = fine.vertices[fine.mesh.edges.astype(np.int64)]
        key = le.min(axis=1) * nvg + le.max(axis=1)
        ge = np.searchsorted(ge_key, key)
        assert np.array_equal(ge_key[ge], key)
        self.edge_global = ge
        nv_loc = fine.vertices.size
        ent_global = np.concatenate([fine.vertices, nvg + ge])              # entity order of the local mesh
        ent_owner = np.concatenate([vown[0], eown])
        ent_ghost = ent_owner[ent_global] != rank
        cell_ent = np.concatenate([fine.mesh.cells.astype(np.int64), nv_loc + fine.mesh.cell_edges.astype(np.int64)], axis=1)
        ent_class = np.zeros(ent_global.size, dtype=np.int64)
        ent_class[cell_ent[ent_ghost[cell_ent].any(axis=1)].ravel()] = 1
        ent_class[ent_ghost] = 2
        self.dofmap = dm = TaylorHoodDofMap(fine.mesh, class_key=ent_class)
        ent_dof = np.concatenate([dm.vertex_node, dm.edge_node])
        order = np.argsort(ent_global, kind="stable")
        self.p2_entity_global = np.empty(dm.n_p2, dtype=np.int64)           # local P2 node -> global entity id
        self.p2_entity_global[ent_dof] = ent_global
        ghost2 = np.zeros(dm.n_p2, dtype=np.uint8)
        ghost2[ent_dof] = ent_ghost
        self.p2_ghost = ghost2
        self.p1_ghost = fine.p1_ghost
        self.p2_owned, self.p1_owned = ghost2 == 0, fine.p1_ghost == 0
        ent_local = [np.concatenate([vloc[0][q], eloc[q]]) for q in range(size)]
        self.p2_lists = _halo_lists(rank, size, ent_global[order], ent_owner, ent_local, ent_dof[order])
        self.p1_lists = fine.p1_lists
        self.p1_global = fine.vertices
        self.n_p2_global = nvg + fine_mesh.num_edges()
        self.n_p1_global = nvg
        # local facet markers: the global marker of every local edge (cut edges are interior: 0)
        self.markers = None
        if markers is not None:
            self.markers = FacetMarkers(fine.mesh, 0)
            self.markers.values[:] = markers.values[ge]
            # a cut edge is a boundary edge of the local mesh but not of the domain
            fine.mesh.facet_on_boundary = fine.mesh.facet_on_boundary & fine_mesh.facet_on_boundary[ge]
            fine.mesh.edge_on_boundary = fine.mesh.facet_on_boundary
        # ---- coarse levels with rank-local prolongations
        self.levels = []
        for l in range(1, nlev):
            rowptr, col, val = prolongs[l - 1]
            nf, ncoarse = meshes[l - 1].num_vertices(), meshes[l].num_vertices()
            P = sp.csr_matrix((val, col, rowptr), shape=(nf, ncoarse))[made[l - 1].vertices]
            P = P.tocoo()
            cols = made[l].g2l[P.col]
            assert (cols >= 0).all(), "a local fine node interpolates from a coarse node the rank does not hold"
            Pl = sp.csr_matrix((P.data, (P.row, cols)), shape=(made[l - 1].n_p1, made[l].n_p1))
            Pl.sort_indices()
            self.levels.append((made[l], (Pl.indptr.astype(np.int32), Pl.indices.astype(np.int32), Pl.data.copy())))
        self.coarse_global_index = made[-1].vertices

    def p2_global(self, global_dofmap):
        """local P2 node -> node of a TaylorHoodDofMap of the GLOBAL fine mesh (comparisons)"""
        ent_to_node = np.concatenate([global_dofmap.vertex_node, global_dofmap.edge_node])
        return ent_to_node[self.p2_entity_global]

    def attach(self, ctx, degree=None, eig_ratio=None):
        ctx.set_partition(self.rank, self.size, self.p2_ghost, self.p1_ghost, None, None,
                          self.n_p2_global, self.n_p1_global)
        ctx.set_halo_lists(0, self.p2_lists)
        ctx.set_halo_lists(1, self.p1_lists)
        ctx.mg_prolongations = []
        for l, (lev, (rowptr, col, val)) in enumerate(self.levels):
            ctx.mg_add_level(lev.mesh.coords, lev.mesh.cells, rowptr, col, val, ghost=lev.p1_ghost, halo=None)
            ctx.set_halo_lists(2 + l, lev.p1_lists)
            ctx.mg_prolongations.append((lev.n_p1, (rowptr, col, val)))
        cg = self.global_meshes[-1]
        ctx.mg_set_global_coarse(cg.coords, cg.cells, 0)
        ctx.mg_set_global_index(self.coarse_global_index)
        # general (graded, curved) meshes: 3 smoothing steps over a ratio of 16 (multigrid.attach_hierarchy)
        ctx.mg_finalize(3 if degree is None else degree, 16.0 if eig_ratio is None else eig_ratio)
        return len(self.levels)
