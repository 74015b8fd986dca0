"""Host-side construction of nested P1 mesh hierarchies for the device multigrid
preconditioners (csrc/multigrid.hip; C ABI nsfem_mg_add_level / nsfem_mg_finalize).

The reference has no iterative solver or preconditioner (it calls sparse LU,
source/ns_solver_base.py:938; README.md:18,32 list them as TODO): this is new functionality
needed to replace the direct solves at scale.  For the structured right-diagonal meshes of
``grid_generator`` the (nx, ny) mesh is the uniform refinement of the (nx/2, ny/2) mesh,
so the P1 spaces are nested and the prolongation is linear interpolation at the new
vertices (edge midpoints of the coarse mesh, including its diagonals).
"""
import numpy as np

from fem_mesh import rectangle_mesh


def structured_prolongation(nx, ny):
    """CSR (rowptr, col, val) of the P1 prolongation from the (nx/2, ny/2) to the (nx, ny)
    right-diagonal mesh; rows = fine vertices (id = iy (nx+1) + ix)."""
    assert nx % 2 == 0 and ny % 2 == 0
    cx = nx // 2
    ix, iy = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), indexing="xy")
    ix, iy = ix.ravel(), iy.ravel()
    ox, oy = ix % 2, iy % 2

    def cid(jx, jy):
        return jy * (cx + 1) + jx

    # first parent: floor, second parent: ceil along the coarse edge the vertex bisects
    # (horizontal, vertical or the right diagonal (0,0)-(1,1) of a coarse quad)
    a = cid((ix - ox) // 2, (iy - oy) // 2)
    b = cid((ix + ox) // 2, (iy + oy) // 2)
    single = (ox == 0) & (oy == 0)
    counts = np.where(single, 1, 2)
    rowptr = np.zeros(ix.size + 1, dtype=np.int32)
    np.cumsum(counts, out=rowptr[1:])
    col = np.empty(rowptr[-1], dtype=np.int32)
    val = np.empty(rowptr[-1], dtype=np.float64)
    first = rowptr[:-1]
    col[first] = a
    val[first] = np.where(single, 1.0, 0.5)
    second = first[~single] + 1
    col[second] = b[~single]
    val[second] = 0.5
    return rowptr, col, val


def interpolation_prolongation(nx, ny, cx, cy):
    """P1 prolongation between NON-NESTED right-diagonal meshes of the same rectangle: the value of the coarse
    (cx, cy) hat functions at every vertex of the fine (nx, ny) mesh (barycentric weights in the coarse triangle
    that holds it).  Lets hierarchies continue below a level with an odd number of cells (333 -> 167 -> 84 -> ...);
    for nx = 2 cx, ny = 2 cy it equals ``structured_prolongation``.  CSR (rowptr, col, val), rows = fine vertices."""
    ix, iy = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), indexing="xy")
    ix, iy = ix.ravel(), iy.ravel()
    # position in coarse cell units, exact rational arithmetic in integers: x = ix cx / nx
    qx, rx = np.divmod(ix * cx, nx)
    qy, ry = np.divmod(iy * cy, ny)
    on_right = qx == cx
    qx = np.where(on_right, cx - 1, qx); rx = np.where(on_right, nx, rx)        # last vertex: s = 1 in the last cell
    on_top = qy == cy
    qy = np.where(on_top, cy - 1, qy); ry = np.where(on_top, ny, ry)
    s_ = rx / float(nx)
    t_ = ry / float(ny)
    v0 = qy * (cx + 1) + qx
    v1, v2, v3 = v0 + 1, v0 + (cx + 1), v0 + (cx + 1) + 1
    below = rx * ny >= ry * nx                      # s >= t: triangle (v0, v1, v3), else (v0, v2, v3)
    w0 = np.where(below, 1.0 - s_, 1.0 - t_)
    wm = np.where(below, s_ - t_, t_ - s_)
    vm = np.where(below, v1, v2)
    w3 = np.where(below, t_, s_)
    cols = np.stack([v0, vm, v3], axis=1)
    vals = np.stack([w0, wm, w3], axis=1)
    keep = vals > 0.0
    counts = keep.sum(axis=1)
    rowptr = np.zeros(ix.size + 1, dtype=np.int32)
    np.cumsum(counts, out=rowptr[1:])
    # (columns ascending within a row: v0 < v1 < v3 and v0 < v2 < v3)
    return rowptr, cols[keep].astype(np.int32), vals[keep].astype(np.float64)


def interpolation_prolongation_3d(nx, ny, nz, cx, cy, cz):
    """The 3D counterpart of ``interpolation_prolongation`` for the Kuhn box meshes of ``fem_mesh.box_mesh``: a point
    with cube-local coordinates ordered  a >= b >= c  (axes p, q, r) lies in the tetrahedron  v000, + e_p, + e_q, v111  and
    has the barycentric weights  1 - a, a - b, b - c, c.  Integer arithmetic decides the ordering; for even sizes it equals
    ``structured_prolongation_3d``."""
    iz, iy, ix = np.meshgrid(np.arange(nz + 1), np.arange(ny + 1), np.arange(nx + 1), indexing="ij")
    idx = [ix.ravel(), iy.ravel(), iz.ravel()]
    nf, nc = [nx, ny, nz], [cx, cy, cz]
    q, r = [], []
    for a in range(3):
        qa, ra = np.divmod(idx[a] * nc[a], nf[a])
        last = qa == nc[a]
        q.append(np.where(last, nc[a] - 1, qa))
        r.append(np.where(last, nf[a], ra))
    # local coordinates as exact fractions r[a] / nf[a]: compare through the common denominator
    den = nf[0] * nf[1] * nf[2]
    num = np.stack([r[a] * (den // nf[a]) for a in range(3)], axis=1)            # [n, 3] integers
    order = np.argsort(-num, axis=1, kind="stable")                              # axes by descending coordinate
    srt = np.take_along_axis(num, order, axis=1) / float(den)
    stride = np.array([1, cx + 1, (cx + 1) * (cy + 1)], dtype=np.int64)
    v0 = q[0] + q[1] * stride[1] + q[2] * stride[2]
    v1 = v0 + stride[order[:, 0]]
    v2 = v1 + stride[order[:, 1]]
    v3 = v0 + stride.sum()
    cols = np.stack([v0, v1, v2, v3], axis=1)
    vals = np.stack([1.0 - srt[:, 0], srt[:, 0] - srt[:, 1], srt[:, 1] - srt[:, 2], srt[:, 2]], axis=1)
    keep = vals > 0.0
    counts = keep.sum(axis=1)
    rowptr = np.zeros(cols.shape[0] + 1, dtype=np.int32)
    np.cumsum(counts, out=rowptr[1:])
    # columns ascending within a row: sort the kept entries of every row
    key = np.where(keep, cols, np.iinfo(np.int64).max)
    perm = np.argsort(key, axis=1, kind="stable")
    cols = np.take_along_axis(cols, perm, axis=1)
    vals = np.take_along_axis(vals, perm, axis=1)
    keep = np.take_along_axis(keep, perm, axis=1)
    return rowptr, cols[keep].astype(np.int32), vals[keep].astype(np.float64)


def structured_prolongation_3d(nx, ny, nz):
    """P1 prolongation between the Kuhn meshes (nx/2, ny/2, nz/2) -> (nx, ny, nz) of
    ``fem_mesh.box_mesh``: every edge of the Kuhn split points in a direction of {0,1}^3, so a fine
    vertex with odd index set S is the midpoint of the coarse edge from floor(i/2) to
    floor(i/2) + 1_S (rows = fine vertices, id = (iz (ny+1) + iy)(nx+1) + ix)."""
    assert nx % 2 == 0 and ny % 2 == 0 and nz % 2 == 0
    cx, cy = nx // 2, ny // 2
    iz, iy, ix = np.meshgrid(np.arange(nz + 1), np.arange(ny + 1), np.arange(nx + 1), indexing="ij")
    ix, iy, iz = ix.ravel(), iy.ravel(), iz.ravel()
    ox, oy, oz = ix % 2, iy % 2, iz % 2

    def cid(jx, jy, jz):
        return (jz * (cy + 1) + jy) * (cx + 1) + jx

    a = cid((ix - ox) // 2, (iy - oy) // 2, (iz - oz) // 2)
    b = cid((ix + ox) // 2, (iy + oy) // 2, (iz + oz) // 2)
    single = (ox == 0) & (oy == 0) & (oz == 0)
    counts = np.where(single, 1, 2)
    rowptr = np.zeros(ix.size + 1, dtype=np.int32)
    np.cumsum(counts, out=rowptr[1:])
    col = np.empty(rowptr[-1], dtype=np.int32)
    val = np.empty(rowptr[-1], dtype=np.float64)
    first = rowptr[:-1]
    col[first] = a
    val[first] = np.where(single, 1.0, 0.5)
    second = first[~single] + 1
    col[second] = b[~single]
    val[second] = 0.5
    return rowptr, col, val


def structured_hierarchy(p0, p1, nx, ny, nz=None, coarsest=None, dense_max=1200, allow_non_nested=True, tail_nodes=0):
    """[(coarse mesh, prolongation CSR to the next finer mesh), ...] finest-first for 2D
    right-diagonal rectangle meshes or (with ``nz``) 3D Kuhn box meshes.
    ``coarsest`` given: coarsen while every direction stays even and >= ``coarsest`` cells.
    ``coarsest`` None: coarsen until the level has at most ``dense_max`` nodes (the device solves
    the coarsest level with a dense inverse; one more level of ~8 small kernels costs more than a
    1000-unknown dense mat-vec).  2D: a level with an odd number of cells in some direction is followed by the
    NON-NESTED mesh of ceil(n / 2) cells (``interpolation_prolongation``; the coarse operators are rediscretised on
    every level anyway; 3D: ``interpolation_prolongation_3d``); ``allow_non_nested=False``: the hierarchy ends where a
    direction becomes odd.
    ``tail_nodes`` > 0 (2D, ``coarsest`` None): where the rule above would stop, NESTED coarsening goes on while every
    direction stays even and the level has more than ``tail_nodes`` nodes -- the device runs the levels of <= 65 x 65
    nodes, the dense coarsest solve included, in ONE single-workgroup launch (csrc/mglegs.hip: the tail), where a
    further level costs a microsecond and a 1089-unknown dense product does not fit."""
    from fem_mesh import box_mesh
    n = [nx, ny] if nz is None else [nx, ny, nz]
    levels = []
    nested = True

    def nodes(m):
        return int(np.prod([k + 1 for k in m]))

    while True:
        even = all(k % 2 == 0 for k in n)
        if not even and (not allow_non_nested or min(n) < 5):
            break                                   # (periodic and multi-rank hierarchies: nested levels only)
        nc = [(k + 1) // 2 for k in n]
        if coarsest is not None:
            if min(nc) < coarsest:
                break
        elif nodes(n) <= dense_max or min(nc) < 2:
            if not (tail_nodes > 0 and nz is None and even and nested and nodes(n) > tail_nodes and min(nc) >= 2):
                break
        if even:
            P = structured_prolongation(*n) if nz is None else structured_prolongation_3d(*n)
        else:
            # an odd number of cells: the next mesh is not nested -- linear interpolation between the two meshes
            P = interpolation_prolongation(n[0], n[1], nc[0], nc[1]) if nz is None else \
                interpolation_prolongation_3d(n[0], n[1], n[2], nc[0], nc[1], nc[2])
        n = nc
        nested = nested and even
        mesh = rectangle_mesh(p0, p1, *n) if nz is None else box_mesh(p0, p1, *n)
        mesh.nested_in_finer = bool(even)          # (how row flags travel down this transfer: nsfem_mg_level_desc)
        levels.append((mesh, P))
    return levels


def periodic_levels(levels, fine_vertex_dof, domain):
    """Turn the vertex-based levels of ``structured_hierarchy`` into levels of the PERIODIC P1
    spaces (dolfin ``constrained_domain``; reference source/ns_solver_base.py:516-518): every
    coarse mesh gets the dof of each vertex (slaves share their master's, as on the fine mesh:
    ``TaylorHoodDofMap``), and the vertex prolongation P becomes  S_f P E_c  -- rows of one
    representative vertex per finer dof, columns of all vertices of a coarse dof added up.
    Returns [(mesh, (rowptr, col, val), cell dof map), ...] and is only valid for nested meshes
    whose periodic images are lattice points on every level (structured meshes are)."""
    import scipy.sparse as sp
    from fem_mesh import _compact, periodic_entity_map
    out = []
    f_dof = np.asarray(fine_vertex_dof, dtype=np.int64)
    for mesh, (rowptr, col, val) in levels:
        nvc = mesh.num_vertices()
        c_dof = _compact(np.arange(nvc, dtype=np.int64)[periodic_entity_map(mesh, domain)[1]])
        n_f, n_c = int(f_dof.max()) + 1, int(c_dof.max()) + 1
        P = sp.csr_matrix((val, col, rowptr), shape=(f_dof.size, nvc))
        rep = np.full(n_f, -1, dtype=np.int64)                    # one vertex per finer dof
        rep[f_dof[::-1]] = np.arange(f_dof.size - 1, -1, -1)
        E = sp.csr_matrix((np.ones(nvc), (np.arange(nvc), c_dof)), shape=(nvc, n_c))
        Pp = (P[rep] @ E).tocsr()
        Pp.sum_duplicates()
        Pp.sort_indices()
        # all periodic images of a point interpolate alike: the representative's row is THE row
        assert abs(Pp.sum(axis=1) - 1.0).max() < 1e-12
        out.append((mesh, (Pp.indptr.astype(np.int32), Pp.indices.astype(np.int32), Pp.data.copy()),
                    c_dof[mesh.cells.astype(np.int64)].astype(np.int32)))
        f_dof = c_dof
    return out


def attach_hierarchy(ctx, mesh, degree=None, eig_ratio=None, coarsest=None, periodic=None):
    """Build the hierarchy of a structured mesh (``mesh.structured`` = (p0, p1, nx, ny)) on the
    device context.  Returns the number of coarse P1 levels (0: mesh cannot be coarsened; the
    two-level P2 -> P1 hierarchy is still built).

    Chebyshev smoother: ``degree`` steps over [lambda_max / eig_ratio, lambda_max].  Defaults
    (None): 2 steps over a ratio of 4 on uniform structured meshes; 3 steps over a ratio of 16 on
    refinement hierarchies of general (graded, curved) meshes, where the narrow interval leaves
    too much of the spectrum to the Krylov method (DFG channel, BDF-2: 53.8 -> 35.3 ms/step,
    29 -> 16 BiCGStab iterations per step)."""
    info = getattr(mesh, "structured", None) if mesh is not None else None
    general = mesh is not None and hasattr(mesh, "mg_levels")
    if degree is None:
        degree = 3 if general else 2
    if eig_ratio is None:
        eig_ratio = 16.0 if general else 4.0
    if general:
        levels = mesh.mg_levels                        # refinement hierarchy of a general mesh
    else:
        # coarsest None: down to the first level with <= 1200 nodes (512^2 -> 32^2 = 1089 nodes,
        # 64^3 -> 8^3 = 729), solved with a dense inverse on the device
        # (experimental fused legs, NSFEM_MG_LEGS=1: nested 2D hierarchies continue down to <= 100 nodes for the
        #  single-workgroup tail of csrc/mglegs.hip)
        import os
        tail = 100 if (periodic is None and os.environ.get("NSFEM_MG_LEGS", "0") not in ("", "0")) else 0
        levels = structured_hierarchy(*info, coarsest=coarsest, allow_non_nested=periodic is None,
                                      tail_nodes=tail) if info is not None else []
    ctx.mg_prolongations = []                          # kept for attach_schur_laplacian
    if periodic is not None:
        # periodic = (constrained domain, P1 dof of every fine-mesh vertex): the coarse levels
        # carry the periodic identification too; needs a coarsenable structured mesh
        domain, fine_vertex_dof = periodic
        for coarse_mesh, (rowptr, col, val), dofmap in periodic_levels(levels, fine_vertex_dof, domain):
            ctx.mg_add_level(coarse_mesh.coords, coarse_mesh.cells, rowptr, col, val, dofmap=dofmap)
            ctx.mg_prolongations.append((int(dofmap.max()) + 1, (rowptr, col, val)))
        ctx.mg_finalize(degree, eig_ratio)
        return len(levels)
    for coarse_mesh, (rowptr, col, val) in levels:
        ctx.mg_add_level(coarse_mesh.coords, coarse_mesh.cells, rowptr, col, val,
                         nested=getattr(coarse_mesh, "nested_in_finer", None))
        ctx.mg_prolongations.append((coarse_mesh.coords.shape[0], (rowptr, col, val)))
    ctx.mg_finalize(degree, eig_ratio)
    return len(levels)


def _host_threads():
    import os
    env = os.environ.get("NSFEM_HOST_THREADS")
    return max(1, int(env)) if env else max(1, min(16, os.cpu_count() or 1))


def _matmul_rows_parallel(A, B, min_rows=20000):
    """A @ B for CSR matrices, the rows of A split into contiguous blocks over host threads (set-up time only).
    Same result as ``A @ B`` bit for bit: a row of the product depends on that row of A alone."""
    import scipy.sparse as sp
    from concurrent.futures import ThreadPoolExecutor
    A = A.tocsr()
    B = B.tocsr()
    nt = min(_host_threads(), max(1, A.shape[0] // min_rows))
    if nt <= 1:
        C = (A @ B).tocsr()
        C.sort_indices()
        return C
    # blocks of (nearly) equal work: rows weighted by their number of entries
    work = np.concatenate([[0], np.cumsum(np.diff(A.indptr) + 1)])
    cuts = np.searchsorted(work, np.linspace(0, work[-1], nt + 1)[1:-1])
    bounds = np.unique(np.concatenate([[0], cuts, [A.shape[0]]]))
    def block(k):
        Ck = (A[bounds[k]:bounds[k + 1]] @ B).tocsr()
        Ck.sort_indices()                                  # (inside the worker: the sort is a third of the time)
        return Ck

    with ThreadPoolExecutor(max_workers=nt) as pool:
        parts = list(pool.map(block, range(bounds.size - 1)))
    # (row blocks of one CSR matrix: concatenate the arrays -- scipy's vstack takes longer than the products)
    counts = np.concatenate([np.diff(p.indptr) for p in parts])
    total = int(counts.sum(dtype=np.int64))
    idx_t = np.int64 if total > np.iinfo(np.int32).max else np.int32
    indptr = np.zeros(A.shape[0] + 1, dtype=idx_t)
    np.cumsum(counts, out=indptr[1:])
    data = np.empty(total, dtype=parts[0].data.dtype)
    indices = np.empty(total, dtype=idx_t)
    o = 0
    for p in parts:
        data[o:o + p.nnz] = p.data
        indices[o:o + p.nnz] = p.indices
        o += p.nnz
    C = sp.csr_matrix((A.shape[0], B.shape[1]), dtype=data.dtype)
    C.data, C.indices, C.indptr = data, indices, indptr     # (assembled arrays: no re-validation pass)
    C.has_sorted_indices = True
    return C


def attach_schur_laplacian(ctx, velocity_bc_dofs, part=None):
    """Algebraic pressure Laplacian of the monolithic scheme's Schur-complement preconditioner:
    A_L = D_f diag(M_v)^{-1} D_f^T on the fine P1 space (D_f = divergence block without the
    Dirichlet velocity columns, M_v = P2 mass matrix) and its Galerkin coarsenings P^T A P along
    the hierarchy ``attach_hierarchy`` installed.  Set-up-time host work (scipy sparse products of
    operators exported from the device); the per-step path only sees the resulting CSR levels.
    A_L is singular (constants) exactly when every velocity boundary dof is constrained.

    ``part`` (a partition of partition.py with more than one rank, already attached): every rank
    builds its ADDITIVE part from the columns of the velocity dofs it owns,
        A_r = D[:, owned] diag(1 / m)[owned] D[:, owned]^T        (sum over the ranks = A_L),
    ghost rows included -- all pressure nodes next to an owned velocity dof are local, and so are
    the cells the entries D_ij and m_j of an owned j are integrated over -- and coarsens it with its
    own prolongations.  No rank ever sees a neighbour's matrix entries; the device adds the ghost
    rows of every product at their owners (nsfem_mg_set_schur_mode)."""
    import scipy.sparse as sp
    import _native as nat
    D = ctx.operator_csr(nat.OP_DIV)                                # n_p1 x (dim n_p2), CSR
    m = ctx.operator_diagonal(nat.OP_MASS_P2)                        # (the same values as the exported matrix's diagonal)
    width = D.shape[1] // m.size
    w = np.repeat(1.0 / m, width)                                   # node-interleaved components
    free = np.ones(D.shape[1], dtype=bool)
    free[np.asarray(velocity_bc_dofs, dtype=np.int64)] = False
    w[~free] = 0.0
    additive = part is not None and part.size > 1
    if additive:
        w[~np.repeat(np.asarray(part.p2_owned, dtype=bool), width)] = 0.0
        ctx.mg_set_schur_mode(True)

    def with_diagonal(A):
        """every row stores its diagonal (ghost rows of a rank part may not touch it)"""
        B = (A + sp.identity(A.shape[0], format="csr")).tocsr()
        B.setdiag(B.diagonal() - 1.0)
        B.sort_indices()
        return B

    # D W D^T with the column scaling applied to D's stored values (no diagonal-matrix product) and the sparse
    # products split into row blocks over host threads (scipy's SpGEMM releases the GIL; every row is formed by
    # the same code in the same order: bit for bit the serial product)
    Dw = sp.csr_matrix((D.data * w[D.indices], D.indices, D.indptr), shape=D.shape)
    A = _matmul_rows_parallel(Dw, D.T.tocsr())
    ones = np.ones(A.shape[0])
    local = float(np.abs(A @ ones).max() / max(np.abs(A.diagonal()).max(), 1e-300))
    singular = bool(ctx.comm_allreduce([local], "max")[0] <= 1e-10)
    ctx.mg_set_schur_operator(0, with_diagonal(A) if additive else A, singular)
    for l, (n_coarse, (rowptr, col, val)) in enumerate(ctx.mg_prolongations):
        P = sp.csr_matrix((val, col, rowptr), shape=(A.shape[0], n_coarse))
        A = _matmul_rows_parallel(P.T.tocsr(), _matmul_rows_parallel(A, P))
        ctx.mg_set_schur_operator(l + 1, with_diagonal(A) if additive else A, singular)
    return singular


# ---------------------------------------------------------------------------------------------
# hierarchies of general (unstructured) triangle meshes by uniform refinement
# ---------------------------------------------------------------------------------------------
def refine_uniform(mesh, markers=None, project=None):
    """Red refinement: one new vertex per edge (its midpoint, optionally moved by
    ``project(mesh, markers, midpoints) -> midpoints`` to follow a curved boundary), four children
    per cell.
    Old vertices keep their ids, the midpoint of edge e gets id nv + e, so the P1 prolongation is
    identity on old vertices and (1/2, 1/2) on new ones.  Facet markers are inherited by the two
    children of every marked edge.  Returns (fine mesh, fine markers | None, prolongation CSR)."""
    from fem_mesh import FacetMarkers, Mesh
    nv, ne = mesh.num_vertices(), mesh.num_edges()
    mid = mesh.edge_midpoints()
    if project is not None:
        mid = np.asarray(project(mesh, markers, mid.copy()), dtype=np.float64)
    coords = np.concatenate([mesh.coords, mid], axis=0)
    c = mesh.cells.astype(np.int64)
    m = nv + mesh.cell_edges.astype(np.int64)          # m[:, k] = midpoint of the edge opposite v_k
    v0, v1, v2 = c[:, 0], c[:, 1], c[:, 2]
    m0, m1, m2 = m[:, 0], m[:, 1], m[:, 2]
    cells = np.stack([np.stack([v0, m2, m1], 1), np.stack([m2, v1, m0], 1),
                      np.stack([m1, m0, v2], 1), np.stack([m0, m1, m2], 1)], axis=1).reshape(-1, 3)
    fine = Mesh(coords, cells.astype(np.int32))
    rowptr = np.concatenate([np.arange(nv + 1), nv + 2 * np.arange(1, ne + 1)]).astype(np.int32)
    col = np.concatenate([np.arange(nv), mesh.edges.astype(np.int64).ravel()]).astype(np.int32)
    val = np.concatenate([np.ones(nv), np.full(2 * ne, 0.5)])
    fine_markers = None
    if markers is not None:
        fine_markers = FacetMarkers(fine, 0)
        nvf = fine.num_vertices()
        fkey = fine.edges[:, 0].astype(np.int64) * nvf + fine.edges[:, 1]       # sorted (np.unique)
        e = mesh.edges.astype(np.int64)
        for end in (0, 1):
            a, b = e[:, end], nv + np.arange(ne)
            key = np.minimum(a, b) * nvf + np.maximum(a, b)
            pos = np.searchsorted(fkey, key)
            assert np.array_equal(fkey[pos], key)
            fine_markers.values[pos] = markers.values
    return fine, fine_markers, (rowptr, col, val)


def refinement_hierarchy(coarse_mesh, coarse_markers, n_refine, project=None):
    """Refine ``n_refine`` times; the finest mesh carries ``mg_levels`` (finest-first list of
    (coarse mesh, prolongation)) which ``attach_hierarchy`` ships to the device.  With a boundary
    projection the spaces are only approximately nested: the coarse operators (integrated on the
    coarse meshes) are then not exactly Galerkin, which is fine for a preconditioner."""
    meshes, marks, prolongs = [coarse_mesh], [coarse_markers], []
    for _ in range(n_refine):
        f, fm, P = refine_uniform(meshes[-1], marks[-1], project)
        meshes.append(f)
        marks.append(fm)
        prolongs.append(P)
    fine = meshes[-1]
    fine.mg_levels = [(meshes[l], prolongs[l]) for l in range(n_refine - 1, -1, -1)]
    return fine, marks[-1]
