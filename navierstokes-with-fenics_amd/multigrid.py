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


def structured_hierarchy(p0, p1, nx, ny, coarsest=16):
    """[(coarse mesh, prolongation CSR to the next finer mesh), ...] finest-first, stopping
    when a direction becomes odd or smaller than ``coarsest`` cells."""
    levels = []
    while nx % 2 == 0 and ny % 2 == 0 and min(nx, ny) // 2 >= coarsest:
        P = structured_prolongation(nx, ny)
        nx, ny = nx // 2, ny // 2
        levels.append((rectangle_mesh(p0, p1, nx, ny), P))
    return levels


def attach_hierarchy(ctx, mesh, degree=2, eig_ratio=4.0, coarsest=16):
    """Build the hierarchy of a structured mesh (``mesh.structured`` = (p0, p1, nx, ny)) on the
    device context.  Returns the number of coarse P1 levels (0: mesh cannot be coarsened; the
    two-level P2 -> P1 hierarchy is still built)."""
    info = getattr(mesh, "structured", None) if mesh is not None else None
    levels = structured_hierarchy(*info, coarsest=coarsest) if info is not None else []
    for coarse_mesh, (rowptr, col, val) in levels:
        ctx.mg_add_level(coarse_mesh.coords, coarse_mesh.cells, rowptr, col, val)
    ctx.mg_finalize(degree, eig_ratio)
    return len(levels)
