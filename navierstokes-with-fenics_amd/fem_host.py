"""Small host-side (numpy) helpers of the solver surface: set-up time work that the
reference does through dolfin ``project`` / ``assemble(ds)`` / facet loops and that
is not part of the per-step device path:

  * right-hand sides of the L2 projections of initial conditions
    (dlfn.project, source/ns_solver_base.py:1151,1168) -- the mass solve itself
    runs on the GPU (nsfem_mass_solve);
  * boundary traction vectors  int_Gamma t . w ds  (source/ns_solver_base.py:142-155);
  * boundary normals / marker sets (source/auxiliary_methods.py:8-67).
"""
import numpy as np


def conical_rule(n, dim=2):
    """Gauss-Legendre conical (Duffy) product rule on the reference triangle / tetrahedron:
    (points [q, dim], weights [q])."""
    g, w = np.polynomial.legendre.leggauss(n)
    g, w = 0.5 * (g + 1.0), 0.5 * w
    if dim == 2:
        xi = np.repeat(g, n)
        eta = np.tile(g, n) * (1.0 - xi)
        wt = np.repeat(w, n) * np.tile(w, n) * (1.0 - xi)
        return np.stack([xi, eta], axis=1), wt
    a, b, c = np.meshgrid(g, g, g, indexing="ij")
    wa, wb, wc = np.meshgrid(w, w, w, indexing="ij")
    pts = np.stack([a, b * (1.0 - a), c * (1.0 - a) * (1.0 - b)], axis=-1).reshape(-1, 3)
    return pts, (wa * wb * wc * (1.0 - a) ** 2 * (1.0 - b)).ravel()


_EDGE_PAIRS = {2: ((1, 2), (0, 2), (0, 1)), 3: ((2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1))}


def _p1_shape(pts):
    return np.concatenate([1.0 - pts.sum(axis=1, keepdims=True), pts], axis=1)


def _p2_shape(pts):
    lam = _p1_shape(pts)
    dim = pts.shape[1]
    return np.stack([lam[:, i] * (2 * lam[:, i] - 1) for i in range(dim + 1)]
                    + [4 * lam[:, a] * lam[:, b] for a, b in _EDGE_PAIRS[dim]], axis=1)


def load_vector(mesh, cell_dofs, n_dofs, fun, degree=2, n_comp=1, quad_n=None):
    """b_i = int f phi_i  for P1 (degree 1) or P2 (degree 2) scalar shape functions on
    triangles / tetrahedra; ``fun(X) -> [n] or [n, n_comp]``; vector results are
    node-interleaved."""
    dim = mesh.coords.shape[1]
    if quad_n is None:          # conical Gauss rule: 36 points (degree 11) in 2D, 64 (degree 7) in 3D
        quad_n = 6 if dim == 2 else 4
    pts, wt = conical_rule(quad_n, dim)
    N = _p2_shape(pts) if degree == 2 else _p1_shape(pts)
    lam = _p1_shape(pts)                                            # [q, dim+1]
    wN = (wt[:, None] * N).T.copy()                                 # [nloc, q]
    cells = mesh.cells.astype(np.int64)
    b = np.zeros(n_dofs * n_comp)
    comp = np.arange(n_comp)[None, None, :]
    chunk = max(1, 4_000_000 // pts.shape[0])                       # bounds the [cells, q, dim] temporaries
    for c0 in range(0, cells.shape[0], chunk):
        x = mesh.coords[cells[c0: c0 + chunk]]                      # [c, dim+1, dim]
        nc = x.shape[0]
        X = np.einsum("qv,cvd->qcd", lam, x)                        # [q, c, dim]
        det = np.abs(np.linalg.det(x[:, 1:] - x[:, :1]))
        f = np.asarray(fun(X.reshape(-1, dim)), dtype=np.float64).reshape(pts.shape[0], nc * n_comp)
        # be[i, c, a] = det_c sum_q w_q N_qi f_qca : one GEMM per chunk
        be = (wN @ f).reshape(N.shape[1], nc, n_comp) * det[None, :, None]
        idx = n_comp * cell_dofs[c0: c0 + chunk].astype(np.int64).T[:, :, None] + comp     # [nloc, c, a]
        b += np.bincount(idx.ravel(), weights=be.ravel(), minlength=b.size)
    return b


# exact P2 mass matrix of a unit-length edge, local order (end, end, midpoint)
_EDGE_MASS = np.array([[4.0, -1.0, 2.0], [-1.0, 4.0, 2.0], [2.0, 2.0, 16.0]]) / 30.0
# exact P2 mass matrix of a unit-area triangle, local order (v0, v1, v2, e(v1v2), e(v0v2), e(v0v1))
_FACE_MASS = np.array([[6, -1, -1, -4, 0, 0], [-1, 6, -1, 0, -4, 0], [-1, -1, 6, 0, 0, -4],
                       [-4, 0, 0, 32, 16, 16], [0, -4, 0, 16, 32, 16], [0, 0, -4, 16, 16, 32]]) / 180.0


def traction_vector(dofmap, facet_ids, values_at_nodes):
    """int_Gamma t . w ds with t interpolated at the P2 nodes of every facet (3 on an edge, 6 on
    a face).  values_at_nodes(X [m, dim]) -> [m, dim]."""
    dim = dofmap.dim
    nodes = dofmap.facet_p2_nodes(facet_ids)                         # [nf, 3 | 6]
    X = dofmap.p2_coords[nodes.ravel()]
    t = np.asarray(values_at_nodes(X), dtype=np.float64).reshape(nodes.shape[0], nodes.shape[1], dim)
    # facet measures from the MESH geometry (dof coordinates are those of the periodic master and
    # would stretch a facet that touches the periodic seam)
    mesh = dofmap.mesh
    xf = mesh.coords[mesh.facets[facet_ids].astype(np.int64)]
    e1 = xf[:, 1] - xf[:, 0]
    if dim == 2:
        measure, M = np.sqrt((e1 * e1).sum(axis=1)), _EDGE_MASS
    else:
        e2 = xf[:, 2] - xf[:, 0]
        measure, M = 0.5 * np.linalg.norm(np.cross(e1, e2), axis=1), _FACE_MASS
    be = np.einsum("f,ij,fja->fia", measure, M, t)
    b = np.zeros(dofmap.n_velocity)
    idx = dim * nodes[:, :, None] + np.arange(dim)[None, None, :]
    np.add.at(b, idx.ravel(), be.ravel())
    return b


def boundary_normal(mesh, markers, boundary_id):
    """Common outward unit normal of a flat boundary part (tuple); asserts flatness."""
    facets = markers.facets_with_id(boundary_id)
    facets = facets[mesh.facet_on_boundary[facets]]
    assert facets.size > 0, "Boundary id {0} was not found".format(boundary_id)
    normals = mesh.facet_normals(facets)
    assert np.abs(normals - normals[0]).max() < 5.0e-14, "boundary is not flat"
    return tuple(float(v) for v in normals[0])


def extract_all_boundary_markers(mesh, markers):
    return markers.ids(boundary_only=True)
