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


def conical_rule(n):
    """Gauss-Legendre (n x n) conical product rule on the reference triangle."""
    g, w = np.polynomial.legendre.leggauss(n)
    g, w = 0.5 * (g + 1.0), 0.5 * w
    xi = np.repeat(g, n)
    eta = np.tile(g, n) * (1.0 - xi)
    wt = np.repeat(w, n) * np.tile(w, n) * (1.0 - xi)
    return xi, eta, wt


def _p2_shape(xi, eta):
    l0, l1, l2 = 1.0 - xi - eta, xi, eta
    return np.stack([l0 * (2 * l0 - 1), l1 * (2 * l1 - 1), l2 * (2 * l2 - 1),
                     4 * l1 * l2, 4 * l0 * l2, 4 * l0 * l1], axis=1)


def _p1_shape(xi, eta):
    return np.stack([1.0 - xi - eta, xi, eta], axis=1)


def load_vector(mesh, cell_dofs, n_dofs, fun, degree=2, n_comp=1, quad_n=6):
    """b_i = int f phi_i  for P1 (degree 1) or P2 (degree 2) scalar shape functions;
    ``fun(X) -> [n] or [n, n_comp]``; vector results are node-interleaved."""
    xi, eta, wt = conical_rule(quad_n)
    N = _p2_shape(xi, eta) if degree == 2 else _p1_shape(xi, eta)
    x = mesh.coords[mesh.cells.astype(np.int64)]                   # [c, 3, 2]
    lam = _p1_shape(xi, eta)                                        # [q, 3]
    X = np.einsum("qv,cvd->cqd", lam, x)                            # [c, q, 2]
    det = np.abs((x[:, 1, 0] - x[:, 0, 0]) * (x[:, 2, 1] - x[:, 0, 1])
                 - (x[:, 2, 0] - x[:, 0, 0]) * (x[:, 1, 1] - x[:, 0, 1]))
    f = np.asarray(fun(X.reshape(-1, 2)), dtype=np.float64).reshape(X.shape[0], X.shape[1], -1)
    be = np.einsum("c,q,qi,cqa->cia", det, wt, N, f)                # [c, nloc, n_comp]
    b = np.zeros(n_dofs * n_comp)
    idx = n_comp * cell_dofs.astype(np.int64)[:, :, None] + np.arange(n_comp)[None, None, :]
    np.add.at(b, idx.ravel(), be.ravel())
    return b


# exact P2 mass matrix of a unit-length edge, local order (end, end, midpoint)
_EDGE_MASS = np.array([[4.0, -1.0, 2.0], [-1.0, 4.0, 2.0], [2.0, 2.0, 16.0]]) / 30.0


def traction_vector(dofmap, facet_ids, values_at_nodes):
    """int_Gamma t . w ds with t interpolated at the three P2 nodes of every facet.
    values_at_nodes(X [m, 2]) -> [m, 2]."""
    nodes = dofmap.facet_p2_nodes(facet_ids)                         # [nf, 3]
    X = dofmap.p2_coords[nodes.ravel()]
    t = np.asarray(values_at_nodes(X), dtype=np.float64).reshape(nodes.shape[0], 3, 2)
    ends = dofmap.p2_coords[nodes[:, 1]] - dofmap.p2_coords[nodes[:, 0]]
    length = np.sqrt((ends * ends).sum(axis=1))
    be = np.einsum("f,ij,fja->fia", length, _EDGE_MASS, t)
    b = np.zeros(dofmap.n_velocity)
    idx = 2 * nodes[:, :, None] + np.arange(2)[None, None, :]
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
