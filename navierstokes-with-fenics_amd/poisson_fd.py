"""Fast diagonalisation of the pressure Poisson operator on tensor-product lattices.

The reference solves the projection step (grad p, grad q) = ... with a sparse LU factorisation every step
(source/ns_ipcs_solver.py:160-171).  On the right-diagonal triangulation of a rectangle (``fem_mesh.rectangle_mesh``,
any line spacing) the P1 stiffness matrix is EXACTLY the tensor sum

    A = K_y (x) W_x + W_y (x) K_x        (node id = j (n_x + 1) + i)

of the 1D P1 stiffness matrices K and the 1D lumped (trapezoid) mass matrices W of the two line meshes -- the cross
terms of the two triangles of a cell cancel.  With the generalised eigenpairs  K v = lambda W v  (V^T W V = I) of
each direction

    A^-1 = (V_y (x) V_x) diag(1 / (lambda_y,j + lambda_x,i)) (V_y (x) V_x)^T ,

i.e. four dense products with (n + 1)-sized matrices: the direct solve the device runs in ``csrc/fastdiag.hip``
(GEMM-shaped work on the matrix cores) instead of a multigrid-preconditioned CG iteration.  Dirichlet conditions on
whole sides drop the side's line from that direction's eigenproblem; the all-Neumann operator is singular
(lambda = 0 + 0): that mode's coefficient is set to zero, the solution is the one without a constant mode in the
W (x) W inner product (pressures are compared modulo a constant, as everywhere else)."""
import numpy as np


def line_matrices(x):
    """1D P1 stiffness matrix (dense) and lumped mass (trapezoid weights) of the line mesh with nodes x"""
    x = np.asarray(x, dtype=np.float64)
    h = np.diff(x)
    assert h.size >= 1 and (h > 0.0).all()
    n = x.size
    K = np.zeros((n, n))
    idx = np.arange(n - 1)
    K[idx, idx] += 1.0 / h
    K[idx + 1, idx + 1] += 1.0 / h
    K[idx, idx + 1] -= 1.0 / h
    K[idx + 1, idx] -= 1.0 / h
    w = np.zeros(n)
    w[:-1] += 0.5 * h
    w[1:] += 0.5 * h
    return K, w


def line_eigenpairs(x, dirichlet_first=False, dirichlet_last=False):
    """(V, lam): K v = lam W v on the free nodes of the line, V^T W V = I; rows of Dirichlet end nodes are zero and
    their eigenvalue slots carry lam = inf (coefficient 0), so V stays square"""
    K, w = line_matrices(x)
    n = w.size
    free = np.ones(n, dtype=bool)
    free[0] = not dirichlet_first
    free[-1] = not dirichlet_last
    f = np.where(free)[0]
    s = 1.0 / np.sqrt(w[f])
    lam_f, Q = np.linalg.eigh(s[:, None] * K[np.ix_(f, f)] * s[None, :])
    V = np.zeros((n, n))
    lam = np.full(n, np.inf)
    V[np.ix_(f, np.arange(f.size))] = s[:, None] * Q
    lam[:f.size] = np.maximum(lam_f, 0.0)
    if not dirichlet_first and not dirichlet_last:
        lam[0] = 0.0                       # the constant: exactly singular (eigh returns ~1e-16)
    return V, lam


def side_pattern(W, H, dirichlet_nodes):
    """Is the Dirichlet node set of a W x H lattice a union of whole sides?  -> (x_first, x_last, y_first, y_last)
    flags, or None when it is not (then the fast solver does not apply)"""
    mask = np.zeros(W * H, dtype=bool)
    mask[np.asarray(dirichlet_nodes, dtype=np.int64)] = True
    m = mask.reshape(H, W)
    flags = (bool(m[:, 0].all()), bool(m[:, -1].all()), bool(m[0, :].all()), bool(m[-1, :].all()))
    want = np.zeros((H, W), dtype=bool)
    if flags[0]:
        want[:, 0] = True
    if flags[1]:
        want[:, -1] = True
    if flags[2]:
        want[0, :] = True
    if flags[3]:
        want[-1, :] = True
    return flags if np.array_equal(want, m) else None


def factors(xs, ys, dirichlet_nodes=()):
    """dict(Vx, Vy, inv) of the W x H lattice with line coordinates xs, ys, or None when the Dirichlet set is not a
    union of whole sides.  inv[j, i] = 1 / (lam_y[j] + lam_x[i]), 0 for the singular mode and the Dirichlet slots"""
    W, H = len(xs), len(ys)
    flags = side_pattern(W, H, dirichlet_nodes)
    if flags is None:
        return None
    Vx, lx = line_eigenpairs(xs, flags[0], flags[1])
    Vy, ly = line_eigenpairs(ys, flags[2], flags[3])
    s = ly[:, None] + lx[None, :]
    with np.errstate(divide="ignore"):
        inv = np.where(np.isfinite(s) & (s > 0.0), 1.0 / np.where(s > 0.0, s, 1.0), 0.0)
    scale = s[np.isfinite(s)].max()
    inv[s <= 1e-13 * scale] = 0.0          # the constant mode of the all-Neumann operator
    return dict(Vx=np.ascontiguousarray(Vx), Vy=np.ascontiguousarray(Vy), inv=np.ascontiguousarray(inv),
                singular=not any(flags))


def apply_reference(f, r):
    """z = A^+ r in numpy (the sums the device kernels compute): r, z of length W * H, node id j W + i"""
    H, W = f["inv"].shape
    R = np.asarray(r, dtype=np.float64).reshape(H, W)
    U = f["Vy"].T @ (R @ f["Vx"])
    U *= f["inv"]
    return (f["Vy"] @ (U @ f["Vx"].T)).ravel()


def lattice_lines(mesh):
    """(xs, ys) when the mesh is a rectangle_mesh lattice (vertex id = j (n_x + 1) + i, any line spacing), else None"""
    info = getattr(mesh, "structured", None)
    if info is None or len(info) != 4:
        return None
    nx, ny = int(info[2]), int(info[3])
    X = np.asarray(mesh.coords, dtype=np.float64)
    if X.shape != ((nx + 1) * (ny + 1), 2):
        return None
    G = X.reshape(ny + 1, nx + 1, 2)
    xs, ys = G[0, :, 0].copy(), G[:, 0, 1].copy()
    if not (np.abs(G[:, :, 0] - xs[None, :]).max() <= 1e-14 * max(1.0, np.abs(xs).max()) and
            np.abs(G[:, :, 1] - ys[:, None]).max() <= 1e-14 * max(1.0, np.abs(ys).max())):
        return None
    return xs, ys
