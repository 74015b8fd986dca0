"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Plain numpy/scipy restatement of the arithmetic that the reference hands to
FEniCS/DOLFIN 2019.1.0 + PETSc LU for the per-time-step Taylor-Hood (P2/P1)
assembly + sparse solve.  The arithmetic itself lives in third-party packages
that are NOT under /root/reference (fenics-dolfin/ffc/fiat/ufl 2019.1.0, petsc
3.14.5 -- environment.yml:19-25,75), so this file restates the published
algorithm (Lagrange P2/P1 on affine simplices, exact Gauss quadrature, Newton
with residual criterion, DirichletBC row replacement, sparse direct LU) and is
anchored on the reference's own call sites:

  forms (IPCS)     source/ns_ipcs_solver.py:19-33,106-196
  forms (BDF)      source/ns_bdf_solver.py:19-34,54-100
  term builders    source/ns_solver_base.py:121-191,370-399,662-673
  Newton control   source/ns_bdf_solver.py:96-100, source/ns_ipcs_solver.py:143-147
  step order       source/ns_ipcs_solver.py:198-208, source/ns_solver_base.py:1174-1203
  time shifting    source/ns_solver_base.py:1012-1016, source/ns_ipcs_solver.py:35-43

PARITY PINNING: the reference's tests hold no numeric fixture for this path
(its solver tests are smoke tests).  The oracle is pinned
by (i) the reference's BDF coefficient tables (tests/test_bdf_time_stepping.py
:92-114) through tests/golden/bdf_tables.json, (ii) sympy-exact element
matrices, and (iii) analytic known answers built from the reference's own test
inputs (Poiseuille channel of tests/test_ipcs_solver.py:37-43, Taylor-Green of
convergence_test/taylor_green_vortex.py:111-117).  Velocity/pressure values of
FEniCS itself cannot be diffed here: "parity unpinned" against FEniCS output.

Conventions shared with the product (inputs, not algorithm):
  * triangles and tetrahedra (``dim`` = coords.shape[1]); scalar P2 node ids come from
    ``p2_dofmap`` [n_cells, 6 | 10]: vertices, then edge nodes in FIAT/UFC edge order
    (2D: e(v1v2), e(v0v2), e(v0v1); 3D: e(v2v3), e(v1v3), e(v1v2), e(v0v3), e(v0v2), e(v0v1));
  * velocity vectors are node-interleaved: index = dim * node + component;
  * mixed vectors are [velocity (dim*N2) | pressure (N1)].
The 3D branch is pinned by sympy-exact tetrahedron matrices and a polynomial Stokes solution
(tests/test_oracle_pinning.py); the reference itself never runs in 3D.
A quadrature rule DIFFERENT from the one in the HIP kernels is used on purpose
(collapsed Gauss-Legendre instead of the 7-point Radon rule): all integrands
are polynomials on affine cells, so both are exact and must agree to round-off.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


# --------------------------------------------------------------------------
# reference element
# --------------------------------------------------------------------------
def collapsed_gauss_rule(n, dim=2):
    """n^dim Gauss-Legendre rule collapsed (Duffy) onto the reference simplex; exact for total
    degree <= 2n-1-(dim-1) at least (the Jacobian factors cost one degree per collapse)."""
    x, w = np.polynomial.legendre.leggauss(n)
    x = 0.5 * (x + 1.0)
    w = 0.5 * w
    if dim == 2:
        a, b = np.meshgrid(x, x, indexing="ij")
        wa, wb = np.meshgrid(w, w, indexing="ij")
        xi = a.ravel()
        eta = (b * (1.0 - a)).ravel()
        wt = (wa * wb * (1.0 - a)).ravel()
        return np.stack([xi, eta], axis=1), wt
    a, b, c = np.meshgrid(x, x, x, indexing="ij")
    wa, wb, wc = np.meshgrid(w, w, w, indexing="ij")
    xi = a
    eta = b * (1.0 - a)
    zeta = c * (1.0 - a) * (1.0 - b)
    wt = wa * wb * wc * (1.0 - a) ** 2 * (1.0 - b)
    return np.stack([xi.ravel(), eta.ravel(), zeta.ravel()], axis=1), wt.ravel()


# local edges (pairs of local vertices) carrying the P2 edge nodes, UFC order
_EDGE_PAIRS = {2: ((1, 2), (0, 2), (0, 1)),
               3: ((2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1))}


def p1_basis(pts):
    dim = pts.shape[1]
    lam0 = 1.0 - pts.sum(axis=1)
    phi = np.concatenate([lam0[:, None], pts], axis=1)            # [q, dim+1]
    dphi = np.zeros((pts.shape[0], dim + 1, dim))
    dphi[:, 0, :] = -1.0
    for a in range(dim):
        dphi[:, a + 1, a] = 1.0
    return phi, dphi


def p2_basis(pts):
    dim = pts.shape[1]
    lam, dlam = p1_basis(pts)
    q = pts.shape[0]
    pairs = _EDGE_PAIRS[dim]
    phi = np.zeros((q, dim + 1 + len(pairs)))
    dphi = np.zeros((q, dim + 1 + len(pairs), dim))
    for i in range(dim + 1):
        phi[:, i] = lam[:, i] * (2.0 * lam[:, i] - 1.0)
        dphi[:, i, :] = (4.0 * lam[:, i] - 1.0)[:, None] * dlam[:, i, :]
    for e, (a, b) in enumerate(pairs):
        phi[:, dim + 1 + e] = 4.0 * lam[:, a] * lam[:, b]
        dphi[:, dim + 1 + e, :] = 4.0 * (lam[:, a, None] * dlam[:, b, :] + lam[:, b, None] * dlam[:, a, :])
    return phi, dphi


_LEVI_CIVITA = np.zeros((3, 3, 3))
for _i, _j, _k in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
    _LEVI_CIVITA[_i, _j, _k] = 1.0
    _LEVI_CIVITA[_i, _k, _j] = -1.0


class Geometry:
    """Affine maps of all cells: detJ (absolute), J^{-T}."""

    def __init__(self, coords, cells):
        x = np.asarray(coords, dtype=np.float64)[np.asarray(cells)]   # [c, dim+1, dim]
        J = np.stack([x[:, k + 1] - x[:, 0] for k in range(x.shape[2])], axis=2)   # columns
        self.x = x
        self.absdet = np.abs(np.linalg.det(J))
        self.JinvT = np.transpose(np.linalg.inv(J), (0, 2, 1))
        self.n_cells = x.shape[0]

    def phys_grad(self, dphi):
        """dphi [q, n, dim] (reference) -> [c, q, n, dim] (physical)."""
        return np.einsum("cab,qnb->cqna", self.JinvT, dphi)


class Space:
    """Problem description handed to every oracle routine (plain arrays)."""

    def __init__(self, coords, cells, p2_dofmap, p1_dofmap, quad_n=5):
        self.coords = np.asarray(coords, dtype=np.float64)
        self.dim = dim = int(self.coords.shape[1])
        self.cells = np.asarray(cells, dtype=np.int64)
        self.p2 = np.asarray(p2_dofmap, dtype=np.int64)
        self.p1 = np.asarray(p1_dofmap, dtype=np.int64)
        self.n2 = int(self.p2.max()) + 1
        self.n1 = int(self.p1.max()) + 1
        self.geo = Geometry(self.coords, self.cells)
        self.pts, self.wts = collapsed_gauss_rule(quad_n, dim)
        self.phi2, dphi2 = p2_basis(self.pts)
        self.phi1, dphi1 = p1_basis(self.pts)
        self.g2 = self.geo.phys_grad(dphi2)      # [c, q, 6, 2]
        self.g1 = self.geo.phys_grad(dphi1)      # [c, q, 3, 2]
        self.wdet = self.geo.absdet[:, None] * self.wts[None, :]   # [c, q]
        # velocity (interleaved) cell dof map [c, n_loc, dim]
        self.vdof = dim * self.p2[:, :, None] + np.arange(dim)[None, None, :]

    # -- helpers -----------------------------------------------------------
    def _coo(self, vals, rows, cols, shape):
        rr = np.broadcast_to(rows, vals.shape).ravel()
        cc = np.broadcast_to(cols, vals.shape).ravel()
        A = sp.coo_matrix((vals.ravel(), (rr, cc)), shape=shape).tocsr()
        A.sum_duplicates()
        return A

    def p2_nodes(self):
        """Coordinates of the scalar P2 nodes [n2, dim] (vertices, edge midpoints)."""
        out = np.zeros((self.n2, self.dim))
        x = self.geo.x
        for i in range(self.dim + 1):
            out[self.p2[:, i]] = x[:, i]
        for e, (a, b) in enumerate(_EDGE_PAIRS[self.dim]):
            out[self.p2[:, self.dim + 1 + e]] = 0.5 * (x[:, a] + x[:, b])
        return out

    def p1_nodes(self):
        out = np.zeros((self.n1, self.dim))
        for i in range(self.dim + 1):
            out[self.p1[:, i]] = self.geo.x[:, i]
        return out

    # -- constant operators ---------------------------------------------------
    def mass_p2(self):
        Me = np.einsum("cq,qi,qj->cij", self.wdet, self.phi2, self.phi2)
        return self._coo(Me, self.p2[:, :, None], self.p2[:, None, :], (self.n2, self.n2))

    def stiffness_p2(self):
        Ke = np.einsum("cq,cqia,cqja->cij", self.wdet, self.g2, self.g2)
        return self._coo(Ke, self.p2[:, :, None], self.p2[:, None, :], (self.n2, self.n2))

    def mass_p1(self):
        Me = np.einsum("cq,qi,qj->cij", self.wdet, self.phi1, self.phi1)
        return self._coo(Me, self.p1[:, :, None], self.p1[:, None, :], (self.n1, self.n1))

    def stiffness_p1(self):
        """(grad p, grad q)  -- source/ns_ipcs_solver.py:160."""
        Ke = np.einsum("cq,cqia,cqja->cij", self.wdet, self.g1, self.g1)
        return self._coo(Ke, self.p1[:, :, None], self.p1[:, None, :], (self.n1, self.n1))

    def vector_mass(self):
        """(v, w) on the interleaved P2^2 space -- source/ns_ipcs_solver.py:183."""
        return sp.kron(self.mass_p2(), sp.identity(self.dim), format="csr")

    def vector_stiffness(self, traction_form=False):
        """inner(grad u, grad v)  or  inner(grad u + grad u^T, sym grad v)
        -- source/ns_solver_base.py:669-673."""
        K = sp.kron(self.stiffness_p2(), sp.identity(self.dim), format="csr")
        if not traction_form:
            return K
        # extra term  sum_ab d_a u_b d_b v_a  ->  rows (i,a) cols (j,b): int d_b phi_i d_a phi_j
        Ke = np.einsum("cq,cqib,cqja->ciajb", self.wdet, self.g2, self.g2)
        rows = self.vdof[:, :, :, None, None]
        cols = self.vdof[:, None, None, :, :]
        E = self._coo(Ke, rows, cols, (self.dim * self.n2, self.dim * self.n2))
        return (K + E).tocsr()

    def divergence(self):
        """D[i, (j,a)] = int psi_i d_a phi_j   (q, div u): source/ns_solver_base.py:399."""
        De = np.einsum("cq,qi,cqja->cija", self.wdet, self.phi1, self.g2)
        rows = self.p1[:, :, None, None]
        cols = self.vdof[:, None, :, :]
        return self._coo(De, rows, cols, (self.n1, self.dim * self.n2))

    def pressure_gradient(self):
        """G[(i,a), j] = int phi_i d_a psi_j   (grad p, w): source/ns_ipcs_solver.py:185."""
        Ge = np.einsum("cq,qi,cqja->ciaj", self.wdet, self.phi2, self.g1)
        rows = self.vdof[:, :, :, None]
        cols = self.p1[:, None, None, :]
        return self._coo(Ge, rows, cols, (self.dim * self.n2, self.n1))

    # -- solution dependent terms -------------------------------------------
    def _u_at_q(self, u):
        ue = u[self.vdof]                                   # [c, 6, 2]
        uq = np.einsum("qk,cka->cqa", self.phi2, ue)        # [c, q, 2]
        gu = np.einsum("cqkb,cka->cqab", self.g2, ue)       # d_b u_a
        return uq, gu

    def convection_residual(self, u, form="standard"):
        """int  c(u) . phi_i  for the four weak forms of source/ns_solver_base.py:370-390."""
        uq, gu = self._u_at_q(u)
        adv = np.einsum("cqab,cqb->cqa", gu, uq)            # (grad u) u
        if form == "standard":
            f = adv
            be = np.einsum("cq,cqa,qi->cia", self.wdet, f, self.phi2)
        elif form == "rotational" and self.dim == 3:
            # cross(curl(u), u), source/ns_solver_base.py:384:  f_a = eps_acd curl_c u_d,
            # curl_c = eps_cef d_e u_f  (gu[f, e] = d_e u_f)
            curl = np.einsum("xef,cqfe->cqx", _LEVI_CIVITA, gu)
            f = np.einsum("axd,cqx,cqd->cqa", _LEVI_CIVITA, curl, uq)
            be = np.einsum("cq,cqa,qi->cia", self.wdet, f, self.phi2)
        elif form == "rotational":
            curl = gu[:, :, 1, 0] - gu[:, :, 0, 1]
            f = np.stack([-curl * uq[:, :, 1], curl * uq[:, :, 0]], axis=2)
            be = np.einsum("cq,cqa,qi->cia", self.wdet, f, self.phi2)
        elif form == "divergence":
            div = np.einsum("cqaa->cq", gu)
            f = adv + 0.5 * div[:, :, None] * uq
            be = np.einsum("cq,cqa,qi->cia", self.wdet, f, self.phi2)
        elif form == "skew_symmetric":
            be = 0.5 * np.einsum("cq,cqa,qi->cia", self.wdet, adv, self.phi2)
            # - 1/2 ((grad v) u) . u  with v = phi_i e_a:  sum_b d_b phi_i u_b u_a
            be -= 0.5 * np.einsum("cq,cqib,cqb,cqa->cia", self.wdet, self.g2, uq, uq)
        else:
            raise ValueError(form)
        b = np.zeros(self.dim * self.n2)
        np.add.at(b, self.vdof.ravel(), be.ravel())
        return b

    def convection_jacobian(self, u, form="standard"):
        """d/du of convection_residual (exact Gateaux derivative = dlfn.derivative)."""
        uq, gu = self._u_at_q(u)
        w, phi, g = self.wdet, self.phi2, self.g2
        d = np.eye(self.dim)
        # building blocks: test (i,a), trial (j,b)
        # T1 = phi_i (u . grad phi_j) delta_ab ; T2 = phi_i phi_j d_b u_a
        udg = np.einsum("cqb,cqjb->cqj", uq, g)             # u . grad phi_j
        T1 = np.einsum("cq,qi,cqj,ab->ciajb", w, phi, udg, d)
        T2 = np.einsum("cq,qi,qj,cqab->ciajb", w, phi, phi, gu)
        if form == "standard":
            Je = T1 + T2
        elif form == "divergence":
            div = np.einsum("cqaa->cq", gu)
            # + 1/2 [ d_b phi_j u_a + div(u) phi_j delta_ab ] phi_i
            T3 = 0.5 * np.einsum("cq,qi,cqjb,cqa->ciajb", w, phi, g, uq)
            T4 = 0.5 * np.einsum("cq,qi,qj,cq,ab->ciajb", w, phi, phi, div, d)
            Je = T1 + T2 + T3 + T4
        elif form == "skew_symmetric":
            # - 1/2 [ d_b' phi_i (phi_j delta_b'b) u_a + d_b' phi_i u_b' phi_j delta_ab ]
            T5 = np.einsum("cq,cqib,qj,cqa->ciajb", w, g, phi, uq)
            udgi = np.einsum("cqb,cqib->cqi", uq, g)
            T6 = np.einsum("cq,cqi,qj,ab->ciajb", w, udgi, phi, d)
            Je = 0.5 * (T1 + T2) - 0.5 * (T5 + T6)
        elif form == "rotational" and self.dim == 3:
            E = _LEVI_CIVITA
            curl = np.einsum("xef,cqfe->cqx", E, gu)
            # d f_a / d u_(j,b) = eps_axb curl_x phi_j + eps_axd eps_xeb d_e phi_j u_d
            Ta = np.einsum("cq,qi,axb,cqx,qj->ciajb", w, phi, E, curl, phi)
            Tb = np.einsum("cq,qi,axd,xeb,cqje,cqd->ciajb", w, phi, E, E, g, uq)
            Je = Ta + Tb
        elif form == "rotational":
            curl = gu[:, :, 1, 0] - gu[:, :, 0, 1]
            # f_a = eps_a * curl * u_{1-a}, eps = (-1, +1)
            # d curl[(j,b)] = (b==1 ? d_0 phi_j : 0) - (b==0 ? d_1 phi_j : 0)
            dcurl = np.stack([-g[:, :, :, 1], g[:, :, :, 0]], axis=3)   # [c,q,j,b]
            eps = np.array([-1.0, 1.0])
            uswap = uq[:, :, ::-1]                                      # u_{1-a}
            Ta = np.einsum("cq,qi,a,cqjb,cqa->ciajb", w, phi, eps, dcurl, uswap)
            swap = np.array([[0.0, 1.0], [1.0, 0.0]])                   # delta_{b,1-a}
            Tb = np.einsum("cq,qi,a,cq,qj,ab->ciajb", w, phi, eps, curl, phi, swap)
            Je = Ta + Tb
        else:
            raise ValueError(form)
        rows = self.vdof[:, :, :, None, None]
        cols = self.vdof[:, None, None, :, :]
        return self._coo(Je, rows, cols, (self.dim * self.n2, self.dim * self.n2))

    def picard_convection(self, u, form="standard"):
        """Picard linearisations of source/ns_solver_base.py:478-499 (trial v = phi_j e_b, test
        w = phi_i e_a): standard (grad v) u . w; rotational cross(curl(u), v) . w; divergence
        (grad v) u . w + 1/2 div(u) v . w; skew 1/2 [(grad v) u . w - (grad w) u . v]."""
        uq, gu = self._u_at_q(u)
        w, phi, g = self.wdet, self.phi2, self.g2
        d = np.eye(self.dim)
        udg = np.einsum("cqb,cqjb->cqj", uq, g)
        T1 = np.einsum("cq,qi,cqj,ab->ciajb", w, phi, udg, d)
        if form == "standard":
            Je = T1
        elif form == "rotational" and self.dim == 3:
            curl = np.einsum("xef,cqfe->cqx", _LEVI_CIVITA, gu)
            Je = np.einsum("cq,qi,axb,cqx,qj->ciajb", w, phi, _LEVI_CIVITA, curl, phi)
        elif form == "rotational":
            curl = gu[:, :, 1, 0] - gu[:, :, 0, 1]
            rot = np.array([[0.0, -1.0], [1.0, 0.0]])                  # (-curl v_1, curl v_0)
            Je = np.einsum("cq,qi,qj,cq,ab->ciajb", w, phi, phi, curl, rot)
        elif form == "divergence":
            div = np.einsum("cqaa->cq", gu)
            Je = T1 + 0.5 * np.einsum("cq,qi,qj,cq,ab->ciajb", w, phi, phi, div, d)
        elif form == "skew_symmetric":
            udgi = np.einsum("cqb,cqib->cqi", uq, g)
            Je = 0.5 * (T1 - np.einsum("cq,cqi,qj,ab->ciajb", w, udgi, phi, d))
        else:
            raise ValueError(form)
        rows = self.vdof[:, :, :, None, None]
        cols = self.vdof[:, None, None, :, :]
        return self._coo(Je, rows, cols, (self.dim * self.n2, self.dim * self.n2))

    # -- boundary integrals ---------------------------------------------------
    def traction_vector(self, facets, traction_nodal, length=None):
        """int_Gamma t . w ds with t given at the 3 P2 nodes of every facet
        (Expression(degree=2) semantics: nodal interpolation, then exact
        integration) -- source/ns_solver_base.py:142-155.
        facets: [nf, 3] scalar P2 node ids (end, end, midpoint); traction_nodal
        [nf, 3, 2]."""
        facets = np.asarray(facets, dtype=np.int64)
        if length is None:      # (pass the facet lengths explicitly on periodic dof maps)
            nodes = self.p2_nodes()
            length = np.linalg.norm(nodes[facets[:, 1]] - nodes[facets[:, 0]], axis=1)
        # 1D P2 mass matrix on an edge (end, end, mid), Simpson-exact
        M1 = np.array([[4.0, -1.0, 2.0], [-1.0, 4.0, 2.0], [2.0, 2.0, 16.0]]) / 30.0
        be = np.einsum("f,ij,fja->fia", length, M1, traction_nodal)
        b = np.zeros(self.dim * self.n2)
        idx = self.dim * facets[:, :, None] + np.arange(self.dim)[None, None, :]
        np.add.at(b, idx.ravel(), be.ravel())
        return b


def boundary_functionals(space, facet_vertices, facet_cell, u, p, nu, sym):
    """dolfin.assemble(f * ds(subdomain)) of the reference's post-processing hooks restated
    (demo/dfg_benchmark.py:44-66, demo/gravity_driven_flow.py:66-70; FacetNormal = outward unit
    normal):  (int (-p n + nu (grad u + sym grad u^T) n) dS [dim],  int u.n dS,  int dS)
    over the facets given by their vertex ids ``facet_vertices`` [nf, dim] and adjacent cells.
    Route deliberately different from the device kernel: Gauss points on the PHYSICAL facet
    (4-point Gauss on edges, 3 x 3 collapsed Gauss on triangles), pulled back to the cell's
    reference coordinates through J^-1; normal from the facet's own geometry, oriented away from
    the cell centroid."""
    dim = space.dim
    fv = np.asarray(facet_vertices, dtype=np.int64)
    fc = np.asarray(facet_cell, dtype=np.int64)
    X = space.coords[fv]                                       # [nf, dim, dim]
    if dim == 2:
        t = X[:, 1] - X[:, 0]
        meas = np.linalg.norm(t, axis=1)
        nrm = np.stack([t[:, 1], -t[:, 0]], axis=1) / meas[:, None]
        g, w = np.polynomial.legendre.leggauss(4)
        lam = np.stack([0.5 * (1.0 - g), 0.5 * (1.0 + g)], axis=1)       # [q, 2]
        wq = 0.5 * w
    else:
        cr = np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0])
        meas = 0.5 * np.linalg.norm(cr, axis=1)
        nrm = cr / np.linalg.norm(cr, axis=1)[:, None]
        pts, wt = collapsed_gauss_rule(3, 2)
        lam = np.concatenate([(1.0 - pts.sum(axis=1))[:, None], pts], axis=1)   # [q, 3]
        wq = 2.0 * wt
    xc = space.geo.x[fc]                                       # [nf, dim+1, dim]
    centroid = xc.mean(axis=1)
    flip = ((centroid - X.mean(axis=1)) * nrm).sum(axis=1) > 0.0
    nrm[flip] *= -1.0
    xq = np.einsum("qv,fvd->fqd", lam, X)                      # physical points [nf, q, dim]
    Jinv = np.transpose(space.geo.JinvT[fc], (0, 2, 1))        # [nf, dim, dim]
    ref = np.einsum("fab,fqb->fqa", Jinv, xq - xc[:, None, 0, :])
    force = np.zeros(dim)
    flux = 0.0
    ue = u[space.vdof[fc]]                                     # [nf, n2loc, dim]
    pe = p[space.p1[fc]]
    for f in range(fv.shape[0]):
        phi2, dphi2 = p2_basis(ref[f])
        phi1, _ = p1_basis(ref[f])
        g2 = np.einsum("ab,qnb->qna", space.geo.JinvT[fc[f]], dphi2)     # physical gradients
        uq = phi2 @ ue[f]                                      # [q, dim]
        G = np.einsum("qkb,ka->qab", g2, ue[f])                # d_b u_a
        pq = phi1 @ pe[f]
        n = nrm[f]
        tr = -pq[:, None] * n[None, :] + nu * np.einsum("qab,b->qa", G + sym * np.transpose(G, (0, 2, 1)), n)
        force += meas[f] * (wq @ tr)
        flux += meas[f] * (wq @ (uq @ n))
    return force, float(flux), float(meas.sum())


# --------------------------------------------------------------------------
# Dirichlet handling (dolfin DirichletBC.apply semantics, third party)
# --------------------------------------------------------------------------
def strang_fix_6():
    """6-point degree-4 rule (FFC/FIAT 'default' scheme for quadrature degree 4 on triangles,
    which the reference's CFL projection requests: source/ns_problem.py:570)."""
    a, b = 0.445948490915965, 0.091576213509771
    wa, wb = 0.223381589678011, 0.109951743655322
    pts = np.array([[a, a], [1 - 2 * a, a], [a, 1 - 2 * a], [b, b], [1 - 2 * b, b], [b, 1 - 2 * b]])
    return pts, 0.5 * np.array([wa, wa, wa, wb, wb, wb])


def keast_14():
    """14-point degree-4 Keast rule on the reference tetrahedron (FFC/FIAT 'default' scheme for
    quadrature degree 4 on tetrahedra: 6 edge midpoints + two vertex orbits; weights sum to 1/6)."""
    import itertools
    bary, w = [], []
    for i, j in itertools.combinations(range(4), 2):
        l = [0.0] * 4
        l[i] = l[j] = 0.5
        bary.append(l)
        w.append(0.0031746031746032)
    for p, q, wt in ((0.1005267652252045, 0.6984197043243866, 0.0147649707904968),
                     (0.3143728734931922, 0.0568813795204234, 0.0221397911142651)):
        for k in range(4):
            l = [p] * 4
            l[k] = q
            bary.append(l)
            w.append(wt)
    return np.array(bary)[:, 1:], np.array(w)


def circumdiameter(x):
    """dolfin CellDiameter of the simplices x [c, d+1, d]: diameter of the circumscribed
    circle / sphere."""
    d = x.shape[2]
    ln = lambda i, j: np.linalg.norm(x[:, i] - x[:, j], axis=1)
    if d == 2:
        e1, e2 = x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]
        area2 = np.abs(e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0])
        return ln(1, 2) * ln(0, 2) * ln(0, 1) / area2              # abc / (2 A)
    pa, pb, pc = ln(0, 1) * ln(2, 3), ln(0, 2) * ln(1, 3), ln(0, 3) * ln(1, 2)
    P = 0.5 * (pa + pb + pc)
    vol6 = np.abs(np.linalg.det(x[:, 1:] - x[:, :1]))
    return 2.0 * np.sqrt(np.maximum(P * (P - pa) * (P - pb) * (P - pc), 0.0)) / vol6


def cfl_number(space, u, step_size):
    """source/ns_problem.py:554-587 restated: per cell, solve the local DG2 mass system
    M c = int phi_i f  with  f = 2 |u| k / h  (h = dolfin CellDiameter = circumdiameter), both
    sides integrated with the degree-4 rule; return max |c_i| over all cells."""
    pts, wts = strang_fix_6() if space.dim == 2 else keast_14()
    phi, _ = p2_basis(pts)                                     # [q, 6 | 10]
    h = circumdiameter(space.geo.x)
    uq = np.einsum("qk,cka->cqa", phi, u[space.vdof])
    f = 2.0 * np.linalg.norm(uq, axis=2) * step_size / h[:, None]          # [c, q]
    M = np.einsum("q,qi,qj->ij", wts, phi, phi)                # reference mass (volume factor cancels)
    rhs = np.einsum("q,qi,cq->ci", wts, phi, f)
    c = np.linalg.solve(M, rhs.T).T
    return float(np.abs(c).max())


def apply_dirichlet_rows(A, dofs):
    """Zero the rows, put 1 on the diagonal, keep the columns (non-symmetric),
    as dolfin::DirichletBC::apply(A) does."""
    A = A.tolil(copy=True) if False else A.tocsr(copy=True)
    dofs = np.asarray(dofs, dtype=np.int64)
    if dofs.size == 0:
        return A
    mask = np.ones(A.shape[0])
    mask[dofs] = 0.0
    A = sp.diags(mask) @ A
    diag = np.zeros(A.shape[0])
    diag[dofs] = 1.0
    return (A + sp.diags(diag)).tocsr()


def newton_solve(residual, jacobian, x, bc_dofs, bc_vals, atol, rtol, maxit,
                 history=None, linear_solve=None):
    """dolfin::NewtonSolver with the 'residual' criterion and relaxation 1
    (third-party algorithm; parameters from source/ns_bdf_solver.py:96-100).
    Dirichlet rows of the residual are x_i - g_i."""
    bc_dofs = np.asarray(bc_dofs, dtype=np.int64)

    def res(x):
        b = residual(x)
        b[bc_dofs] = x[bc_dofs] - bc_vals
        return b

    b = res(x)
    r0 = np.linalg.norm(b)
    r = r0
    if history is not None:
        history.append(r)
    it = 0
    converged = r < atol        # relative residual is 1 at iteration 0
    while not converged and it < maxit:
        A = apply_dirichlet_rows(jacobian(x), bc_dofs)
        if linear_solve is None:
            dx = spla.splu(A.tocsc()).solve(b)
        else:
            dx = linear_solve(A, b)
        x = x - dx
        it += 1
        b = res(x)
        r = np.linalg.norm(b)
        if history is not None:
            history.append(r)
        converged = (r / r0 < rtol) or (r < atol)
    if not converged:
        raise RuntimeError("Newton solver did not converge (|r| = %.3e after %d its)" % (r, it))
    return x, it


def linear_solve_dirichlet(A, b, bc_dofs, bc_vals, pin_nullspace=False):
    """dolfin LinearVariationalSolver: apply bcs (row replacement, b_i = g_i),
    sparse LU.  ``pin_nullspace`` handles the singular pure-Neumann pressure
    problem the reference never meets in its tests: one dof is
    pinned and results are compared modulo a constant."""
    bc_dofs = np.asarray(bc_dofs, dtype=np.int64)
    b = b.copy()
    if bc_dofs.size:
        A = apply_dirichlet_rows(A, bc_dofs)
        b[bc_dofs] = bc_vals
    elif pin_nullspace:
        A = apply_dirichlet_rows(A, np.array([0]))
        b[0] = 0.0
    return spla.splu(A.tocsc()).solve(b)


# --------------------------------------------------------------------------
# IPCS step (source/ns_ipcs_solver.py)
# --------------------------------------------------------------------------
class IPCSOracle:
    """State and step of the incremental pressure-correction scheme.

    coefficients: dict with convective_term, pressure_term, viscous_term,
    body_force_term (floats or None) -- source/auxiliary_classes.py:251-306.
    """

    def __init__(self, space, coeffs, form="standard", traction_form=False,
                 tol=1e-10, maxit=50, refactor_every_step=True):
        self.s = space
        self.c = coeffs
        self.form = form
        self.tol, self.maxit = tol, maxit
        s = space
        self.M = s.vector_mass()
        self.K = s.vector_stiffness(traction_form)
        self.D = s.divergence()
        self.G = s.pressure_gradient()
        self.Ap = s.stiffness_p1()
        self.traction_form = traction_form
        n2, n1 = s.n2, s.n1
        self.vel = [np.zeros(s.dim * n2) for _ in range(3)]       # _velocities[0..2]
        self.ustar = np.zeros(s.dim * n2)                         # _intermediate_velocity
        self.p = np.zeros(n1)
        self.p_old = np.zeros(n1)
        self.body_force = None        # nodal P2 values, interleaved
        self.traction = None          # assembled boundary vector
        self.refactor = refactor_every_step
        self.newton_history = []
        self.newton_its = []

    def set_initial(self, u0, p0=None):
        self.vel[0][:] = u0
        self.vel[1][:] = u0
        if p0 is not None:
            self.p[:] = p0
            self.p_old[:] = p0

    def _setup_reference_rebuild(self):
        """The reference's LinearVariationalSolver re-assembles and re-factorises
        the constant Poisson and mass matrices on every call; the CPU baseline
        reproduces that cost."""
        s = self.s
        self.M = s.vector_mass()
        self.Ap = s.stiffness_p1()

    def step(self, alpha, k, vel_bc=(np.zeros(0, int), np.zeros(0)),
             p_bc=(np.zeros(0, int), np.zeros(0))):
        s, c = self.s, self.c
        a0, a1, a2 = alpha
        cc = c.get("convective_term") or 0.0
        cp = c["pressure_term"]
        cv = c["viscous_term"]
        cb = c.get("body_force_term")
        vd, vv = vel_bc
        pd, pv = p_bc
        # ---- diffusion step (Newton) : ns_ipcs_solver.py:106-147
        const = self.M @ (a1 * self.vel[1] + a2 * self.vel[2]) / k - cp * (self.D.T @ self.p_old)
        if self.body_force is not None:
            const -= cb * (self.M @ self.body_force)
        if self.traction is not None:
            const += self.traction
        L = (a0 / k) * self.M + cv * self.K

        def residual(x):
            b = L @ x + const
            if cc:
                b += cc * s.convection_residual(x, self.form)
            return b

        def jacobian(x):
            if cc:
                return L + cc * s.convection_jacobian(x, self.form)
            return L

        hist = []
        self.ustar, its = newton_solve(residual, jacobian, self.ustar, vd, vv,
                                       self.tol, 10.0 * self.tol, self.maxit, hist)
        self.newton_history.append(hist)
        self.newton_its.append(its)
        # ---- projection step : ns_ipcs_solver.py:149-171
        if self.refactor:
            self._setup_reference_rebuild()
        rhs = self.Ap @ self.p_old - (a0 / k) * (self.D @ self.ustar)
        self.p = linear_solve_dirichlet(self.Ap, rhs, pd, pv, pin_nullspace=(len(pd) == 0))
        # ---- velocity correction : ns_ipcs_solver.py:173-196
        rhs = self.M @ self.ustar - (k / a0) * (self.G @ (self.p - self.p_old))
        self.vel[0] = linear_solve_dirichlet(self.M, rhs, vd, vv)

    def advance(self):
        """ns_ipcs_solver.py:35-43."""
        self.vel[2] = self.vel[1].copy()
        self.vel[1] = self.vel[0].copy()
        self.p_old = self.p.copy()


# --------------------------------------------------------------------------
# monolithic BDF step (source/ns_bdf_solver.py)
# --------------------------------------------------------------------------
class BDFOracle:
    def __init__(self, space, coeffs, form="standard", traction_form=False,
                 tol=1e-10, maxit=50, pin_pressure=False):
        self.s = space
        self.c = coeffs
        self.form = form
        self.tol, self.maxit = tol, maxit
        s = space
        self.M = s.vector_mass()
        self.K = s.vector_stiffness(traction_form)
        self.D = s.divergence()
        self.nv = s.dim * s.n2
        self.n = self.nv + s.n1
        self.sol = [np.zeros(self.n) for _ in range(3)]      # _solutions[0..2]
        self.body_force = None
        self.traction = None
        self.pin_pressure = pin_pressure
        self.newton_history = []
        self.newton_its = []
        # rotating frame (2D): angular velocity / acceleration at the new time level
        # (source/ns_solver_base.py:173-211): + 2 c_cor omega (e_z x u, w) + c_e omega' (e_z x x, w)
        self.omega = 0.0
        self.omega_dot = 0.0

    def set_initial(self, u0, p0=None):
        for i in (0, 1):
            self.sol[i][: self.nv] = u0
            if p0 is not None:
                self.sol[i][self.nv:] = p0

    def step(self, alpha, k, bc=(np.zeros(0, int), np.zeros(0))):
        """bc dofs index the mixed vector."""
        s, c = self.s, self.c
        a0, a1, a2 = alpha
        nv = self.nv
        cc = c.get("convective_term") or 0.0
        cp = c["pressure_term"]
        cv = c["viscous_term"]
        cb = c.get("body_force_term")
        const = self.M @ (a1 * self.sol[1][:nv] + a2 * self.sol[2][:nv]) / k
        if self.body_force is not None:
            const -= cb * (self.M @ self.body_force)
        if self.traction is not None:
            const += self.traction
        L = (a0 / k) * self.M + cv * self.K
        if s.dim == 3:
            # 3D: omega / omega_dot are vectors, cross(Omega, u) and cross(dOmega/dt, x)
            # (source/ns_solver_base.py:186-190, 207-209)
            if np.any(self.omega_dot):
                rot = np.cross(np.asarray(self.omega_dot, dtype=float)[None, :], s.p2_nodes()).ravel()
                const += c["euler_term"] * (self.M @ rot)
            if np.any(self.omega):
                wx, wy, wz = (float(v) for v in self.omega)
                cross = np.array([[0.0, -wz, wy], [wz, 0.0, -wx], [-wy, wx, 0.0]])
                L = L + 2.0 * c["coriolis_term"] * sp.kron(s.mass_p2(), cross, format="csr")
        else:
            if self.omega_dot:
                X = s.p2_nodes()
                rot = np.stack([-X[:, 1], X[:, 0]], axis=1).ravel()          # e_z x x, exact in P2
                const += c["euler_term"] * self.omega_dot * (self.M @ rot)
            if self.omega:
                skew = sp.kron(s.mass_p2(), np.array([[0.0, -1.0], [1.0, 0.0]]), format="csr")
                L = L + 2.0 * c["coriolis_term"] * self.omega * skew
        Bt = -cp * self.D.T
        B = -cp * self.D
        bd, bv = bc
        bd = np.asarray(bd, dtype=np.int64)
        bv = np.asarray(bv, dtype=np.float64)
        if self.pin_pressure:
            bd = np.concatenate([bd, [nv]])
            bv = np.concatenate([bv, [0.0]])

        def residual(x):
            u, p = x[:nv], x[nv:]
            fu = L @ u + const + Bt @ p
            if cc:
                fu += cc * s.convection_residual(u, self.form)
            return np.concatenate([fu, B @ u])

        def jacobian(x):
            u = x[:nv]
            A = L + cc * s.convection_jacobian(u, self.form) if cc else L
            return sp.bmat([[A, Bt], [B, None]], format="csr")

        hist = []
        self.sol[0], its = newton_solve(residual, jacobian, self.sol[0], bd, bv,
                                        self.tol, 10.0 * self.tol, self.maxit, hist)
        self.newton_history.append(hist)
        self.newton_its.append(its)

    def advance(self):
        """ns_solver_base.py:1012-1016."""
        self.sol[2] = self.sol[1].copy()
        self.sol[1] = self.sol[0].copy()


def bdf_alpha(step_number, omega):
    """First-derivative BDF coefficients: source/bdf_time_stepping.py:27-31,123-127."""
    if step_number == 0:
        return (1.0, -1.0, 0.0)
    return ((1.0 + 2.0 * omega) / (1.0 + omega), -(1.0 + omega), omega * omega / (1.0 + omega))
