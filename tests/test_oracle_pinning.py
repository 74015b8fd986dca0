"""Pins the CPU oracle (oracle/fem_oracle.py) with independent known answers, because
the reference's own tests hold no numeric fixture for the assembly+solve path
("parity unpinned" against FEniCS output):

  K5  sympy-exact element matrices on the reference triangle;
  K7  structural identities (Jacobian = d residual, K 1 = 0, mass sums, div/grad adjointness);
  K1  Poiseuille channel of the reference's tests/test_ipcs_solver.py:37-43 /
      tests/test_stationary_solvers.py:183-188: exactly representable steady state;
  K3  Taylor-Green vortex (convergence_test/taylor_green_vortex.py:111-117) with the
      exact solution as Dirichlet data: second order in time for the BDF-2 schemes;
  K4  hydrostatic balance in a closed box.
"""
import numpy as np
import pytest
import sympy as sy

import fem_oracle as fo
from fem_mesh import FacetMarkers, TaylorHoodDofMap, rectangle_mesh


def make_space(nx, ny, p0=(0.0, 0.0), p1=(1.0, 1.0)):
    mesh = rectangle_mesh(p0, p1, nx, ny)
    dm = TaylorHoodDofMap(mesh)
    return mesh, dm, fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)


def boundary_nodes(mesh, dm, predicate):
    marks = FacetMarkers(mesh)
    marks.mark(predicate, 1)
    f = marks.facets_with_id(1)
    return np.unique(dm.facet_p2_nodes(f)), np.unique(dm.facet_p1_nodes(f))


# ---------------------------------------------------------------- K5: sympy
def _sympy_p2():
    x, y = sy.symbols("x y")
    l = [1 - x - y, x, y]
    N = [l[i] * (2 * l[i] - 1) for i in range(3)] + [4 * l[1] * l[2], 4 * l[0] * l[2], 4 * l[0] * l[1]]
    return x, y, l, N


def _integrate(expr, x, y):
    return sy.integrate(sy.integrate(expr, (y, 0, 1 - x)), (x, 0, 1))


def test_k5_reference_element_matrices_exact():
    x, y, l, N = _sympy_p2()
    # one reference cell as a mesh
    coords = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    cells = np.array([[0, 1, 2]])
    s = fo.Space(coords, cells, np.arange(6)[None, :], np.arange(3)[None, :])
    M = np.array([[float(_integrate(N[i] * N[j], x, y)) for j in range(6)] for i in range(6)])
    K = np.array([[float(_integrate(sy.diff(N[i], x) * sy.diff(N[j], x) + sy.diff(N[i], y) * sy.diff(N[j], y), x, y))
                   for j in range(6)] for i in range(6)])
    np.testing.assert_allclose(s.mass_p2().toarray(), M, rtol=0, atol=1e-15)
    np.testing.assert_allclose(s.stiffness_p2().toarray(), K, rtol=0, atol=2e-15)
    D = s.divergence().toarray()
    G = s.pressure_gradient().toarray()
    for i in range(3):
        for j in range(6):
            for a, var in enumerate((x, y)):
                assert abs(D[i, 2 * j + a] - float(_integrate(l[i] * sy.diff(N[j], var), x, y))) < 1e-15
                assert abs(G[2 * j + a, i] - float(_integrate(N[j] * sy.diff(l[i], var), x, y))) < 1e-15


def test_k5_convection_tensor_exact():
    x, y, l, N = _sympy_p2()
    coords = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    s = fo.Space(coords, np.array([[0, 1, 2]]), np.arange(6)[None, :], np.arange(3)[None, :])
    rng = np.random.default_rng(3)
    u = rng.integers(-3, 4, size=12).astype(float)
    ux = sum(sy.Rational(int(u[2 * k])) * N[k] for k in range(6))
    uy = sum(sy.Rational(int(u[2 * k + 1])) * N[k] for k in range(6))
    adv = (sy.diff(ux, x) * ux + sy.diff(ux, y) * uy, sy.diff(uy, x) * ux + sy.diff(uy, y) * uy)
    exact = np.array([float(_integrate(adv[a] * N[i], x, y)) for i in range(6) for a in range(2)])
    np.testing.assert_allclose(s.convection_residual(u), exact, rtol=1e-13, atol=1e-13)


# ---------------------------------------------------------------- K7: structure
@pytest.mark.parametrize("form", ["standard", "rotational", "divergence", "skew_symmetric"])
def test_k7_jacobian_is_derivative_of_residual(form):
    _, _, s = make_space(5, 4, p1=(1.3, 0.9))
    rng = np.random.default_rng(0)
    u, d = rng.standard_normal(2 * s.n2), rng.standard_normal(2 * s.n2)
    eps = 1e-6
    fd = (s.convection_residual(u + eps * d, form) - s.convection_residual(u - eps * d, form)) / (2 * eps)
    J = s.convection_jacobian(u, form)
    assert np.linalg.norm(J @ d - fd) <= 1e-8 * np.linalg.norm(fd)


def test_k7_structural_identities():
    mesh, dm, s = make_space(6, 5, p1=(2.0, 1.0))
    M, K = s.mass_p2(), s.stiffness_p2()
    assert abs(M.sum() - 2.0) < 1e-13                       # sum of mass = |Omega|
    assert np.abs(K @ np.ones(s.n2)).max() < 1e-12          # constants are in the kernel
    assert abs(s.mass_p1().sum() - 2.0) < 1e-13
    assert np.abs(s.stiffness_p1() @ np.ones(s.n1)).max() < 1e-12
    # (q, div u) = -(grad q, u) + boundary term: for u vanishing on the boundary D = -G^T
    D, G = s.divergence(), s.pressure_gradient()
    n2b, _ = boundary_nodes(mesh, dm, lambda X: np.ones(X.shape[0], bool))
    interior = np.ones(2 * s.n2, bool)
    interior[2 * n2b] = interior[2 * n2b + 1] = False
    assert np.abs((D + G.T).toarray()[:, interior]).max() < 1e-13
    # skew-symmetric form conserves energy: c(u; u, u) = 0
    u = np.random.default_rng(1).standard_normal(2 * s.n2)
    assert abs(u @ s.convection_residual(u, "skew_symmetric")) < 1e-10 * (u @ u)
    # div of a linear field, tested against 1: int div u = area * tr(A)
    X = s.p2_nodes()
    ulin = np.stack([0.3 * X[:, 0] + 0.1 * X[:, 1], -0.2 * X[:, 0] + 0.5 * X[:, 1]], axis=1).ravel()
    assert abs((D @ ulin).sum() - 2.0 * 0.8) < 1e-12


# ---------------------------------------------------------------- K1: Poiseuille
def _channel(n):
    mesh, dm, s = make_space(10 * n, n, p1=(10.0, 1.0))
    marks = FacetMarkers(mesh)
    marks.mark(lambda X: np.abs(X[:, 0]) < 1e-12, 1)
    marks.mark(lambda X: np.abs(X[:, 0] - 10.0) < 1e-12, 2)
    marks.mark(lambda X: np.abs(X[:, 1]) < 1e-12, 3)
    marks.mark(lambda X: np.abs(X[:, 1] - 1.0) < 1e-12, 4)
    vd, vv = [], []
    for mid in (1, 3, 4):     # inlet first, walls later (walls win at the corners)
        nodes = np.unique(dm.facet_p2_nodes(marks.facets_with_id(mid)))
        yy = dm.p2_coords[nodes, 1]
        ux = 6.0 * yy * (1.0 - yy) if mid == 1 else np.zeros_like(yy)
        vd += [2 * nodes, 2 * nodes + 1]
        vv += [ux, np.zeros_like(yy)]
    vd, vv = np.concatenate(vd), np.concatenate(vv)
    _, first = np.unique(vd[::-1], return_index=True)
    keep = len(vd) - 1 - first
    pd = np.unique(dm.facet_p1_nodes(marks.facets_with_id(2)))
    return mesh, dm, s, (vd[keep], vv[keep]), (pd, np.zeros(pd.size))


def test_k1_poiseuille_is_a_fixed_point_of_both_schemes():
    mesh, dm, s, vbc, pbc = _channel(3)
    Re = 10.0
    X2, X1 = s.p2_nodes(), s.p1_nodes()
    u_ex = np.stack([6.0 * X2[:, 1] * (1.0 - X2[:, 1]), np.zeros(s.n2)], axis=1).ravel()
    p_ex = 12.0 / Re * (10.0 - X1[:, 0])
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=1.0 / Re, body_force_term=None)
    # IPCS started from the exact steady state stays there (to round-off)
    ip = fo.IPCSOracle(s, coef, refactor_every_step=False)
    ip.set_initial(u_ex, p_ex)
    ip.vel[2][:] = u_ex
    ip.ustar[:] = u_ex
    for step in range(2):
        ip.step((1.5, -2.0, 0.5), 0.05, vbc, pbc)
        ip.advance()
    assert np.abs(ip.vel[0] - u_ex).max() < 1e-10
    assert np.abs(ip.p - p_ex).max() < 1e-9
    # monolithic BDF marches from rest to the same steady state
    bd = fo.BDFOracle(s, coef)
    mixed_bc = (np.concatenate([vbc[0], 2 * s.n2 + pbc[0]]), np.concatenate([vbc[1], pbc[1]]))
    for step in range(40):
        bd.step(fo.bdf_alpha(step, 1.0), 0.5, mixed_bc)
        bd.advance()
    assert np.abs(bd.sol[0][: 2 * s.n2] - u_ex).max() < 1e-8
    assert np.abs(bd.sol[0][2 * s.n2:] - p_ex).max() < 1e-7


# ---------------------------------------------------------------- K3: Taylor-Green
def _taylor_green(t, X, Re, g=2.0 * np.pi):
    x, y = X[:, 0], X[:, 1]
    eu, ep = np.exp(-2.0 * g * g * t / Re), np.exp(-4.0 * g * g * t / Re)
    u = np.stack([eu * np.cos(g * x) * np.sin(g * y), -eu * np.sin(g * x) * np.cos(g * y)], axis=1)
    p = -0.25 * ep * (np.cos(2 * g * x) + np.cos(2 * g * y))
    return u, p


@pytest.mark.parametrize("scheme", ["bdf", "ipcs"])
def test_k3_taylor_green_second_order_in_time(scheme):
    Re, T, n = 10.0, 0.08, 12
    mesh, dm, s = make_space(n, n)
    n2b, _ = boundary_nodes(mesh, dm, lambda X: np.ones(X.shape[0], bool))
    vd = np.concatenate([2 * n2b, 2 * n2b + 1])
    X2, X1 = s.p2_nodes(), s.p1_nodes()
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=1.0 / Re, body_force_term=None)
    errs = []
    for nsteps in (4, 8, 16):
        k = T / nsteps
        u0, p0 = _taylor_green(0.0, X2, Re)[0].ravel(), _taylor_green(0.0, X1, Re)[1]
        orc = (fo.BDFOracle(s, coef, pin_pressure=True) if scheme == "bdf"
               else fo.IPCSOracle(s, coef, refactor_every_step=False))
        orc.set_initial(u0, p0)
        if scheme == "ipcs":
            orc.ustar[:] = u0
        for step in range(nsteps):
            ub = _taylor_green((step + 1) * k, X2[n2b], Re)[0]
            vv = np.concatenate([ub[:, 0], ub[:, 1]])
            if scheme == "bdf":
                orc.step(fo.bdf_alpha(step, 1.0), k, (vd, vv))
            else:
                orc.step(fo.bdf_alpha(step, 1.0), k, (vd, vv))
            orc.advance()
        u = orc.sol[1][: 2 * s.n2] if scheme == "bdf" else orc.vel[1]
        errs.append(np.abs(u - _taylor_green(T, X2, Re)[0].ravel()).max())
    # error = C_t k^2 + C_h h^3: halving k must shrink the temporal part ~4x
    assert errs[0] > errs[1] > errs[2]
    rate = np.log2((errs[0] - errs[2]) / (errs[1] - errs[2]) - 1.0) if errs[1] > errs[2] else 2.0
    assert rate > 1.5, (errs, rate)
    assert errs[2] < 2e-2


# ---------------------------------------------------------------- K4: hydrostatics
def test_k4_hydrostatic_balance():
    mesh, dm, s = make_space(6, 6)
    n2b, _ = boundary_nodes(mesh, dm, lambda X: np.ones(X.shape[0], bool))
    vd = np.concatenate([2 * n2b, 2 * n2b + 1])
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.01, body_force_term=4.0)
    bd = fo.BDFOracle(s, coef, pin_pressure=True)
    bd.body_force = np.tile([0.0, -1.0], s.n2)
    bd.step((1.0, -1.0, 0.0), 0.1, (vd, np.zeros(vd.size)))
    u, p = bd.sol[0][: 2 * s.n2], bd.sol[0][2 * s.n2:]
    assert np.abs(u).max() < 1e-12
    # p = c_b f . x + const  (linear, exactly representable in P1)
    y = s.p1_nodes()[:, 1]
    assert np.abs((p - p[0]) - 4.0 * (-1.0) * (y - y[0])).max() < 1e-10


def test_cfl_number_closed_form():
    """constant velocity on a uniform right-diagonal mesh: the projected CFL field is the constant
    2 |u| k / h with h the hypotenuse (circumdiameter of a right triangle)."""
    from fem_mesh import TaylorHoodDofMap, rectangle_mesh
    mesh = rectangle_mesh((0.0, 0.0), (1.0, 1.0), 4, 4)
    dm = TaylorHoodDofMap(mesh)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    u = np.tile([0.6, 0.8], dm.n_p2)
    assert abs(fo.cfl_number(s, u, 0.1) - 2.0 * 0.1 / (np.sqrt(2.0) / 4)) < 1e-13


# ---------------------------------------------------------------- 3D (tetrahedra)
def make_space3(n, p1=(1.0, 1.0, 1.0)):
    from fem_mesh import box_mesh
    mesh = box_mesh((0.0, 0.0, 0.0), p1, n[0], n[1], n[2])
    dm = TaylorHoodDofMap(mesh)
    return mesh, dm, fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)


def test_k5_reference_tetrahedron_matrices_exact():
    """P2 / P1 Lagrange on the reference tetrahedron (UFC local order) against sympy-exact
    integrals: mass, stiffness, divergence and one convection vector."""
    x, y, z = sy.symbols("x y z")
    l = [1 - x - y - z, x, y, z]
    pairs = ((2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1))
    N = [l[i] * (2 * l[i] - 1) for i in range(4)] + [4 * l[a] * l[b] for a, b in pairs]

    def integ(e):
        return sy.integrate(sy.integrate(sy.integrate(e, (z, 0, 1 - x - y)), (y, 0, 1 - x)), (x, 0, 1))

    coords = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
    s = fo.Space(coords, np.array([[0, 1, 2, 3]]), np.arange(10)[None, :], np.arange(4)[None, :])
    M = np.array([[float(integ(N[i] * N[j])) for j in range(10)] for i in range(10)])
    np.testing.assert_allclose(s.mass_p2().toarray(), M, rtol=0, atol=1e-16)
    K = np.array([[float(integ(sum(sy.diff(N[i], v) * sy.diff(N[j], v) for v in (x, y, z))))
                   for j in range(10)] for i in range(10)])
    np.testing.assert_allclose(s.stiffness_p2().toarray(), K, rtol=0, atol=2e-15)
    D = s.divergence().toarray()
    for i in range(4):
        for j in (0, 3, 5, 9):
            for a, var in enumerate((x, y, z)):
                assert abs(D[i, 3 * j + a] - float(integ(l[i] * sy.diff(N[j], var)))) < 1e-15
    rng = np.random.default_rng(7)
    u = rng.integers(-2, 3, size=30).astype(float)
    comp = [sum(sy.Rational(int(u[3 * k + a])) * N[k] for k in range(10)) for a in range(3)]
    adv = [sum(sy.diff(comp[a], v) * comp[b] for b, v in enumerate((x, y, z))) for a in range(3)]
    exact = np.array([float(integ(adv[a] * N[i])) for i in (0, 4, 9) for a in range(3)])
    got = s.convection_residual(u).reshape(10, 3)[[0, 4, 9]].ravel()
    np.testing.assert_allclose(got, exact, rtol=1e-13, atol=1e-14)


@pytest.mark.parametrize("form", ["standard", "divergence", "skew_symmetric"])
def test_k7_3d_jacobian_and_structure(form):
    mesh, dm, s = make_space3((2, 3, 2), p1=(1.0, 1.5, 0.8))
    assert dm.n_p2 == 5 * 7 * 5 and dm.n_p1 == 3 * 4 * 3
    rng = np.random.default_rng(0)
    u, d = rng.standard_normal(3 * s.n2), rng.standard_normal(3 * s.n2)
    eps = 1e-6
    fd = (s.convection_residual(u + eps * d, form) - s.convection_residual(u - eps * d, form)) / (2 * eps)
    assert np.linalg.norm(s.convection_jacobian(u, form) @ d - fd) <= 1e-8 * np.linalg.norm(fd)
    vol = 1.0 * 1.5 * 0.8
    assert abs(s.mass_p2().sum() - vol) < 1e-13 and abs(s.mass_p1().sum() - vol) < 1e-13
    assert np.abs(s.stiffness_p2() @ np.ones(s.n2)).max() < 1e-12
    X = s.p2_nodes()
    assert np.abs(X - dm.p2_coords).max() < 1e-14
    A = np.array([[0.3, 0.1, -0.2], [0.0, 0.5, 0.4], [0.7, -0.3, 0.2]])
    ulin = (X @ A.T).ravel()
    assert abs((s.divergence() @ ulin).sum() - vol * np.trace(A)) < 1e-12
    if form == "skew_symmetric":
        assert abs(u @ s.convection_residual(u, form)) < 1e-10 * (u @ u)


def test_k1_3d_duct_stokes_polynomial_solution_is_reproduced():
    """Stokes flow with the quadratic velocity u = (y(1-y) + z(1-z), 0, 0), p = -4 c_v x (+c):
    -c_v lap u + grad p = (4 c_v, 0, 0) - (4 c_v, 0, 0) = 0.  u is in P2, p in P1, so one
    stationary BDF step (alpha = 0) with u prescribed on the whole boundary reproduces it to
    round-off on a Kuhn mesh."""
    mesh, dm, s = make_space3((2, 2, 2))
    cv = 0.37
    coef = dict(convective_term=0.0, pressure_term=1.0, viscous_term=cv, body_force_term=None)
    orc = fo.BDFOracle(s, coef, pin_pressure=True)
    X = dm.p2_coords
    exact = np.stack([X[:, 1] * (1 - X[:, 1]) + X[:, 2] * (1 - X[:, 2]), 0 * X[:, 0], 0 * X[:, 0]], axis=1)
    marks = FacetMarkers(mesh)
    marks.mark(lambda Y: np.ones(Y.shape[0], bool), 1)
    nodes = np.unique(dm.facet_p2_nodes(marks.facets_with_id(1)))
    dofs = (3 * nodes[:, None] + np.arange(3)[None, :]).ravel()
    orc.step((0.0, 0.0, 0.0), 1.0, (dofs, exact[nodes].ravel()))
    nv = 3 * dm.n_p2
    assert np.abs(orc.sol[0][:nv] - exact.ravel()).max() < 1e-12
    p = orc.sol[0][nv:]
    pex = -4.0 * cv * dm.p1_coords[:, 0]
    assert np.abs((p - p[0]) - (pex - pex[0])).max() < 1e-10


def test_cfl_number_closed_form_3d_and_keast_rule():
    """The degree-4 tetrahedron rule of the CFL projection integrates all monomials up to degree 4
    exactly (and not degree 5); constant velocity on a Kuhn mesh: CFL = 2 |u| k / h with h the
    space diagonal of the cubes (all six tetrahedra share the cube's circumsphere)."""
    import math
    pts, w = fo.keast_14()
    assert pts.shape == (14, 3) and abs(w.sum() - 1.0 / 6.0) < 1e-15
    worst5 = 0.0
    for i in range(6):
        for j in range(6 - i):
            for k in range(6 - i - j):
                exact = math.factorial(i) * math.factorial(j) * math.factorial(k) / math.factorial(i + j + k + 3)
                err = abs((w * pts[:, 0] ** i * pts[:, 1] ** j * pts[:, 2] ** k).sum() - exact) / exact
                if i + j + k <= 4:
                    assert err < 1e-13
                else:
                    worst5 = max(worst5, err)
    assert worst5 > 1e-3
    mesh, dm, s = make_space3((2, 3, 2), p1=(1.0, 0.9, 0.4))
    h = math.sqrt(0.5 ** 2 + 0.3 ** 2 + 0.2 ** 2)
    assert np.abs(fo.circumdiameter(s.geo.x) - h).max() < 1e-14
    u = np.tile([2.0, -1.0, 2.0], dm.n_p2)
    assert abs(fo.cfl_number(s, u, 0.05) - 2.0 * 3.0 * 0.05 / h) < 1e-13


def test_boundary_functionals_closed_forms():
    """oracle restatement of assemble(f * ds) (demo/dfg_benchmark.py:44-66): polynomial fields
    on the unit square / cube, checked against Gauss' theorem evaluated by hand --
    int -p n = -int grad p,  int nu (grad u + grad u^T) n = nu int (lap u + grad div u),
    int u.n = int div u; Poiseuille wall drag 6 nu L per unit height-1 channel wall."""
    from fem_mesh import TaylorHoodDofMap
    from grid_generator import hyper_cube, hyper_rectangle
    for dim, n in ((2, 3), (3, 2)):
        mesh, marks = hyper_cube(dim, n)
        dm = TaylorHoodDofMap(mesh)
        s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
        X, Y = dm.p2_coords, dm.p1_coords
        if dim == 2:
            u = np.stack([X[:, 0] ** 2 + X[:, 1], X[:, 0] * X[:, 1]], axis=1).ravel()
            expect_force = np.array([-2.0, 1.0]) + 0.3 * np.array([5.0, 0.0])
            expect_flux = 1.5
        else:
            u = np.stack([X[:, 0] ** 2 + X[:, 1], X[:, 0] * X[:, 2], X[:, 2] ** 2 - X[:, 1]], axis=1).ravel()
            # lap u = (2, 0, 2), div u = 2x + 2z -> grad div = (2, 0, 2)
            expect_force = np.array([-2.0, 1.0, 0.0]) + 0.3 * np.array([4.0, 0.0, 4.0])
            expect_flux = 2.0
        p = 1.0 + 2.0 * Y[:, 0] - Y[:, 1]
        bf = np.nonzero(mesh.facet_on_boundary)[0]
        force, flux, meas = fo.boundary_functionals(s, mesh.facets[bf], mesh.facet_cell[bf], u, p, 0.3, 1.0)
        assert np.abs(force - expect_force).max() < 1e-13
        assert abs(flux - expect_flux) < 1e-13 and abs(meas - 2.0 * dim) < 1e-13
    # Poiseuille channel (tests/test_ipcs_solver.py:37-43 inputs): wall shear on the bottom wall
    L, nu = 3.0, 0.1
    mesh, marks = hyper_rectangle((0.0, 0.0), (L, 1.0), (6, 4))
    dm = TaylorHoodDofMap(mesh)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    X, Y = dm.p2_coords, dm.p1_coords
    u = np.stack([6.0 * X[:, 1] * (1.0 - X[:, 1]), 0.0 * X[:, 1]], axis=1).ravel()
    p = 12.0 * nu * (L - Y[:, 0])
    wall = marks.facets_with_id(3)                      # bottom, outward normal (0, -1)
    force, flux, meas = fo.boundary_functionals(s, mesh.facets[wall], mesh.facet_cell[wall], u, p, nu, 1.0)
    # traction on the fluid: -p n + nu (du_x/dy) n_y e_x = (-6 nu, p) -> integrated: (-6 nu L, 6 nu L^2)
    assert abs(force[0] + 6.0 * nu * L) < 1e-13 and abs(force[1] - 6.0 * nu * L * L) < 1e-12
    assert abs(flux) < 1e-14 and abs(meas - L) < 1e-14
