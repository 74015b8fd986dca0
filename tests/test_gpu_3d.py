"""GPU parity on TETRAHEDRAL Taylor-Hood meshes (BoxMesh-style Kuhn meshes, BASELINE configs 3-4 in
small): the 3D element kernels (csrc/assembly3d.hip), the 3x3 / 1x3 / 3x1 block SpMV
instantiations and the IPCS / monolithic steps against the dimension-generic oracle, which is
pinned in 3D by sympy-exact tetrahedron matrices and a polynomial Stokes solution
(tests/test_oracle_pinning.py).  The reference never exercises its 3D branches."""
import numpy as np
import pytest

import _native as nat
import fem_oracle as fo
from fem_mesh import FacetMarkers, TaylorHoodDofMap, box_mesh
from gpu_common import rel

pytestmark = pytest.mark.gpu


def box3(n, p1=(1.0, 1.0, 1.0)):
    mesh = box_mesh((0.0, 0.0, 0.0), p1, *n)
    dm = TaylorHoodDofMap(mesh)
    marks = FacetMarkers(mesh)
    for axis in range(3):
        marks.mark(lambda X, a=axis: np.abs(X[:, a]) < 1e-12, 2 * axis + 1)
        marks.mark(lambda X, a=axis: np.abs(X[:, a] - p1[a]) < 1e-12, 2 * axis + 2)
    return mesh, dm, marks


def context3(mesh, dm):
    return nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1)


def lid_bc(dm, marks):
    """no-slip on five faces, lid (1, 0, 0) on z = top (wins on shared edges)"""
    last = {}
    for mid in (1, 2, 3, 4, 5, 6):
        nodes = np.unique(dm.facet_p2_nodes(marks.facets_with_id(mid)))
        for a in range(3):
            val = 1.0 if (mid == 6 and a == 0) else 0.0
            last.update(zip((3 * nodes + a).tolist(), [val] * nodes.size))
    d = np.array(sorted(last), dtype=np.int64)
    return d, np.array([last[i] for i in d.tolist()])


@pytest.fixture(scope="module")
def setup3():
    mesh, dm, marks = box3((3, 2, 2), p1=(1.0, 0.8, 0.6))
    ctx = context3(mesh, dm)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    yield mesh, dm, marks, ctx, s
    ctx.close()


def test_3d_constant_operators_and_spmv(setup3):
    _, dm, _, ctx, s = setup3
    pairs = [(nat.OP_MASS_P2, s.mass_p2()), (nat.OP_STIFF_P2, s.stiffness_p2()),
             (nat.OP_STIFF_P1, s.stiffness_p1()), (nat.OP_MASS_P1, s.mass_p1()),
             (nat.OP_DIV, s.divergence()), (nat.OP_GRAD, s.pressure_gradient()),
             (nat.OP_DIVT, s.divergence().T.tocsr())]
    rng = np.random.default_rng(3)
    for op, ref in pairs:
        A = ctx.operator_csr(op)
        assert A.shape == ref.shape
        assert abs(A - ref).max() <= 1e-13 * abs(ref).max(), op
        x = rng.standard_normal(ref.shape[1])
        assert rel(ctx.operator_apply(op, x), ref @ x) < 1e-13


@pytest.mark.parametrize("form_id,form", [(0, "standard"), (1, "rotational"), (2, "divergence"),
                                          (3, "skew_symmetric")])
def test_3d_momentum_residual_jacobian_and_newton_update(setup3, form_id, form):
    _, dm, marks, ctx, s = setup3
    rng = np.random.default_rng(5)
    Re, k, alpha = 20.0, 0.05, (1.5, -2.0, 0.5)
    u = [rng.standard_normal(dm.n_velocity) for _ in range(4)]
    p_old = rng.standard_normal(dm.n_p1)
    f = rng.standard_normal(dm.n_velocity)
    ctx.set_coeffs(1.0, 1.0, 1.0 / Re, 0.7)
    ctx.set_bdf(alpha, k)
    ctx.set_convective_form(form_id)
    bd, bv = lid_bc(dm, marks)
    ctx.set_dirichlet(nat.VELOCITY, bd, bv)
    for slot, v in ((nat.U1, u[1]), (nat.U2, u[2]), (nat.USTAR, u[3]), (nat.P_OLD, p_old),
                    (nat.BODY_FORCE, f)):
        ctx.set_state(slot, v)
    ctx.assemble(nat.SYS_MOMENTUM, new_step=True)
    M, K, D = s.vector_mass(), s.vector_stiffness(), s.divergence()
    L = alpha[0] / k * M + K / Re
    b = L @ u[3] + M @ (alpha[1] * u[1] + alpha[2] * u[2]) / k - D.T @ p_old - 0.7 * (M @ f) \
        + s.convection_residual(u[3], form)
    b[bd] = u[3][bd] - bv
    assert rel(ctx.get_rhs(nat.SYS_MOMENTUM), b) < 1e-13
    J = ctx.operator_csr(nat.OP_MOMENTUM_JAC)
    Jref = L + s.convection_jacobian(u[3], form)
    assert abs(J - Jref).max() <= 1e-13 * abs(Jref).max()
    # matrix-free Jacobian (what the fused 3D step drivers apply) = assembled Jacobian with
    # identity rows, Newton and Picard linearisations
    x = rng.standard_normal(dm.n_velocity)
    Jbc = fo.apply_dirichlet_rows(Jref, bd)
    assert rel(ctx.operator_apply(nat.OP_MOMENTUM_JAC_MF, x), Jbc @ x) < 1e-13
    ctx.set_convective_form(form_id, picard=True)
    Pref = L + s.picard_convection(u[3], form)
    Pbc = fo.apply_dirichlet_rows(Pref, bd)
    assert rel(ctx.operator_apply(nat.OP_MOMENTUM_JAC_MF, x), Pbc @ x) < 1e-13
    ctx.assemble(nat.SYS_MOMENTUM)                         # assembled Picard matrix of the seam
    P = ctx.operator_csr(nat.OP_MOMENTUM_JAC)
    assert abs(P - Pref).max() <= 1e-13 * abs(Pref).max()
    ctx.set_convective_form(form_id)
    ctx.assemble(nat.SYS_MOMENTUM)
    ctx.solve(nat.SYS_MOMENTUM, rtol=1e-13)
    dx = fo.spla.splu(fo.apply_dirichlet_rows(Jref, bd).tocsc()).solve(b)
    assert rel(ctx.get_state(nat.USTAR), u[3] - dx) < 1e-10
    ctx.set_coeffs(1.0, 1.0, 1.0 / Re)
    ctx.set_convective_form(0)


def test_3d_unknown_convective_form_is_refused(setup3):
    _, dm, _, ctx, _ = setup3
    with pytest.raises(nat.NativeError):
        ctx.set_convective_form(7)


def test_3d_ipcs_lid_driven_cavity_steps_match_oracle():
    """3D lid-driven cavity, Re = 50: Newton histories and fields of three IPCS steps against the
    LU oracle (Jacobi-preconditioned Krylov solves here)."""
    mesh, dm, marks = box3((3, 3, 3))
    ctx = context3(mesh, dm)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.02, body_force_term=None)
    orc = fo.IPCSOracle(s, coef, refactor_every_step=False)
    ctx.set_coeffs(1.0, 1.0, 0.02)
    vbc = lid_bc(dm, marks)
    pbc = (np.zeros(0, np.int64), np.zeros(0))
    ctx.set_dirichlet(nat.VELOCITY, *vbc)
    ctx.set_dirichlet(nat.PRESSURE, *pbc)
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = 1e-13
    for step in range(3):
        alpha = fo.bdf_alpha(step, 1.0)
        ctx.set_bdf(alpha, 0.05)
        info = ctx.step_ipcs(opts)
        orc.step(alpha, 0.05, vbc, pbc)
        assert info.newton_iterations == orc.newton_its[step]
        ctx.advance(0)
        orc.advance()
    assert rel(ctx.get_state(nat.U1), orc.vel[1]) < 1e-9
    pg, po = ctx.get_state(nat.P_OLD), orc.p_old
    assert rel(pg - pg.mean(), po - po.mean()) < 1e-8
    ctx.close()


def test_3d_multigrid_hierarchy_ipcs_and_monolithic_match_oracle():
    """Kuhn meshes are nested under grid refinement: geometric multigrid (P2 -> P1 -> 4^3 -> 2^3)
    preconditions the 3D IPCS and monolithic BDF steps; both agree with the LU oracle and the
    iteration counts are those of a working V-cycle."""
    from multigrid import attach_hierarchy
    mesh, dm, marks = box3((4, 4, 4))
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.02, body_force_term=None)
    vbc = lid_bc(dm, marks)
    pbc = (np.zeros(0, np.int64), np.zeros(0))
    for scheme in ("ipcs", "bdf"):
        ctx = context3(mesh, dm)
        assert attach_hierarchy(ctx, mesh, coarsest=1) == 2
        ctx.set_coeffs(1.0, 1.0, 0.02)
        ctx.set_dirichlet(nat.VELOCITY, *vbc)
        ctx.set_dirichlet(nat.PRESSURE, *pbc)
        ctx.set_dirichlet(nat.PRESSURE_PRECOND, *pbc)
        orc = fo.IPCSOracle(s, coef, refactor_every_step=False) if scheme == "ipcs" \
            else fo.BDFOracle(s, coef, pin_pressure=True)
        opts = ctx.default_step_opts()
        for o in (opts.momentum, opts.poisson, opts.correction):
            o.rtol = 1e-13
        opts.momentum.precond = opts.poisson.precond = 1
        for step in range(2):
            alpha = fo.bdf_alpha(step, 1.0)
            ctx.set_bdf(alpha, 0.05)
            if scheme == "ipcs":
                info = ctx.step_ipcs(opts)
                orc.step(alpha, 0.05, vbc, pbc)
                assert info.krylov_iterations_poisson <= 25
            else:
                info = ctx.step_bdf(opts)
                orc.step(alpha, 0.05, vbc)
            assert info.newton_iterations == orc.newton_its[step]
            assert info.krylov_iterations_momentum <= 40 * info.newton_iterations
            ctx.advance(0 if scheme == "ipcs" else 1)
            orc.advance()
        nv = dm.n_velocity
        uo, po = (orc.vel[1], orc.p_old) if scheme == "ipcs" else (orc.sol[1][:nv], orc.sol[1][nv:])
        assert rel(ctx.get_state(nat.U1), uo) < 1e-9
        pg = ctx.get_state(nat.P_OLD)
        assert rel(pg - pg.mean(), po - po.mean()) < 1e-8
        ctx.close()


def test_3d_cfl_number_matches_oracle():
    """nsfem_cfl_number on tetrahedra (k3_cfl: DG2 projection with the 14-point degree-4 rule,
    circumsphere diameter) vs the oracle on a distorted mesh with a random velocity; constant
    velocity on the Kuhn mesh gives 2 |u| k / h with h = the cubes' space diagonal."""
    from fem_mesh import Mesh
    mesh, dm, _ = box3((4, 3, 3), p1=(1.0, 0.9, 0.6))
    rng = np.random.default_rng(11)
    coords = mesh.coords.copy()
    inner = np.all((coords > 1e-12) & (coords < np.array([1.0, 0.9, 0.6]) - 1e-12), axis=1)
    coords[inner] += 0.02 * rng.standard_normal((int(inner.sum()), 3))
    mesh2 = Mesh(coords, mesh.cells)
    dm2 = TaylorHoodDofMap(mesh2)
    ctx = context3(mesh2, dm2)
    s = fo.Space(mesh2.coords, mesh2.cells, dm2.p2_dofmap, dm2.p1_dofmap)
    u = rng.standard_normal(dm2.n_velocity)
    ctx.set_state(nat.U0, u)
    ref = fo.cfl_number(s, u, 0.01)
    assert abs(ctx.cfl_number(nat.U0, 0.01) - ref) < 1e-12 * ref
    ctx.close()
    ctx = context3(mesh, dm)
    ctx.set_state(nat.U0, np.tile([1.0, 2.0, 2.0], dm.n_p2))
    h = np.sqrt((1.0 / 4) ** 2 + 0.3 ** 2 + 0.2 ** 2)
    assert abs(ctx.cfl_number(nat.U0, 0.01) - 2.0 * 3.0 * 0.01 / h) < 1e-13
    ctx.close()


def test_3d_spherical_shell_rotating_inner_sphere_matches_oracle():
    """grid_generator.spherical_shell(3, ...) (cubed-sphere shell cut into tetrahedra; the
    reference uses mshr) driven by a rotating inner sphere (u = e_z x x), outer sphere at rest:
    IPCS steps on the unstructured tetrahedral mesh against the LU oracle."""
    from grid_generator import SphericalAnnulusBoundaryMarkers as ids, spherical_shell
    mesh, marks = spherical_shell(3, (0.4, 1.0), 8)
    dm = TaylorHoodDofMap(mesh)
    ctx = context3(mesh, dm)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    X = dm.p2_coords
    last = {}
    for mid in (ids.exterior_boundary.value, ids.interior_boundary.value):
        nodes = np.unique(dm.facet_p2_nodes(marks.facets_with_id(mid)))
        spin = 1.0 if mid == ids.interior_boundary.value else 0.0
        vals = (-spin * X[nodes, 1], spin * X[nodes, 0], np.zeros(nodes.size))
        for a in range(3):
            last.update(zip((3 * nodes + a).tolist(), vals[a].tolist()))
    d = np.array(sorted(last), dtype=np.int64)
    vbc = (d, np.array([last[i] for i in d.tolist()]))
    pbc = (np.zeros(0, np.int64), np.zeros(0))
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.05, body_force_term=None)
    orc = fo.IPCSOracle(s, coef, refactor_every_step=False)
    ctx.set_coeffs(1.0, 1.0, 0.05)
    ctx.set_dirichlet(nat.VELOCITY, *vbc)
    ctx.set_dirichlet(nat.PRESSURE, *pbc)
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = 1e-13
    opts.correction.precond = 2
    for step in range(2):
        alpha = fo.bdf_alpha(step, 1.0)
        ctx.set_bdf(alpha, 0.05)
        info = ctx.step_ipcs(opts)
        orc.step(alpha, 0.05, vbc, pbc)
        assert info.newton_iterations == orc.newton_its[step]
        ctx.advance(0)
        orc.advance()
    u = ctx.get_state(nat.U1)
    assert rel(u, orc.vel[1]) < 1e-9
    assert np.abs(u.reshape(-1, 3)[:, :2]).max() > 0.3          # the fluid is dragged along
    pg, po = ctx.get_state(nat.P_OLD), orc.p_old
    assert rel(pg - pg.mean(), po - po.mean()) < 1e-8
    ctx.close()


def test_3d_traction_form_operator_and_open_boundary_steps_match_oracle():
    """Traction form of the viscous term on tetrahedra (k3_visc_extra: E[(i,a),(j,b)] =
    int d_b phi_i d_a phi_j, source/ns_solver_base.py:669-671) and boundary tractions
    (source/ns_solver_base.py:121-155) on an open face: operator parity, then IPCS steps with a
    body force and a traction on x = 1 (pressure prescribed there) against the LU oracle."""
    import fem_host
    mesh, dm, marks = box3((3, 3, 2), p1=(1.0, 0.9, 0.6))
    ctx = context3(mesh, dm)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    ctx.set_viscous_form(True)
    E = ctx.operator_csr(nat.OP_VISCOUS_EXTRA)
    K = fo.sp.kron(s.stiffness_p2(), fo.sp.identity(3))
    ref = s.vector_stiffness(traction_form=True) - K
    assert abs(E - ref).max() <= 1e-13 * abs(K).max()
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.05, body_force_term=1.0)
    orc = fo.IPCSOracle(s, coef, traction_form=True, refactor_every_step=False)
    X = dm.p2_coords
    f = np.stack([np.sin(np.pi * X[:, 1]), -1.0 + X[:, 0], 0.5 * X[:, 2]], axis=1).ravel()
    orc.body_force = f
    facets = marks.facets_with_id(2)                                 # x = 1
    traction = lambda Y: np.stack([0.3 * Y[:, 1], -0.1 + 0.0 * Y[:, 1], 0.2 * Y[:, 2]], axis=1)
    orc.traction = fem_host.traction_vector(dm, facets, traction)
    # the face load integrates the P2 interpolant exactly: total force = int t dA
    tot = orc.traction.reshape(-1, 3).sum(axis=0)
    assert np.abs(tot - [0.3 * 0.45 * 0.54, -0.1 * 0.54, 0.2 * 0.3 * 0.54]).max() < 1e-14
    ctx.set_coeffs(1.0, 1.0, 0.05, 1.0)
    ctx.set_state(nat.BODY_FORCE, f)
    ctx.set_state(nat.TRACTION, orc.traction)
    last = {}
    for mid in (1, 3, 4, 5, 6):
        nodes = np.unique(dm.facet_p2_nodes(marks.facets_with_id(mid)))
        for a in range(3):
            last.update(zip((3 * nodes + a).tolist(), [0.0] * nodes.size))
    d = np.array(sorted(last), dtype=np.int64)
    vbc = (d, np.zeros(d.size))
    pn = np.unique(dm.facet_p1_nodes(facets))
    pbc = (pn, 0.2 * np.ones(pn.size))
    ctx.set_dirichlet(nat.VELOCITY, *vbc)
    ctx.set_dirichlet(nat.PRESSURE, *pbc)
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = 1e-13
    for step in range(2):
        alpha = fo.bdf_alpha(step, 1.0)
        ctx.set_bdf(alpha, 0.05)
        info = ctx.step_ipcs(opts)
        orc.step(alpha, 0.05, vbc, pbc)
        assert info.newton_iterations == orc.newton_its[step]
        ctx.advance(0)
        orc.advance()
    assert rel(ctx.get_state(nat.U1), orc.vel[1]) < 1e-9
    assert rel(ctx.get_state(nat.P_OLD), orc.p_old) < 1e-9
    # assembled Jacobian of the seam with the traction block
    u = ctx.get_state(nat.U1)
    ctx.set_state(nat.USTAR, u)
    ctx.assemble(nat.SYS_MOMENTUM, new_step=True)
    J = ctx.operator_csr(nat.OP_MOMENTUM_JAC)
    Jref = 1.5 / 0.05 * s.vector_mass() + 0.05 * s.vector_stiffness(True) + s.convection_jacobian(u)
    assert abs(J - Jref).max() <= 1e-12 * abs(Jref).max()
    ctx.close()


def test_3d_rotating_frame_coriolis_and_euler_terms_match_oracle():
    """3D branches of source/ns_solver_base.py:173-211 (the reference marks them no cover):
    2 c_cor (Omega x u, w) in residual and Jacobian -- matrix-free and assembled -- and
    c_e (dOmega/dt x x, w) on the right-hand side, with vector-valued Omega: two monolithic BDF
    steps of the lid-driven cavity in a rotating frame against the LU oracle."""
    from fem_mesh import box_mesh
    from multigrid import attach_hierarchy
    mesh = box_mesh((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), 4, 4, 4)
    dm = TaylorHoodDofMap(mesh)
    marks = FacetMarkers(mesh)
    for axis in range(3):
        marks.mark(lambda X, a=axis: np.abs(X[:, a]) < 1e-12, 2 * axis + 1)
        marks.mark(lambda X, a=axis: np.abs(X[:, a] - 1.0) < 1e-12, 2 * axis + 2)
    ctx = context3(mesh, dm)
    attach_hierarchy(ctx, mesh, coarsest=2)
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.05, body_force_term=None,
                coriolis_term=1.5, euler_term=0.7)
    ctx.set_coeffs(1.0, 1.0, 0.05, None, 1.5, 0.7)
    bd, bv = lid_bc(dm, marks)
    ctx.set_dirichlet(nat.VELOCITY, bd, bv)
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    ctx.set_dirichlet(nat.PRESSURE_PRECOND, np.zeros(0, np.int32), np.zeros(0))
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    orc = fo.BDFOracle(s, coef, pin_pressure=True)
    with pytest.raises(nat.NativeError):
        ctx.set_angular_velocity(1.0, 0.0)                       # scalar form is the 2D one
    for mode in (2, 1):                                          # matrix-free, assembled Jacobian
        o = ctx.default_step_opts()
        o.momentum.rtol, o.momentum.precond, o.matrix_free = 1e-13, 1, mode
        for step, (omega, omega_dot) in enumerate([((0.3, -0.5, 0.8), (0.2, 0.1, -0.3)),
                                                   ((0.4, -0.4, 1.1), (-0.1, 0.3, -0.4))]):
            ctx.set_angular_velocity(omega, omega_dot)
            alpha = fo.bdf_alpha(step, 1.0)
            ctx.set_bdf(alpha, 0.05)
            info = ctx.step_bdf(o)
            ctx.advance(1)
            if mode == 2:
                orc.omega, orc.omega_dot = omega, omega_dot
                orc.step(alpha, 0.05, (bd, bv))
                orc.advance()
                assert info.newton_iterations == orc.newton_its[-1]
        nv = dm.n_velocity
        if mode == 2:
            u_mf, p_mf = ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD)
            assert rel(u_mf, orc.sol[1][:nv]) < 1e-9
            po = orc.sol[1][nv:]
            assert rel(p_mf - p_mf.mean(), po - po.mean()) < 1e-8
            for slot in (nat.U0, nat.U1, nat.U2):
                ctx.set_state(slot, np.zeros(nv))
            for slot in (nat.P, nat.P_OLD):
                ctx.set_state(slot, np.zeros(dm.n_p1))
        else:
            assert rel(ctx.get_state(nat.U1), u_mf) < 1e-10
    ctx.close()


def test_3d_boundary_force_and_flux_match_oracle(setup3):
    """nsfem_boundary_force on tetrahedra (edge-midpoint rule on the boundary faces) against the
    oracle's facet integral (3 x 3 collapsed Gauss on the physical face) for random nodal fields:
    per marked side and over the whole boundary; Gauss' theorem against the divergence rows."""
    mesh, dm, marks, ctx, s = setup3
    rng = np.random.default_rng(17)
    u, p = rng.standard_normal(dm.n_velocity), rng.standard_normal(dm.n_p1)
    ctx.set_state(nat.U0, u)
    ctx.set_state(nat.P, p)
    sets = [marks.facets_with_id(m) for m in (1, 2, 6)] + [np.nonzero(mesh.facet_on_boundary)[0]]
    for facets in sets:
        fc, fl = mesh.facet_cell_local(facets)
        for nu, sym in ((0.01, 1.0), (0.3, 0.0)):
            force, flux, meas = ctx.boundary_force(fc, fl, nu, sym)
            f_o, flux_o, meas_o = fo.boundary_functionals(s, mesh.facets[facets], mesh.facet_cell[facets],
                                                          u, p, nu, sym)
            assert np.abs(force - f_o).max() < 1e-12 * max(1.0, np.abs(f_o).max())
            assert abs(flux - flux_o) < 1e-12 * max(1.0, abs(flux_o))
            assert abs(meas - meas_o) < 1e-13 * meas_o
    assert abs(meas - 2.0 * (1.0 * 0.8 + 1.0 * 0.6 + 0.8 * 0.6)) < 1e-12
    assert abs(flux - ctx.operator_apply(nat.OP_DIV, u).sum()) < 1e-11


def test_3d_channel_re1000_bdf2_open_outlet_matches_oracle():
    """BASELINE configs[4] in small, at its Reynolds number: 2 : 1 : 1 channel, (8, 4, 4) cubes,
    inlet 16 y (1 - y) z (1 - z), no-slip side walls, natural outflow, Re = 1000, fully implicit
    BDF-2 on the mixed system (source/ns_bdf_solver.py:36-106) with the algebraic Schur Laplacian
    of the open outlet -- against the LU oracle: same Newton counts, fields to 1e-8; then the
    invariants bench.py --workload channel3d-bdf checks (mass balance from nsfem_boundary_force,
    inflow flux -4/9)."""
    from multigrid import attach_hierarchy, attach_schur_laplacian
    n = 4
    mesh, dm, marks = box3((2 * n, n, n), p1=(2.0, 1.0, 1.0))
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    inlet = np.unique(dm.facet_p2_nodes(marks.facets_with_id(1)))
    walls = np.unique(np.concatenate([dm.facet_p2_nodes(marks.facets_with_id(m)).ravel() for m in (3, 4, 5, 6)]))
    Xi = dm.p2_coords[inlet]
    prof = 16.0 * Xi[:, 1] * (1.0 - Xi[:, 1]) * Xi[:, 2] * (1.0 - Xi[:, 2])
    last = dict(zip((3 * inlet).tolist(), prof.tolist()))
    for a in (1, 2):
        last.update(zip((3 * inlet + a).tolist(), [0.0] * inlet.size))
    for a in range(3):
        last.update(zip((3 * walls + a).tolist(), [0.0] * walls.size))
    d = np.array(sorted(last), dtype=np.int64)
    vbc = (d, np.array([last[i] for i in d.tolist()]))
    nu = 1.0e-3
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=nu, body_force_term=None)
    ctx = context3(mesh, dm)
    attach_hierarchy(ctx, mesh, coarsest=1)
    ctx.set_coeffs(1.0, 1.0, nu)
    ctx.set_dirichlet(nat.VELOCITY, *vbc)
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    ctx.set_dirichlet(nat.PRESSURE_PRECOND, np.zeros(0, np.int32), np.zeros(0))
    assert attach_schur_laplacian(ctx, d) is False            # open outlet: non-singular
    orc = fo.BDFOracle(s, coef)
    opts = ctx.default_step_opts()
    opts.momentum.rtol, opts.momentum.precond, opts.momentum.max_iter = 1e-13, 1, 500
    k = 0.5 / n
    for step in range(3):
        alpha = fo.bdf_alpha(step, 1.0)
        ctx.set_bdf(alpha, k)
        info = ctx.step_bdf(opts)
        orc.step(alpha, k, vbc)
        assert info.converged and info.newton_iterations == orc.newton_its[step]
        assert info.krylov_iterations_momentum <= 60 * info.newton_iterations
        ctx.advance(1)
        orc.advance()
    nv = dm.n_velocity
    assert rel(ctx.get_state(nat.U1), orc.sol[1][:nv]) < 1e-8
    assert rel(ctx.get_state(nat.P_OLD), orc.sol[1][nv:]) < 1e-8
    flux = {}
    for m in range(1, 7):
        fc, fl = mesh.facet_cell_local(marks.facets_with_id(m))
        flux[m] = ctx.boundary_force(fc, fl, nu, 0.0, nat.U0, nat.P)[1]
    # inflow = integral of the P2 interpolant of the (biquadratic) inlet profile: area / 3 x the
    # edge-midpoint values of every inlet face; -4/9 up to O(h^4)
    fin = marks.facets_with_id(1)
    Xm = dm.p2_coords[dm.facet_p2_nodes(fin)[:, 3:]]
    pm = 16.0 * Xm[..., 1] * (1.0 - Xm[..., 1]) * Xm[..., 2] * (1.0 - Xm[..., 2])
    tri = mesh.coords[mesh.facets[fin].astype(np.int64)]
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    assert abs(flux[1] + (area / 3.0 * pm.sum(axis=1)).sum()) < 1e-13
    assert abs(flux[1] + 4.0 / 9.0) < 1e-3
    assert abs(sum(flux.values())) < 1e-9 and flux[2] > 0.4   # what enters leaves through x = 2
    assert max(abs(flux[m]) for m in (3, 4, 5, 6)) < 1e-14
    ctx.close()


def test_3d_hierarchy_continues_below_an_odd_level_with_a_non_nested_kuhn_mesh():
    """10^3 cubes: 10 -> 5 (nested) -> 3 (NON-NESTED: multigrid.interpolation_prolongation_3d, the point's Kuhn tetrahedron
    from the ordering of its cube-local coordinates).  The multigrid-preconditioned IPCS steps reach the same discrete
    solution as the Jacobi-preconditioned ones (rtol 1e-12 both), with the iteration counts of a working V-cycle."""
    from multigrid import attach_hierarchy, structured_hierarchy
    mesh, dm, marks = box3((10, 10, 10))
    assert [m.structured[2] for m, _ in structured_hierarchy(*mesh.structured, coarsest=2)] == [5, 3]
    vbc = lid_bc(dm, marks)
    pbc = (np.zeros(0, np.int64), np.zeros(0))
    out = {}
    for tag in ("jacobi", "multigrid"):
        ctx = context3(mesh, dm)
        if tag == "multigrid":
            assert attach_hierarchy(ctx, mesh, coarsest=2) == 2
        ctx.set_coeffs(1.0, 1.0, 0.02)
        ctx.set_dirichlet(nat.VELOCITY, *vbc)
        ctx.set_dirichlet(nat.PRESSURE, *pbc)
        opts = ctx.default_step_opts()
        for o in (opts.momentum, opts.poisson, opts.correction):
            o.rtol = 1e-12
            o.max_iter = 4000
        if tag == "multigrid":
            opts.momentum.precond = opts.poisson.precond = 1
        its = []
        for step in range(2):
            ctx.set_bdf(fo.bdf_alpha(step, 1.0), 0.05)
            info = ctx.step_ipcs(opts)
            ctx.advance(0)
            its.append((info.newton_iterations, info.krylov_iterations_momentum, info.krylov_iterations_poisson))
        out[tag] = (ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD), its)
        ctx.close()
    (u0, p0, its0), (u1, p1, its1) = out["jacobi"], out["multigrid"]
    assert rel(u1, u0) < 1e-9 and rel(p1 - p1.mean(), p0 - p0.mean()) < 1e-8
    assert all(b[2] <= 25 and b[2] < a[2] for a, b in zip(its0, its1)), (its0, its1)
