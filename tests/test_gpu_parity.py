"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs, plus size-independent properties at BASELINE.json's full size.

Tolerances (north_star): 1e-10 relative for linear (Stokes) steps, 1e-6 relative after
the nonlinear loop; operators and residuals are compared to round-off (1e-13 relative).
The Krylov solves use rtol 1e-12/1e-13, the oracle sparse LU.
"""
import numpy as np
import pytest

import _native as nat
import fem_oracle as fo
from gpu_common import box, cavity_bc, context, rel, velocity_bc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup16():
    mesh, dm, marks = box(16, 12, p1=(1.5, 1.0))
    ctx = context(mesh, dm)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    yield mesh, dm, marks, ctx, s
    ctx.close()


def test_constant_operators_match_oracle(setup16):
    _, _, _, ctx, s = setup16
    pairs = [(nat.OP_MASS_P2, s.mass_p2()), (nat.OP_STIFF_P2, s.stiffness_p2()),
             (nat.OP_STIFF_P1, s.stiffness_p1()), (nat.OP_MASS_P1, s.mass_p1()),
             (nat.OP_DIV, s.divergence()), (nat.OP_GRAD, s.pressure_gradient()),
             (nat.OP_DIVT, s.divergence().T.tocsr())]
    for op, ref in pairs:
        A = ctx.operator_csr(op)
        assert A.shape == ref.shape
        assert abs(A - ref).max() <= 1e-13 * abs(ref).max(), op
    ctx.set_viscous_form(True)
    E = ctx.operator_csr(nat.OP_VISCOUS_EXTRA)
    K = fo.sp.kron(s.stiffness_p2(), fo.sp.identity(2))
    ref = s.vector_stiffness(traction_form=True) - K
    assert abs(E - ref).max() <= 1e-13 * abs(K).max()
    ctx.set_viscous_form(False)


def test_spmv_kernels_match_scipy(setup16):
    _, dm, _, ctx, s = setup16
    rng = np.random.default_rng(11)
    for op, ref in [(nat.OP_DIV, s.divergence()), (nat.OP_GRAD, s.pressure_gradient()),
                    (nat.OP_STIFF_P1, s.stiffness_p1()), (nat.OP_MASS_P2, s.mass_p2())]:
        x = rng.standard_normal(ref.shape[1])
        y = ctx.operator_apply(op, x)
        assert rel(y, ref @ x) < 1e-13
    # linearity of the production kernel
    x1, x2 = rng.standard_normal(dm.n_p1), rng.standard_normal(dm.n_p1)
    y = ctx.operator_apply(nat.OP_GRAD, 2.0 * x1 - 3.0 * x2)
    assert rel(y, 2.0 * ctx.operator_apply(nat.OP_GRAD, x1) - 3.0 * ctx.operator_apply(nat.OP_GRAD, x2)) < 1e-13


def test_momentum_residual_and_jacobian_match_oracle(setup16):
    mesh, dm, marks, ctx, s = setup16
    rng = np.random.default_rng(5)
    Re, k, alpha = 50.0, 0.02, (1.5, -2.0, 0.5)
    u = [rng.standard_normal(dm.n_velocity) for _ in range(4)]
    p_old = rng.standard_normal(dm.n_p1)
    f = rng.standard_normal(dm.n_velocity)
    ctx.set_coeffs(1.0, 1.0, 1.0 / Re, 0.7)
    ctx.set_bdf(alpha, k)
    bd, bv = cavity_bc(dm, marks)
    ctx.set_dirichlet(nat.VELOCITY, bd, bv)
    for slot, v in ((nat.U1, u[1]), (nat.U2, u[2]), (nat.USTAR, u[3]), (nat.P_OLD, p_old),
                    (nat.BODY_FORCE, f)):
        ctx.set_state(slot, v)
    ctx.assemble(nat.SYS_MOMENTUM, new_step=True)
    M, K, D = s.vector_mass(), s.vector_stiffness(), s.divergence()
    L = alpha[0] / k * M + K / Re
    b = L @ u[3] + M @ (alpha[1] * u[1] + alpha[2] * u[2]) / k - D.T @ p_old - 0.7 * (M @ f) \
        + s.convection_residual(u[3])
    b[bd] = u[3][bd] - bv
    assert rel(ctx.get_rhs(nat.SYS_MOMENTUM), b) < 1e-13
    assert abs(ctx.residual_norm(nat.SYS_MOMENTUM) - np.linalg.norm(b)) < 1e-12 * np.linalg.norm(b)
    J = ctx.operator_csr(nat.OP_MOMENTUM_JAC)
    Jref = L + s.convection_jacobian(u[3])
    assert abs(J - Jref).max() <= 1e-13 * abs(Jref).max()
    # one Newton update through the device BiCGStab vs sparse LU
    ctx.solve(nat.SYS_MOMENTUM, rtol=1e-13)
    dx = fo.spla.splu(fo.apply_dirichlet_rows(Jref, bd).tocsc()).solve(b)
    assert rel(ctx.get_state(nat.USTAR), u[3] - dx) < 1e-10
    ctx.set_coeffs(1.0, 1.0, 1.0 / Re)


def _run_ipcs(ctx, orc, nsteps, k, vbc, pbc, rtol=1e-13, body_force=None):
    ctx.set_dirichlet(nat.VELOCITY, *vbc)
    ctx.set_dirichlet(nat.PRESSURE, *pbc)
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = rtol
    out = []
    for step in range(nsteps):
        alpha = fo.bdf_alpha(step, 1.0)
        ctx.set_bdf(alpha, k)
        info = ctx.step_ipcs(opts)
        orc.step(alpha, k, vbc, pbc)
        out.append((info, ctx.get_state(nat.USTAR), ctx.get_state(nat.U0), ctx.get_state(nat.P)))
        ctx.advance(0)
        orc.advance()
    return out


def test_ipcs_cavity_steps_match_oracle():
    """Config-1-sized lid-driven cavity (n = 16 here so that the LU oracle is instant),
    Re = 100: Newton residual histories, u*, u and p (modulo a constant: the reference's
    Poisson problem is singular)."""
    mesh, dm, marks = box(16, 16)
    ctx = context(mesh, dm)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.01, body_force_term=None)
    orc = fo.IPCSOracle(s, coef, refactor_every_step=False)
    ctx.set_coeffs(1.0, 1.0, 0.01)
    vbc = cavity_bc(dm, marks)
    pbc = (np.zeros(0, np.int64), np.zeros(0))
    res = _run_ipcs(ctx, orc, 4, 0.01, vbc, pbc)
    for step, (info, us, u, p) in enumerate(res):
        hist = orc.newton_history[step]
        assert info.newton_iterations == orc.newton_its[step]
        for i, r in enumerate(hist[:-1]):          # last entry is at round-off level
            assert abs(info.newton_residuals[i] - r) <= 1e-6 * r + 1e-12 * hist[0]
    # final state after 4 nonlinear steps: north_star tolerance 1e-6, achieved ~1e-11
    assert rel(ctx.get_state(nat.U1), orc.vel[1]) < 1e-9
    assert rel(ctx.get_state(nat.USTAR), orc.ustar) < 1e-9
    pg, po = ctx.get_state(nat.P_OLD), orc.p_old
    assert rel(pg - pg.mean(), po - po.mean()) < 1e-8
    ctx.close()


def test_ipcs_cavity_multigrid_preconditioning_same_answer():
    """Multigrid-preconditioned Krylov (the production setting) vs the LU oracle."""
    from multigrid import attach_hierarchy
    mesh, dm, marks = box(32, 32)
    ctx = context(mesh, dm)
    assert attach_hierarchy(ctx, mesh, coarsest=4) == 3
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.01, body_force_term=None)
    orc = fo.IPCSOracle(s, coef, refactor_every_step=False)
    ctx.set_coeffs(1.0, 1.0, 0.01)
    vbc = cavity_bc(dm, marks)
    ctx.set_dirichlet(nat.VELOCITY, *vbc)
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = 1e-13
    opts.momentum.precond = opts.poisson.precond = 1
    for step in range(3):
        alpha = fo.bdf_alpha(step, 1.0)
        ctx.set_bdf(alpha, 0.01)
        info = ctx.step_ipcs(opts)
        orc.step(alpha, 0.01, vbc)
        assert info.newton_iterations == orc.newton_its[step]
        assert info.krylov_iterations_poisson <= 15          # mesh-independent convergence
        assert info.krylov_iterations_momentum <= 10 * info.newton_iterations
        ctx.advance(0)
        orc.advance()
    assert rel(ctx.get_state(nat.U1), orc.vel[1]) < 1e-9
    pg, po = ctx.get_state(nat.P_OLD), orc.p_old
    assert rel(pg - pg.mean(), po - po.mean()) < 1e-8
    ctx.close()


def test_multigrid_with_pressure_dirichlet_and_time_dependent_inlet():
    """Channel with pressure outlet (masked coarse levels) on a mesh that coarsens once in
    y; inlet values change every step (values refresh without rebuilding the hierarchy)."""
    from multigrid import attach_hierarchy
    mesh, dm, marks = box(80, 8, p1=(10.0, 1.0))
    ctx = context(mesh, dm)
    assert attach_hierarchy(ctx, mesh, coarsest=4) == 1
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.1, body_force_term=None)
    orc = fo.IPCSOracle(s, coef, refactor_every_step=False)
    ctx.set_coeffs(1.0, 1.0, 0.1)
    zero = lambda X: np.zeros((X.shape[0], 2))
    pn = np.unique(dm.facet_p1_nodes(marks.facets_with_id(2)))
    pbc = (pn, np.zeros(pn.size))
    ctx.set_dirichlet(nat.PRESSURE, *pbc)
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = 1e-13
    opts.momentum.precond = opts.poisson.precond = 1
    for step in range(3):
        t = 0.01 * (step + 1)
        inlet = lambda X: np.stack([6.0 * X[:, 1] * (1.0 - X[:, 1]) * (1.0 + 0.5 * np.sin(np.pi * t)),
                                    0.0 * X[:, 1]], axis=1)
        vbc = velocity_bc(dm, marks, [(1, inlet), (3, zero), (4, zero)])
        ctx.set_dirichlet(nat.VELOCITY, *vbc)
        alpha = fo.bdf_alpha(step, 1.0)
        ctx.set_bdf(alpha, 0.01)
        info = ctx.step_ipcs(opts)
        orc.step(alpha, 0.01, vbc, pbc)
        assert info.krylov_iterations_poisson <= 40
        ctx.advance(0)
        orc.advance()
    assert rel(ctx.get_state(nat.U1), orc.vel[1]) < 1e-9
    assert rel(ctx.get_state(nat.P_OLD), orc.p_old) < 1e-9
    ctx.close()


def test_ipcs_stokes_channel_linear_steps_1e10():
    """Linear (convective term None) channel steps with a pressure Dirichlet outlet --
    the reference's tests/test_ipcs_solver.py case; north_star tolerance 1e-10."""
    mesh, dm, marks = box(30, 3, p1=(10.0, 1.0))
    ctx = context(mesh, dm)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    coef = dict(convective_term=None, pressure_term=1.0, viscous_term=0.1, body_force_term=None)
    orc = fo.IPCSOracle(s, coef, refactor_every_step=False)
    ctx.set_coeffs(None, 1.0, 0.1)
    zero = lambda X: np.zeros((X.shape[0], 2))
    inlet = lambda X: np.stack([6.0 * X[:, 1] * (1.0 - X[:, 1]), 0.0 * X[:, 1]], axis=1)
    vbc = velocity_bc(dm, marks, [(1, inlet), (3, zero), (4, zero)])
    pn = np.unique(dm.facet_p1_nodes(marks.facets_with_id(2)))
    pbc = (pn, np.zeros(pn.size))
    res = _run_ipcs(ctx, orc, 3, 0.002, vbc, pbc, rtol=1e-14)
    assert all(info.newton_iterations == 1 for info, *_ in res)
    assert rel(ctx.get_state(nat.U1), orc.vel[1]) < 1e-10
    assert rel(ctx.get_state(nat.P_OLD), orc.p_old) < 1e-10
    ctx.close()


def test_ipcs_body_force_and_traction_form():
    mesh, dm, marks = box(10, 10)
    ctx = context(mesh, dm)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.02, body_force_term=1.0)
    orc = fo.IPCSOracle(s, coef, traction_form=True, refactor_every_step=False)
    X = dm.p2_coords
    f = np.stack([np.sin(np.pi * X[:, 1]), -1.0 + X[:, 0]], axis=1).ravel()
    orc.body_force = f
    facets = marks.facets_with_id(2)
    nodes = dm.facet_p2_nodes(facets)
    tvals = np.stack([0.3 * X[nodes, 1], -0.1 + 0.0 * X[nodes, 1]], axis=2)
    orc.traction = s.traction_vector(nodes, tvals)
    ctx.set_coeffs(1.0, 1.0, 0.02, 1.0)
    ctx.set_viscous_form(True)
    ctx.set_state(nat.BODY_FORCE, f)
    ctx.set_state(nat.TRACTION, orc.traction)
    zero = lambda X: np.zeros((X.shape[0], 2))
    vbc = velocity_bc(dm, marks, [(1, zero), (3, zero), (4, zero)])
    # open boundary 2 carries the traction; the pressure is pinned there (otherwise the
    # Poisson problem of the scheme is singular AND incompatible: ill-posed)
    pn = np.unique(dm.facet_p1_nodes(facets))
    pbc = (pn, 0.2 * np.ones(pn.size))
    _run_ipcs(ctx, orc, 2, 0.05, vbc, pbc)
    assert rel(ctx.get_state(nat.U1), orc.vel[1]) < 1e-9
    assert rel(ctx.get_state(nat.P_OLD), orc.p_old) < 1e-9
    ctx.close()


def test_mass_projection_and_mean_pressure():
    mesh, dm, marks = box(8, 8)
    ctx = context(mesh, dm)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    rng = np.random.default_rng(2)
    b = rng.standard_normal(dm.n_velocity)
    x = ctx.mass_solve(nat.VELOCITY, b)
    assert rel(s.vector_mass() @ x, b) < 1e-11
    p = rng.standard_normal(dm.n_p1)
    ctx.set_state(nat.P, p)
    mean = ctx.shift_mean_pressure(0.25)
    ref_mean = (s.mass_p1() @ p).sum() / 1.0
    assert abs(mean - ref_mean) < 1e-13
    assert np.abs(ctx.get_state(nat.P) - (p - (ref_mean - 0.25))).max() < 1e-13
    ctx.close()


def test_cfl_number_matches_oracle():
    """nsfem_cfl_number vs the oracle's local-projection restatement of
    source/ns_problem.py:554-587 on a distorted mesh with a random velocity; a constant field on
    the uniform mesh gives the closed form 2 |u| k / h."""
    mesh, dm, marks = box(12, 8, p1=(2.0, 1.0))
    rng = np.random.default_rng(5)
    coords = mesh.coords.copy()
    interior = ~np.isin(np.arange(coords.shape[0]), np.unique(mesh.edges[mesh.edge_on_boundary]))
    coords[interior] += 0.02 * rng.standard_normal((int(interior.sum()), 2))
    from fem_mesh import Mesh, TaylorHoodDofMap
    mesh2 = Mesh(coords, mesh.cells)
    dm2 = TaylorHoodDofMap(mesh2)
    ctx = context(mesh2, dm2)
    s = fo.Space(mesh2.coords, mesh2.cells, dm2.p2_dofmap, dm2.p1_dofmap)
    u = rng.standard_normal(dm2.n_velocity)
    ctx.set_state(nat.U0, u)
    assert abs(ctx.cfl_number(nat.U0, 0.01) - fo.cfl_number(s, u, 0.01)) < 1e-12 * fo.cfl_number(s, u, 0.01)
    ctx.close()
    ctx = context(mesh, dm)
    u = np.tile([3.0, 4.0], dm.n_p2)
    ctx.set_state(nat.U0, u)
    hx, hy = 2.0 / 12, 1.0 / 8
    h = np.hypot(hx, hy)                                     # circumdiameter of a right triangle
    assert abs(ctx.cfl_number(nat.U0, 0.01) - 2.0 * 5.0 * 0.01 / h) < 1e-13
    ctx.close()


def test_error_paths_are_loud():
    mesh, dm, marks = box(4, 4)
    ctx = context(mesh, dm)
    with pytest.raises(nat.NativeError):
        ctx.set_state(nat.P, np.zeros(3))                       # wrong size
    with pytest.raises(nat.NativeError):
        ctx.set_dirichlet(nat.VELOCITY, np.array([10 ** 6]), np.array([0.0]))
    with pytest.raises(nat.NativeError):
        ctx.solve(nat.SYS_POISSON)                              # nothing assembled
    ctx.set_coeffs(1.0, 1.0, 0.01)
    ctx.set_dirichlet(nat.VELOCITY, *cavity_bc(dm, marks))
    ctx.set_bdf((1.0, -1.0, 0.0), 0.01)
    opts = ctx.default_step_opts()
    opts.momentum.max_iter = 1                                  # Krylov cannot converge
    with pytest.raises(nat.NativeError) as err:
        ctx.step_ipcs(opts)
    assert err.value.code == nat.ERR_NOT_CONVERGED
    with pytest.raises(nat.NativeError):                         # monolithic step without a hierarchy
        ctx.step_bdf(ctx.default_step_opts())
    import scipy.sparse as sp
    with pytest.raises(nat.NativeError):                         # Schur operator before mg_finalize
        ctx.mg_set_schur_operator(0, sp.identity(dm.n_p1, format="csr"), False)
    from multigrid import attach_hierarchy
    attach_hierarchy(ctx, mesh, coarsest=2)
    with pytest.raises(nat.NativeError):                         # wrong size / no such level
        ctx.mg_set_schur_operator(0, sp.identity(dm.n_p1 + 1, format="csr"), False)
    with pytest.raises(nat.NativeError):
        ctx.mg_set_schur_operator(7, sp.identity(dm.n_p1, format="csr"), False)
    with pytest.raises(nat.NativeError):                         # hierarchy is final
        ctx.mg_add_level(mesh.coords, mesh.cells, np.zeros(dm.n_p1 + 1, np.int32), np.zeros(0, np.int32), np.zeros(0))
    with pytest.raises(nat.NativeError):                         # rotating frame without a coefficient
        ctx.set_angular_velocity(1.0)
        ctx.step_ipcs(ctx.default_step_opts())
    ctx.close()
    with pytest.raises(nat.NativeError):                         # bad dof map entry
        nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap + 10 ** 6, dm.p1_dofmap, dm.n_p2, dm.n_p1)
    with pytest.raises(nat.NativeError):                         # degenerate cell
        bad = mesh.coords.copy()
        bad[mesh.cells[0]] = bad[mesh.cells[0, 0]]
        nat.NsfemContext(bad, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1)


def test_full_size_properties_n512():
    """BASELINE config 2 size (n = 512, 2,364,419 dofs): properties that need no oracle."""
    mesh, dm, marks = box(512, 512)
    assert dm.n_dofs == 2364419
    ctx = context(mesh, dm)
    rng = np.random.default_rng(0)
    ones2, ones1 = np.ones(dm.n_p2), np.ones(dm.n_p1)
    # K 1 = 0, sum(M) = |Omega|, A_p 1 = 0
    assert np.abs(ctx.operator_apply(nat.OP_STIFF_P2, ones2)).max() < 1e-9
    assert abs(ctx.operator_apply(nat.OP_MASS_P2, ones2).sum() - 1.0) < 1e-12
    assert np.abs(ctx.operator_apply(nat.OP_STIFF_P1, ones1)).max() < 1e-9
    # adjointness  y . (D x) = x . (D^T y)  between the independently integrated D and D^T
    x, y = rng.standard_normal(dm.n_velocity), rng.standard_normal(dm.n_p1)
    lhs = y @ ctx.operator_apply(nat.OP_DIV, x)
    rhs = x @ ctx.operator_apply(nat.OP_DIVT, y)
    assert abs(lhs - rhs) < 1e-10 * max(abs(lhs), 1.0)
    # divergence of a linear field integrates exactly: sum_i (D u)_i = |Omega| tr(A)
    X = dm.p2_coords
    ulin = np.stack([0.3 * X[:, 0] + 0.1 * X[:, 1], -0.2 * X[:, 0] + 0.5 * X[:, 1]], axis=1).ravel()
    assert abs(ctx.operator_apply(nat.OP_DIV, ulin).sum() - 0.8) < 1e-11
    # two IPCS steps of the cavity: Newton converges, boundary values hold, the corrected
    # velocity is discretely closer to divergence free than u*, mass CG needs O(10) its
    ctx.set_coeffs(1.0, 1.0, 0.01)
    bd, bv = cavity_bc(dm, marks)
    ctx.set_dirichlet(nat.VELOCITY, bd, bv)
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = 1e-10
    for step in range(2):
        ctx.set_bdf(fo.bdf_alpha(step, 1.0), 1e-3)
        info = ctx.step_ipcs(opts)
        assert 1 <= info.newton_iterations <= 6
        assert info.newton_residuals[info.newton_iterations] < max(
            1e-10, 1e-9 * info.newton_residuals[0])
        assert info.krylov_iterations_correction < 60
        u, us = ctx.get_state(nat.U0), ctx.get_state(nat.USTAR)
        assert np.abs(u[bd] - bv).max() == 0.0
        assert np.isfinite(u).all() and np.isfinite(ctx.get_state(nat.P)).all()
        ctx.advance(0)
    ctx.close()


# --------------------------------------------------------------- monolithic BDF-2
def _run_bdf(ctx, orc, nsteps, k, vbc, rtol=1e-12):
    opts = ctx.default_step_opts()
    opts.momentum.rtol = rtol
    opts.momentum.precond = 1
    opts.momentum.max_iter = 500
    infos = []
    for step in range(nsteps):
        alpha = fo.bdf_alpha(step, 1.0)
        ctx.set_bdf(alpha, k)
        infos.append(ctx.step_bdf(opts))
        orc.step(alpha, k, vbc)
        ctx.advance(1)
        orc.advance()
    return infos


def test_bdf_monolithic_cavity_matches_oracle():
    """ImplicitBDFSolver path (source/ns_bdf_solver.py:36-106): mixed Newton system solved by
    block-preconditioned BiCGStab vs sparse LU of the same saddle-point matrix."""
    from multigrid import attach_hierarchy
    mesh, dm, marks = box(16, 16)
    ctx = context(mesh, dm)
    attach_hierarchy(ctx, mesh, coarsest=4)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.01, body_force_term=None)
    orc = fo.BDFOracle(s, coef, pin_pressure=True)
    ctx.set_coeffs(1.0, 1.0, 0.01)
    vbc = cavity_bc(dm, marks)
    ctx.set_dirichlet(nat.VELOCITY, *vbc)
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    ctx.set_dirichlet(nat.PRESSURE_PRECOND, np.zeros(0, np.int32), np.zeros(0))
    infos = _run_bdf(ctx, orc, 3, 0.01, vbc)
    nv = dm.n_velocity
    for step, info in enumerate(infos):
        assert info.newton_iterations == orc.newton_its[step]
        hist = orc.newton_history[step]
        for i, r in enumerate(hist[:-1]):
            assert abs(info.newton_residuals[i] - r) <= 1e-6 * r + 1e-12 * hist[0]
    assert rel(ctx.get_state(nat.U1), orc.sol[1][:nv]) < 1e-9
    pg, po = ctx.get_state(nat.P_OLD), orc.sol[1][nv:]
    assert rel(pg - pg.mean(), po - po.mean()) < 1e-8
    ctx.close()


def test_bdf_monolithic_open_channel_and_stokes_limit():
    from multigrid import attach_hierarchy
    mesh, dm, marks = box(32, 4, p1=(8.0, 1.0))
    zero = lambda X: np.zeros((X.shape[0], 2))
    inlet = lambda X: np.stack([6.0 * X[:, 1] * (1.0 - X[:, 1]), 0.0 * X[:, 1]], axis=1)
    vbc = velocity_bc(dm, marks, [(1, inlet), (3, zero), (4, zero)])
    schur = np.unique(dm.facet_p1_nodes(marks.facets_with_id(2))).astype(np.int32)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    for cc, tol in ((1.0, 1e-9), (None, 1e-10)):           # Navier-Stokes, then linear Stokes
        ctx = context(mesh, dm)
        attach_hierarchy(ctx, mesh, coarsest=2)
        coef = dict(convective_term=cc, pressure_term=1.0, viscous_term=0.1, body_force_term=None)
        orc = fo.BDFOracle(s, coef)
        ctx.set_coeffs(cc, 1.0, 0.1)
        ctx.set_dirichlet(nat.VELOCITY, *vbc)
        ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
        ctx.set_dirichlet(nat.PRESSURE_PRECOND, schur, np.zeros(schur.size))
        infos = _run_bdf(ctx, orc, 3, 0.01, vbc, rtol=1e-13)
        if cc is None:
            assert all(i.newton_iterations == 1 for i in infos)
        nv = dm.n_velocity
        assert rel(ctx.get_state(nat.U1), orc.sol[1][:nv]) < tol
        assert rel(ctx.get_state(nat.P_OLD), orc.sol[1][nv:]) < tol     # open outlet fixes the level
        ctx.close()


def test_unstructured_mesh_ipcs_and_bdf_match_oracle():
    """General triangle meshes: vertices of a structured mesh are jittered, and cells and
    vertices randomly renumbered (no lattice, no hierarchy: the multigrid degenerates to the
    two-level P2 -> P1 cycle with a dense P1 solve).  Channel-type BCs, both schemes."""
    from fem_mesh import FacetMarkers, Mesh, TaylorHoodDofMap
    from multigrid import attach_hierarchy
    rng = np.random.default_rng(42)
    base, _, _ = box(14, 10, p1=(1.4, 1.0))
    coords = base.coords.copy()
    interior = (coords[:, 0] > 1e-9) & (coords[:, 0] < 1.4 - 1e-9) & (coords[:, 1] > 1e-9) & \
        (coords[:, 1] < 1.0 - 1e-9)
    coords[interior] += 0.03 * (rng.random((interior.sum(), 2)) - 0.5)
    vperm = rng.permutation(coords.shape[0])                # new id of old vertex
    new_coords = np.empty_like(coords)
    new_coords[vperm] = coords
    cells = vperm[base.cells][rng.permutation(base.cells.shape[0])].astype(np.int32)
    mesh = Mesh(new_coords, cells)
    assert not hasattr(mesh, "structured")
    dm = TaylorHoodDofMap(mesh)
    marks = FacetMarkers(mesh)
    marks.mark(lambda X: np.abs(X[:, 0]) < 1e-12, 1)
    marks.mark(lambda X: np.abs(X[:, 0] - 1.4) < 1e-12, 2)
    marks.mark(lambda X: np.abs(X[:, 1]) < 1e-12, 3)
    marks.mark(lambda X: np.abs(X[:, 1] - 1.0) < 1e-12, 4)
    zero = lambda X: np.zeros((X.shape[0], 2))
    inlet = lambda X: np.stack([4.0 * X[:, 1] * (1.0 - X[:, 1]), 0.0 * X[:, 1]], axis=1)
    vbc = velocity_bc(dm, marks, [(1, inlet), (3, zero), (4, zero)])
    pn = np.unique(dm.facet_p1_nodes(marks.facets_with_id(2)))
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.05, body_force_term=None)
    # ---- IPCS with pressure outlet
    ctx = context(mesh, dm)
    assert attach_hierarchy(ctx, mesh) == 0
    orc = fo.IPCSOracle(s, coef, refactor_every_step=False)
    ctx.set_coeffs(1.0, 1.0, 0.05)
    ctx.set_dirichlet(nat.VELOCITY, *vbc)
    ctx.set_dirichlet(nat.PRESSURE, pn, np.zeros(pn.size))
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = 1e-13
    opts.momentum.precond = opts.poisson.precond = 1
    for step in range(3):
        alpha = fo.bdf_alpha(step, 1.0)
        ctx.set_bdf(alpha, 0.02)
        info = ctx.step_ipcs(opts)
        orc.step(alpha, 0.02, vbc, (pn, np.zeros(pn.size)))
        assert info.newton_iterations == orc.newton_its[step]
        ctx.advance(0)
        orc.advance()
    assert rel(ctx.get_state(nat.U1), orc.vel[1]) < 1e-9
    assert rel(ctx.get_state(nat.P_OLD), orc.p_old) < 1e-9
    ctx.close()
    # ---- monolithic BDF with natural outflow
    ctx = context(mesh, dm)
    attach_hierarchy(ctx, mesh)
    orc = fo.BDFOracle(s, coef)
    ctx.set_coeffs(1.0, 1.0, 0.05)
    ctx.set_dirichlet(nat.VELOCITY, *vbc)
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    ctx.set_dirichlet(nat.PRESSURE_PRECOND, pn.astype(np.int32), np.zeros(pn.size))
    _run_bdf(ctx, orc, 3, 0.02, vbc, rtol=1e-13)
    nv = dm.n_velocity
    assert rel(ctx.get_state(nat.U1), orc.sol[1][:nv]) < 1e-9
    assert rel(ctx.get_state(nat.P_OLD), orc.sol[1][nv:]) < 1e-9
    ctx.close()


@pytest.mark.parametrize("form_id,form", [(0, "standard"), (1, "rotational"), (2, "divergence"),
                                          (3, "skew_symmetric")])
def test_convective_forms_residual_jacobian_and_step(setup16, form_id, form):
    """All four weak forms of source/ns_solver_base.py:370-390: device residual and Newton
    matrix vs the oracle's exact Gateaux derivative, Picard matrix (:478-499), and IPCS steps with
    that form."""
    mesh, dm, marks, ctx, s = setup16
    rng = np.random.default_rng(17 + form_id)
    u = rng.standard_normal(dm.n_velocity)
    zeros_v, zeros_p = np.zeros(dm.n_velocity), np.zeros(dm.n_p1)
    ctx.set_coeffs(0.8, 1.0, 0.02, 1.0)      # (the shared context carries a body force slot)
    ctx.set_bdf((1.5, -2.0, 0.5), 0.05)
    ctx.set_dirichlet(nat.VELOCITY, np.zeros(0, np.int32), np.zeros(0))
    for slot, v in ((nat.U1, zeros_v), (nat.U2, zeros_v), (nat.USTAR, u), (nat.P_OLD, zeros_p),
                    (nat.BODY_FORCE, zeros_v)):
        ctx.set_state(slot, v)
    ctx.set_convective_form(form_id)
    ctx.assemble(nat.SYS_MOMENTUM, new_step=True)
    L = 1.5 / 0.05 * s.vector_mass() + 0.02 * s.vector_stiffness()
    b = L @ u + 0.8 * s.convection_residual(u, form)
    assert rel(ctx.get_rhs(nat.SYS_MOMENTUM), b) < 1e-13
    Jref = L + 0.8 * s.convection_jacobian(u, form)
    J = ctx.operator_csr(nat.OP_MOMENTUM_JAC)
    assert abs(J - Jref).max() <= 1e-13 * abs(Jref).max()
    # matrix-free application (one thread per cell, no assembly) = assembled matrix
    x = rng.standard_normal(dm.n_velocity)
    assert rel(ctx.operator_apply(nat.OP_MOMENTUM_JAC_MF, x), Jref @ x) < 1e-13
    ctx.set_convective_form(form_id, picard=True)
    ctx.assemble(nat.SYS_MOMENTUM)
    Jp = ctx.operator_csr(nat.OP_MOMENTUM_JAC)                 # device Picard matrix, any form
    assert rel(ctx.operator_apply(nat.OP_MOMENTUM_JAC_MF, x), Jp @ x) < 1e-13
    Jpref = L + 0.8 * s.picard_convection(u, form)            # (:478-499), every form
    assert abs(Jp - Jpref).max() <= 1e-13 * abs(Jpref).max()
    ctx.set_convective_form(0)
    # two IPCS steps of the cavity with this form
    m2, dm2, marks2 = box(12, 12)
    c2 = context(m2, dm2)
    s2 = fo.Space(m2.coords, m2.cells, dm2.p2_dofmap, dm2.p1_dofmap)
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.01, body_force_term=None)
    orc = fo.IPCSOracle(s2, coef, form=form, refactor_every_step=False)
    c2.set_coeffs(1.0, 1.0, 0.01)
    vbc = cavity_bc(dm2, marks2)
    c2.set_dirichlet(nat.VELOCITY, *vbc)
    c2.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    opts = c2.default_step_opts()
    opts.convective_form = form_id
    opts.matrix_free = 2 if form_id % 2 else 1      # both Jacobian modes give the same steps
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = 1e-13
    for step in range(2):
        alpha = fo.bdf_alpha(step, 1.0)
        c2.set_bdf(alpha, 0.02)
        info = c2.step_ipcs(opts)
        orc.step(alpha, 0.02, vbc)
        assert info.newton_iterations == orc.newton_its[step]
        c2.advance(0)
        orc.advance()
    assert rel(c2.get_state(nat.U1), orc.vel[1]) < 1e-9
    c2.close()


def test_bdf_rotating_frame_coriolis_and_euler_terms_match_oracle():
    """source/ns_solver_base.py:173-211: 2 c_cor omega (e_z x u, w) in residual and Jacobian,
    c_e omega' (e_z x x, w) on the right-hand side -- two BDF steps of a lid-driven cavity in a
    rotating frame against the LU oracle."""
    from multigrid import attach_hierarchy
    mesh, dm, marks = box(8, 8)
    ctx = context(mesh, dm)
    attach_hierarchy(ctx, mesh, coarsest=2)
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.05, body_force_term=None,
                coriolis_term=1.5, euler_term=0.7)
    ctx.set_coeffs(1.0, 1.0, 0.05, None, 1.5, 0.7)
    bd, bv = cavity_bc(dm, marks)
    ctx.set_dirichlet(nat.VELOCITY, bd, bv)
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    ctx.set_dirichlet(nat.PRESSURE_PRECOND, np.zeros(0, np.int32), np.zeros(0))
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    orc = fo.BDFOracle(s, coef, pin_pressure=True)
    o = ctx.default_step_opts()
    o.momentum.rtol, o.momentum.precond = 1e-13, 1
    for step, (omega, omega_dot) in enumerate([(0.8, 0.3), (1.1, -0.4)]):
        ctx.set_angular_velocity(omega, omega_dot)
        orc.omega, orc.omega_dot = omega, omega_dot
        alpha = fo.bdf_alpha(step, 1.0)
        ctx.set_bdf(alpha, 0.05)
        info = ctx.step_bdf(o)
        ctx.advance(1)
        orc.step(alpha, 0.05, (bd, bv))
        orc.advance()
        assert info.newton_iterations == orc.newton_its[-1]
    nv = dm.n_velocity
    assert rel(ctx.get_state(nat.U1), orc.sol[1][:nv]) < 1e-9
    p, po = ctx.get_state(nat.P_OLD), orc.sol[1][nv:]
    assert rel(p - p.mean(), po - po.mean()) < 1e-8
    ctx.close()


@pytest.mark.parametrize("scheme", ["ipcs", "bdf"])
def test_inexact_newton_reaches_the_reference_criterion_and_the_same_fields(scheme):
    """bench.py's throughput settings (newton_forcing = 1e-4, Krylov rtol 1e-8): every step still
    ends on the reference's Newton criterion, evaluated on the true nonlinear residual, and the
    fields agree with the LU oracle far inside north_star's 1e-6."""
    from multigrid import attach_hierarchy
    mesh, dm, marks = box(16, 16)
    ctx = context(mesh, dm)
    attach_hierarchy(ctx, mesh, coarsest=2)
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.01, body_force_term=None)
    ctx.set_coeffs(1.0, 1.0, 0.01)
    bd, bv = cavity_bc(dm, marks)
    ctx.set_dirichlet(nat.VELOCITY, bd, bv)
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    ctx.set_dirichlet(nat.PRESSURE_PRECOND, np.zeros(0, np.int32), np.zeros(0))
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    orc = fo.IPCSOracle(s, coef) if scheme == "ipcs" else fo.BDFOracle(s, coef, pin_pressure=True)
    o = ctx.default_step_opts()
    for k in (o.momentum, o.poisson, o.correction):
        k.rtol = 1e-8
    o.momentum.precond = o.poisson.precond = 1
    o.newton_forcing = 1e-4
    exact_its = inexact_its = 0
    for step in range(4):
        alpha = fo.bdf_alpha(step, 1.0)
        ctx.set_bdf(alpha, 0.01)
        info = ctx.step_ipcs(o) if scheme == "ipcs" else ctx.step_bdf(o)
        ctx.advance(0 if scheme == "ipcs" else 1)
        orc.step(alpha, 0.01, (bd, bv))
        orc.advance()
        res = [info.newton_residuals[i] for i in range(info.newton_iterations + 1)]
        assert info.converged and (res[-1] < 1e-10 or res[-1] / res[0] < 1e-9)
        inexact_its += info.krylov_iterations_momentum
    nv = dm.n_velocity
    if scheme == "ipcs":
        uo, po = orc.vel[1], orc.p_old
    else:
        uo, po = orc.sol[1][:nv], orc.sol[1][nv:]
    assert rel(ctx.get_state(nat.U1), uo) < 1e-7
    p = ctx.get_state(nat.P_OLD)
    assert rel(p - p.mean(), po - po.mean()) < 1e-6
    ctx.close()


@pytest.mark.parametrize("poisson", ["fast_diagonalization", "multigrid"])
def test_timed_settings_on_the_lattice_kernels_match_the_oracle_at_config0_size(poisson):
    """What bench.py times -- ALL throughput knobs together (Krylov rtol 1e-8, Newton forcing 1e-4, truncated
    velocity cycle, Chebyshev mass solve, and the projection step either by fast diagonalisation on the matrix cores
    = the bench default, or by multigrid-CG with the extrapolated pressure start = `--poisson-solver mg` and every
    partitioned run) on the one-launch lattice kernels
    (k_cheb_lattice with fused transfers, k_jac_lattice) -- against the LU oracle at BASELINE configs[0]'s size,
    64 x 64 cells (129 x 129 P2 lattice: 3 x 11 tiles of the lattice smoother), 4 steps of the Re = 100 cavity.
    Velocity and pressure agree to north_star's 1e-6 and every step ends on the reference's Newton criterion
    evaluated on the true nonlinear residual (source/ns_ipcs_solver.py:144-147, 198-208)."""
    from multigrid import attach_hierarchy
    n = 64
    mesh, dm, marks = box(n, n)
    mesh.structured = ((0.0, 0.0), (1.0, 1.0), n, n)
    ctx = context(mesh, dm)
    attach_hierarchy(ctx, mesh)
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.01, body_force_term=None)
    ctx.set_coeffs(1.0, 1.0, 0.01)
    bd, bv = cavity_bc(dm, marks)
    ctx.set_dirichlet(nat.VELOCITY, bd, bv)
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    ctx.mg_set_truncation(4.0, 0.1)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    orc = fo.IPCSOracle(s, coef, refactor_every_step=False)
    o = ctx.default_step_opts()
    for k in (o.momentum, o.poisson, o.correction):
        k.rtol = 1e-8
    o.momentum.precond = o.poisson.precond = 1
    if poisson == "fast_diagonalization":
        import poisson_fd as pf
        ctx.poisson_set_fast_diag(pf.factors(*pf.lattice_lines(mesh), np.zeros(0, np.int32)))
        o.poisson.precond = 3
    o.correction.precond = 2                         # Chebyshev mass solve with a-priori bounds
    o.newton_forcing = 1e-4
    o.pressure_extrapolation = 1
    dt = 8.0e-3                                      # the step size of the n = 512 run scaled with h (same CFL)
    for step in range(4):
        alpha = fo.bdf_alpha(step, 1.0)
        ctx.set_bdf(alpha, dt)
        info = ctx.step_ipcs(o)
        ctx.advance(0)
        orc.step(alpha, dt, (bd, bv))
        orc.advance()
        res = [info.newton_residuals[i] for i in range(info.newton_iterations + 1)]
        assert info.converged and (res[-1] < 1e-10 or res[-1] / res[0] < 1e-9), res
    assert ctx.smoother_info()["multistep_lattice_kernel"]
    ji = ctx.jacobian_info()
    assert ji["path"] == "lattice-kernel" and ji["lattice_launches"] > 0
    assert rel(ctx.get_state(nat.U1), orc.vel[1]) < 1e-6
    p = ctx.get_state(nat.P_OLD)
    assert rel(p - p.mean(), orc.p_old - orc.p_old.mean()) < 1e-6
    ctx.close()


@pytest.mark.parametrize("nx,ny,outlet", [(32, 32, False), (40, 24, True), (64, 64, False)])
def test_fast_diagonalisation_projection_step_matches_the_oracle(nx, ny, outlet):
    """Projection step by fast diagonalisation (nsfem_krylov_opts.precond = 3: four dense products on the matrix cores,
    csrc/fastdiag.hip) in place of the multigrid-CG solve: the device product z = A^+ r equals the numpy evaluation of
    the same factors, and IPCS steps of the cavity (closed: singular Neumann problem; with an outflow side: pressure
    Dirichlet on a whole side) match the LU oracle (source/ns_ipcs_solver.py:160-171) as tightly as the CG path run
    with rtol 1e-12 does -- in ONE pass of the direct solve."""
    import poisson_fd as pf
    from multigrid import attach_hierarchy
    ext = (nx / float(max(nx, ny)), ny / float(max(nx, ny)))
    mesh, dm, marks = box(nx, ny, p1=ext)
    mesh.structured = ((0.0, 0.0), ext, nx, ny)
    bd, bv = cavity_bc(dm, marks)
    nodes = np.zeros(0, np.int32)
    if outlet:
        nodes = np.where(np.abs(mesh.coords[:, 0] - ext[0]) < 1e-12)[0].astype(np.int32)
    ctx = context(mesh, dm)
    attach_hierarchy(ctx, mesh)
    ctx.set_coeffs(1.0, 1.0, 0.01)
    ctx.set_dirichlet(nat.VELOCITY, bd, bv)
    ctx.set_dirichlet(nat.PRESSURE, nodes, np.zeros(nodes.size))
    f = pf.factors(*pf.lattice_lines(mesh), nodes)
    ctx.poisson_set_fast_diag(f)
    r = np.random.default_rng(nx).standard_normal(dm.n_p1)
    r[nodes] = 0.0
    if not outlet:
        r -= r.mean()
    assert rel(ctx.mg_apply(2, r), pf.apply_reference(f, r)) < 1e-13
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.01, body_force_term=None)
    orc = fo.IPCSOracle(s, coef, refactor_every_step=False)
    o = ctx.default_step_opts()
    o.momentum.precond = 1
    o.poisson.precond = 3
    pbc = (nodes.astype(np.int64), np.zeros(nodes.size)) if outlet else None
    for step in range(3):
        alpha = fo.bdf_alpha(step, 1.0)
        ctx.set_bdf(alpha, 0.01)
        info = ctx.step_ipcs(o)
        ctx.advance(0)
        orc.step(alpha, 0.01, (bd, bv), pbc) if outlet else orc.step(alpha, 0.01, (bd, bv))
        orc.advance()
        assert info.krylov_iterations_poisson == 1
    assert rel(ctx.get_state(nat.U1), orc.vel[1]) < 1e-9
    p = ctx.get_state(nat.P_OLD)
    if outlet:
        assert rel(p, orc.p_old) < 1e-8
    else:
        assert rel(p - p.mean(), orc.p_old - orc.p_old.mean()) < 1e-8
    ctx.close()


def test_truncated_velocity_cycle_is_only_a_preconditioner_change():
    """nsfem_mg_set_truncation: at a small time step the velocity operator alpha0/k M + c_v K is
    mass dominated on the coarser levels; the cycle stops at the first such level and solves it by
    Chebyshev iteration with a-priori bounds.  Same converged fields, about the same Krylov
    iteration counts as the full cycle."""
    from multigrid import attach_hierarchy
    mesh, dm, marks = box(64, 64)
    res = {}
    for ratio in (0.0, 4.0, 1.0e6):
        ctx = context(mesh, dm)
        attach_hierarchy(ctx, mesh, coarsest=4)
        ctx.mg_set_truncation(ratio, 0.1)
        ctx.set_coeffs(1.0, 1.0, 0.01)
        ctx.set_dirichlet(nat.VELOCITY, *cavity_bc(dm, marks))
        ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
        opts = ctx.default_step_opts()
        for o in (opts.momentum, opts.poisson, opts.correction):
            o.rtol = 1e-12
        opts.momentum.precond = opts.poisson.precond = 1
        its = 0
        for step in range(3):
            ctx.set_bdf(fo.bdf_alpha(step, 1.0), 2e-3)
            info = ctx.step_ipcs(opts)
            its += info.krylov_iterations_momentum
            ctx.advance(0)
        res[ratio] = (ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD), its)
        ctx.close()
    for ratio in (4.0, 1.0e6):
        assert rel(res[ratio][0], res[0.0][0]) < 1e-9
        assert rel(res[ratio][1], res[0.0][1]) < 1e-8
    assert res[4.0][2] <= res[0.0][2] + 6
    with pytest.raises(nat.NativeError):
        context(mesh, dm).mg_set_truncation(-1.0, 0.1)


# ------------------------------------------------ boundary functionals (drag / lift / flux)
def _random_state(dm, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(dm.n_velocity), rng.standard_normal(dm.n_p1)


@pytest.mark.parametrize("nu,sym", [(0.5 / 100.0, 1.0), (0.013, 0.0)])
def test_boundary_force_matches_oracle_on_the_dfg_cylinder(nu, sym):
    """nsfem_boundary_force (one thread per facet, 2-point Gauss on the edge) against the oracle's
    facet integral (4-point Gauss on the physical edge, pulled back through J^-1) on the curved
    cylinder boundary of the small DFG channel mesh -- the drag / lift functional of the
    reference's demo/dfg_benchmark.py:44-66 -- and on the whole boundary (mass flux of
    demo/gravity_driven_flow.py:66-70); random nodal fields, so nothing cancels."""
    import grid_generator as gg
    mesh, marks = gg.dfg_channel(2, 1)
    dm = TaylorHoodDofMapOf(mesh)
    ctx = context(mesh, dm)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    u, p = _random_state(dm, 3)
    ctx.set_state(nat.U0, u)
    ctx.set_state(nat.P, p)
    for facets in (marks.facets_with_id(gg.DFGBoundaryMarkers.cylinder.value),
                   np.nonzero(mesh.facet_on_boundary)[0]):
        fc, fl = mesh.facet_cell_local(facets)
        force, flux, meas = ctx.boundary_force(fc, fl, nu, sym)
        f_o, flux_o, meas_o = fo.boundary_functionals(s, mesh.facets[facets], mesh.facet_cell[facets], u, p, nu, sym)
        assert np.abs(force - f_o).max() < 1e-12 * max(1.0, np.abs(f_o).max())
        assert abs(flux - flux_o) < 1e-12 * max(1.0, abs(flux_o))
        assert abs(meas - meas_o) < 1e-13 * meas_o
    # Gauss' theorem on the discrete field: total boundary flux = sum of the divergence rows
    # tested with the constant pressure function
    assert abs(flux - ctx.operator_apply(nat.OP_DIV, u).sum()) < 1e-10
    # the polygonal cylinder of this mesh has the perimeter of its chords
    fcyl = marks.facets_with_id(gg.DFGBoundaryMarkers.cylinder.value)
    fc, fl = mesh.facet_cell_local(fcyl)
    _, _, perim = ctx.boundary_force(fc, fl, 0.0, 0.0)
    assert 0.95 * np.pi < perim < np.pi
    ctx.close()


def TaylorHoodDofMapOf(mesh):
    from fem_mesh import TaylorHoodDofMap
    return TaylorHoodDofMap(mesh)


def test_n512_production_solver_options_match_the_exact_solver_run():
    """What bench.py times, at the size it times it (BASELINE configs[1], 2,364,419 dofs):
    geometric multigrid (V(0,3) momentum / V(2,2) Poisson), truncated velocity cycle (4, 0.1),
    Chebyshev mass solve, inexact Newton (forcing 1e-4), Krylov rtol 1e-8, iteration hints --
    against the same mesh and steps with direct-solver accuracy (Krylov rtol 1e-12, exact Newton,
    Jacobi-CG mass solve, untruncated cycle).  Fields must agree to north_star's nonlinear
    tolerance 1e-6 (pressure modulo a constant: enclosed flow)."""
    from multigrid import attach_hierarchy
    mesh, dm, marks = box(512, 512)
    ctx = context(mesh, dm)
    assert attach_hierarchy(ctx, mesh) == 4
    ctx.set_coeffs(1.0, 1.0, 0.01)
    bd, bv = cavity_bc(dm, marks)
    ctx.set_dirichlet(nat.VELOCITY, bd, bv)
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    n_steps, k = 6, 1.0e-3

    def run(production):
        for slot in (nat.U0, nat.U1, nat.U2, nat.USTAR):
            ctx.set_state(slot, np.zeros(dm.n_velocity))
        for slot in (nat.P, nat.P_OLD):
            ctx.set_state(slot, np.zeros(dm.n_p1))
        opts = ctx.default_step_opts()
        opts.momentum.precond = opts.poisson.precond = 1
        if production:
            ctx.mg_set_truncation(4.0, 0.1)
            for o in (opts.momentum, opts.poisson, opts.correction):
                o.rtol = 1.0e-8
            opts.correction.precond = 2
            opts.newton_forcing = 1.0e-4
        else:
            ctx.mg_set_truncation(0.0, 0.1)
        its = []
        for step in range(n_steps):
            ctx.set_bdf(fo.bdf_alpha(step, 1.0), k)
            info = ctx.step_ipcs(opts)
            assert info.converged
            r = info.newton_residuals
            assert r[info.newton_iterations] < max(1e-10, 1e-9 * r[0])      # the reference's criterion
            its.append((info.newton_iterations, info.krylov_iterations_momentum,
                        info.krylov_iterations_poisson, info.krylov_iterations_correction))
            ctx.advance(0)
        return ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD), its

    u_fast, p_fast, its_fast = run(True)
    u_ref, p_ref, its_ref = run(False)
    assert np.abs(u_fast[bd] - bv).max() == 0.0
    assert rel(u_fast, u_ref) < 1e-6
    assert rel(p_fast - p_fast.mean(), p_ref - p_ref.mean()) < 1e-6
    # the production settings are the cheaper ones: fewer Krylov iterations in total
    assert sum(i[1] for i in its_fast) < sum(i[1] for i in its_ref)
    ctx.close()


@pytest.mark.parametrize("dim,n", [(2, 96), (3, 16)])
def test_sell_kernel_on_parity_numbering_equals_csr_kernel_on_lattice_numbering(dim, n):
    """NSFEM_SELL=2 (read when the first pattern is built -- this test runs in a child process; the
    default, 1, uses the kernel for tetrahedral P2 operators only): with the parity-class numbering
    of structured meshes
    (TaylorHoodDofMap(reorder="parity")) the scalar P2 / P1 operators use the SELL-64 SpMV kernel
    (k_spmv_sell: one wavefront per 64 rows, coalesced x gather); the lexicographic numbering keeps
    the CSR-stream kernel.  Same mesh, same problem, both numberings: products against the exported
    CSR matrices, then multigrid-preconditioned IPCS steps (every epilogue of the kernel: store,
    residual, Chebyshev smoothing step, masked rows) -- the fields agree node by node to solver
    tolerance."""
    import os
    import subprocess
    import sys
    if os.environ.get("NSFEM_SELL") != "2":            # 2: SELL on every qualifying pattern (2D too)
        env = dict(os.environ, NSFEM_SELL="2", NSFEM_DICT="0")     # (the dictionary kernel would pre-empt SELL)
        here = os.path.abspath(__file__)
        node = "%s::test_sell_kernel_on_parity_numbering_equals_csr_kernel_on_lattice_numbering[%d-%d]" % (here, dim, n)
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", node],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        assert "1 passed" in r.stdout
        return
    from fem_mesh import TaylorHoodDofMap, box_mesh, rectangle_mesh
    from multigrid import attach_hierarchy
    mesh = rectangle_mesh((0.0, 0.0), (1.0, 1.0), n, n) if dim == 2 else box_mesh((0, 0, 0), (1, 1, 1), n, n, n)
    rng = np.random.default_rng(7)
    out = {}
    for mode in ("parity", True):
        dm = TaylorHoodDofMap(mesh, reorder=mode)
        ctx = context(mesh, dm)
        for op in (nat.OP_MASS_P2, nat.OP_STIFF_P2, nat.OP_STIFF_P1):
            A = ctx.operator_csr(op)
            x = rng.standard_normal(A.shape[1])
            assert rel(ctx.operator_apply(op, x), A @ x) < 1e-13
        attach_hierarchy(ctx, mesh, coarsest=4 if dim == 2 else 2)
        X = dm.p2_coords
        on = np.zeros(dm.n_p2, bool)
        for a in range(dim):
            on |= (np.abs(X[:, a]) < 1e-12) | (np.abs(X[:, a] - 1.0) < 1e-12)
        nodes = np.nonzero(on)[0]
        lid = np.abs(X[nodes, dim - 1] - 1.0) < 1e-12
        dofs = np.concatenate([dim * nodes + a for a in range(dim)]).astype(np.int32)
        vals = np.concatenate([np.where(lid, 1.0, 0.0)] + [np.zeros(nodes.size)] * (dim - 1))
        ctx.set_coeffs(1.0, 1.0, 0.01)
        ctx.set_dirichlet(nat.VELOCITY, dofs, vals)
        ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
        opts = ctx.default_step_opts()
        opts.momentum.precond = opts.poisson.precond = 1
        opts.correction.precond = 2
        its = []
        for step in range(3):
            ctx.set_bdf(fo.bdf_alpha(step, 1.0), 0.5 / n)
            info = ctx.step_ipcs(opts)
            its.append((info.newton_iterations, info.krylov_iterations_momentum, info.krylov_iterations_poisson))
            ctx.advance(0)
        # node-by-node comparison through the coordinates (the numberings differ)
        key = np.lexsort(tuple(np.round(X[:, a] * 4 * n).astype(np.int64) for a in range(dim)))
        out[mode] = (ctx.get_state(nat.U1).reshape(-1, dim)[key], ctx.get_state(nat.P_OLD), its)
        ctx.close()
    (ua, pa, ia), (ub, pb, ib) = out["parity"], out[True]
    assert rel(ua, ub) < 1e-9 and rel(pa - pa.mean(), pb - pb.mean()) < 1e-8
    for a, b in zip(ia, ib):
        assert a[0] == b[0] and abs(a[1] - b[1]) <= 1 and abs(a[2] - b[2]) <= 1


@pytest.mark.parametrize("dim,n,exact,order", [(2, 64, True, True), (2, 48, False, True), (3, 32, True, "parity"),
                                                (3, 24, False, True)])
def test_stencil_dictionary_kernel_equals_the_csr_kernels(dim, n, exact, order, monkeypatch):
    """Lattice meshes: the scalar P2 / P1 operators are stored a second time as a stencil dictionary
    (rows with equal column offsets and values share one entry; k_spmv_dict reads one byte per row
    instead of 12 bytes per nonzero).  With a binary mesh spacing (n = 64, 32) every row equals its
    representative BIT FOR BIT and every product uses the dictionary; otherwise (n = 48, 24: equal
    to 2^-40) only smoothing steps and Newton-Jacobian products do.  (The slab-blocked parity
    numbering of tetrahedral meshes needs n >= 32 for a workgroup to see <= 32 distinct rows; the
    lexicographic numbering has 64 distinct rows at any size.)  Same problem with
    NSFEM_DICT=0 (CSR-stream / SELL kernels) and with the dictionary: same iteration counts, fields
    equal to solver tolerance; exported CSR matrices against dictionary products; an unstructured
    mesh gets no dictionary."""
    import grid_generator as gg
    from fem_mesh import TaylorHoodDofMap, box_mesh, rectangle_mesh
    from multigrid import attach_hierarchy
    mesh = rectangle_mesh((0.0, 0.0), (1.0, 1.0), n, n) if dim == 2 else box_mesh((0, 0, 0), (1, 1, 1), n, n, n)
    mesh.structured = ((0.0,) * dim, (1.0,) * dim) + (n,) * dim
    dm = TaylorHoodDofMap(mesh, reorder=order)
    X = dm.p2_coords
    on = np.zeros(dm.n_p2, bool)
    for a in range(dim):
        on |= (np.abs(X[:, a]) < 1e-12) | (np.abs(X[:, a] - 1.0) < 1e-12)
    nodes = np.nonzero(on)[0]
    lid = np.abs(X[nodes, dim - 1] - 1.0) < 1e-12
    dofs = np.concatenate([dim * nodes + a for a in range(dim)]).astype(np.int32)
    vals = np.concatenate([np.where(lid, 1.0, 0.0)] + [np.zeros(nodes.size)] * (dim - 1))
    rng = np.random.default_rng(5)
    xm = rng.standard_normal(dm.n_p2)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("NSFEM_DICT", mode)
        ctx = context(mesh, dm)
        attach_hierarchy(ctx, mesh, coarsest=4 if dim == 2 else 2)
        ctx.set_coeffs(1.0, 1.0, 0.01)
        ctx.set_dirichlet(nat.VELOCITY, dofs, vals)
        ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
        opts = ctx.default_step_opts()
        for o in (opts.momentum, opts.poisson, opts.correction):
            o.rtol = 1e-11
        opts.momentum.precond = opts.poisson.precond = 1
        opts.correction.precond = 2
        its = []
        for step in range(3):
            ctx.set_bdf((1.0, -1.0, 0.0) if step == 0 else (1.5, -2.0, 0.5), 0.01)
            info = ctx.step_ipcs(opts)
            ctx.advance(0)
            its.append((info.newton_iterations, info.krylov_iterations_momentum, info.krylov_iterations_poisson))
        info_s = ctx.smoother_info()
        # products with the mass matrix after the steps (the dictionary is attached by then)
        M = ctx.operator_csr(nat.OP_MASS_P2)
        assert rel(ctx.operator_apply(nat.OP_MASS_P2, xm), M @ xm) < 1e-13
        out[mode] = (ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD), its, info_s)
        ctx.close()
    u0, p0, its0, s0 = out["0"]
    u1, p1, its1, s1 = out["1"]
    assert s0["kind"] != "stencil-dictionary"
    assert s1["kind"] == "stencil-dictionary" and s1["bitwise_exact"] == exact
    assert 0 < s1["stencils"] < dm.n_p2 // 8 and s1["csr_bytes"] > 0
    for a, b in zip(its0, its1):
        assert a[0] == b[0] and abs(a[1] - b[1]) <= 1 and abs(a[2] - b[2]) <= 1
    assert rel(u1, u0) < 1e-9
    assert rel(p1 - p1.mean(), p0 - p0.mean()) < 1e-8
    if dim == 2 and exact:
        monkeypatch.setenv("NSFEM_DICT", "1")
        dmesh, _ = gg.dfg_channel(4, 3)
        ddm = TaylorHoodDofMap(dmesh)
        c = context(dmesh, ddm)
        attach_hierarchy(c, dmesh)
        c.set_coeffs(1.0, 1.0, 0.01)
        c.set_bdf((1.0, -1.0, 0.0), 0.01)
        assert ddm.n_p2 > 4096 and c.smoother_info()["kind"] != "stencil-dictionary"
        c.close()


@pytest.mark.parametrize("dim,n", [(2, 64), (2, 48), (3, 32)])
def test_block_dictionaries_of_the_monolithic_scheme(dim, n, monkeypatch):
    """Monolithic BDF-2 steps on lattice meshes: the divergence (1 x dim blocks) and its transpose
    (dim x 1) get block stencil dictionaries (k_spmv_dict_blk: column = first column of the row +
    offset, because the P2 and P1 numberings differ) next to the scalar ones; they serve the
    Newton-Jacobian products and the block preconditioner -- and, bit for bit equal (n = 64, 32),
    the residual too.  NSFEM_DICT=0 vs default: same Newton counts, BiCGStab counts +-1, fields to
    solver tolerance."""
    from fem_mesh import TaylorHoodDofMap, box_mesh, preferred_p2_order, rectangle_mesh
    from multigrid import attach_hierarchy
    mesh = rectangle_mesh((0.0, 0.0), (1.0, 1.0), n, n) if dim == 2 else box_mesh((0, 0, 0), (1, 1, 1), n, n, n)
    mesh.structured = ((0.0,) * dim, (1.0,) * dim) + (n,) * dim
    dm = TaylorHoodDofMap(mesh, reorder=preferred_p2_order(dim))
    X = dm.p2_coords
    on = np.zeros(dm.n_p2, bool)
    for a in range(dim):
        on |= (np.abs(X[:, a]) < 1e-12) | (np.abs(X[:, a] - 1.0) < 1e-12)
    nodes = np.nonzero(on)[0]
    lid = np.abs(X[nodes, dim - 1] - 1.0) < 1e-12
    dofs = np.concatenate([dim * nodes + a for a in range(dim)]).astype(np.int32)
    vals = np.concatenate([np.where(lid, 1.0, 0.0)] + [np.zeros(nodes.size)] * (dim - 1))
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("NSFEM_DICT", mode)
        ctx = context(mesh, dm)
        attach_hierarchy(ctx, mesh, coarsest=4 if dim == 2 else 2)
        ctx.set_coeffs(1.0, 1.0, 0.01)
        ctx.set_dirichlet(nat.VELOCITY, dofs, vals)
        ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
        ctx.set_dirichlet(nat.PRESSURE_PRECOND, np.zeros(0, np.int32), np.zeros(0))
        opts = ctx.default_step_opts()
        opts.momentum.rtol, opts.momentum.precond = 1e-11, 1
        its = []
        for step in range(2):
            ctx.set_bdf((1.0, -1.0, 0.0) if step == 0 else (1.5, -2.0, 0.5), 0.01)
            info = ctx.step_bdf(opts)
            ctx.advance(1)
            its.append((info.newton_iterations, info.krylov_iterations_momentum))
        # the divergence through the C ABI against its exported CSR matrix (dictionary attached by now)
        D = ctx.operator_csr(nat.OP_DIV)
        x = np.random.default_rng(1).standard_normal(D.shape[1])
        assert rel(ctx.operator_apply(nat.OP_DIV, x), D @ x) < 1e-13
        out[mode] = (ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD), its)
        ctx.close()
    (u0, p0, its0), (u1, p1, its1) = out["0"], out["1"]
    for a, b in zip(its0, its1):
        assert a[0] == b[0] and abs(a[1] - b[1]) <= max(1, 0.05 * a[1])
    assert rel(u1, u0) < 1e-9
    assert rel(p1 - p1.mean(), p0 - p0.mean()) < 1e-7


@pytest.mark.parametrize("kind", ["2d", "3d", "unstructured"])
def test_device_built_patterns_equal_the_host_built_ones(kind, monkeypatch):
    """Round 4: sparsity patterns, slot maps and inverted indices come from one radix sort per pattern on the device
    (csrc/pattern_device.hip) instead of the threaded host construction (csrc/pattern.cpp; what dolfin's
    SparsityPatternBuilder does inside the reference's variational solvers, source/ns_ipcs_solver.py:136-147).  The
    two must agree array for array: every exported operator (pattern AND values: the values go through the slot map
    and the inverted index) is bitwise the same, and so is an IPCS step."""
    import fem_mesh as fm
    if kind == "2d":
        mesh, dm, marks = box(12, 9, p1=(1.0, 0.75))
    elif kind == "3d":
        mesh = fm.box_mesh((0.0, 0.0, 0.0), (1.0, 0.5, 0.75), 4, 3, 5)
        dm = fm.TaylorHoodDofMap(mesh, reorder="parity")
    else:
        mesh, dm, marks = box(10, 10)
        rng = np.random.default_rng(5)
        inner = np.nonzero((np.abs(mesh.coords - 0.5) < 0.49).all(axis=1))[0]
        coords = mesh.coords.copy()
        coords[inner] += 0.02 * rng.standard_normal((inner.size, 2))
        mesh = fm.Mesh(coords, mesh.cells)
        dm = fm.TaylorHoodDofMap(mesh)
    dim = mesh.coords.shape[1]
    X = dm.p2_coords
    wall = np.nonzero((np.abs(X - X.min(axis=0)) < 1e-12).any(axis=1) | (np.abs(X - X.max(axis=0)) < 1e-12).any(axis=1))[0]
    lid = np.abs(X[wall, dim - 1] - X[:, dim - 1].max()) < 1e-12
    bd = np.concatenate([dim * wall + a for a in range(dim)]).astype(np.int32)
    bv = np.concatenate([np.where(lid, 1.0, 0.0)] + [np.zeros(wall.size)] * (dim - 1))
    results = []
    for host in (True, False):
        if host:
            monkeypatch.setenv("NSFEM_PATTERN_HOST", "1")
        else:
            monkeypatch.delenv("NSFEM_PATTERN_HOST", raising=False)
        ctx = context(mesh, dm)
        ops = {}
        for name in ("OP_MASS_P2", "OP_STIFF_P2", "OP_STIFF_P1", "OP_MASS_P1", "OP_DIV", "OP_GRAD", "OP_DIVT"):
            A = ctx.operator_csr(getattr(nat, name))
            ops[name] = (A.indptr.copy(), A.indices.copy(), A.data.copy())
        ctx.set_coeffs(1.0, 1.0, 0.01)
        ctx.set_dirichlet(nat.VELOCITY, bd, bv)
        ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
        o = ctx.default_step_opts()
        o.matrix_free = 1                                   # the assembled Jacobian: slot map + inverted index at work
        for step in range(2):
            ctx.set_bdf(fo.bdf_alpha(step, 1.0), 0.01)
            info = ctx.step_ipcs(o)
            assert info.converged
            ctx.advance(0)
        results.append((ops, ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD)))
        ctx.close()
    (ops_h, u_h, p_h), (ops_d, u_d, p_d) = results
    for name in ops_h:
        for a, b in zip(ops_h[name], ops_d[name]):
            assert np.array_equal(a, b), name
    assert np.array_equal(u_h, u_d) and np.array_equal(p_h, p_d)
