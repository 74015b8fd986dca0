"""Every SpMV kernel family of the lattice workloads pinned to the ORACLE's matrices directly.

The kernels that carry the structured-mesh configurations of BASELINE.json -- the stencil-dictionary
kernels k_spmv_dict / k_spmv_dict_w8 (one smoothing step per launch), the multi-step lattice kernel
k_cheb_lattice (several Chebyshev-Jacobi steps of a lattice operator in one launch, iterate staged in
LDS), SELL-64 and the CSR kernels -- are driven through the C ABI test hook nsfem_kernel_apply on
seeded data and compared with products of `fo.Space(...).mass_p2() / stiffness_p2() / mass_p1() /
stiffness_p1()` computed by the oracle (scipy), not with matrices exported from the device: a wrong
dictionary entry, offset, diagonal or mask branch fails here.

Tolerance: 1e-13 relative (fp64 round-off of sums of <= 65 products) where the dictionary equals the
CSR values bit for bit (binary mesh spacing), 2^-38 |A| |x| where it is only equal to 2^-40 (n = 48)."""
import numpy as np
import pytest

import _native as nat
import fem_oracle as fo
from gpu_common import context, rel

pytestmark = pytest.mark.gpu

CSR, SELL, DICT, LATTICE = 1, 2, 3, 4


def _lattice_case(dim, n, order=True):
    from fem_mesh import TaylorHoodDofMap, box_mesh, rectangle_mesh
    mesh = rectangle_mesh((0.0, 0.0), (1.0, 1.0), n, n) if dim == 2 else box_mesh((0, 0, 0), (1, 1, 1), n, n, n)
    mesh.structured = ((0.0,) * dim, (1.0,) * dim) + (n,) * dim
    dm = TaylorHoodDofMap(mesh, reorder=order)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    return mesh, dm, s


def _cheb_reference(A, nv, x, b, d, mask, c1, c2, ident=False, from_zero=False):
    """the smoothing sequence in numpy on the oracle's matrix: d = c1 d + c2 D^-1 (b - A x), x += d;
    rows flagged 1: x = d = 0 (x = b in the last step when `ident`)"""
    n = A.shape[0]
    X = (np.zeros((n, nv)) if from_zero else x.reshape(n, nv).copy())
    B, D = b.reshape(n, nv), (np.zeros((n, nv)) if (d is None or from_zero) else d.reshape(n, nv).copy())
    M = np.zeros((n, nv), bool) if mask is None else (mask.reshape(n, nv) == 1)
    dinv = 1.0 / A.diagonal()
    for k in range(len(c1)):
        R = B - A @ X
        D = np.where(M, 0.0, c1[k] * D + c2[k] * dinv[:, None] * R)
        X = np.where(M, 0.0, X + D)
        if ident and k == len(c1) - 1:
            X = np.where(M, B, X)
    return X.ravel(), D.ravel()


def _random_case(rng, n, nv, with_mask=True):
    x, b, d = (rng.standard_normal(n * nv) for _ in range(3))
    mask = (rng.random(n * nv) < 0.07).astype(np.uint8) if with_mask else None
    return x, b, d, mask


@pytest.mark.parametrize("dim,n,exact", [(2, 64, True), (2, 48, False), (3, 16, True)])
@pytest.mark.parametrize("space", [0, 1])
def test_products_and_residuals_of_every_family_match_the_oracle(dim, n, exact, space):
    """y = (a M + b K) x and y = b - A x through the CSR kernels, the stencil dictionary (and SELL-64 where the
    pattern has the layout), nv = 1, 2, 3 interleaved components, with identity / zero row masks"""
    mesh, dm, s = _lattice_case(dim, n)
    ctx = context(mesh, dm)
    a, bc = 1.7, 0.31
    A = (a * (s.mass_p2() if space == 0 else s.mass_p1()) + bc * (s.stiffness_p2() if space == 0 else s.stiffness_p1())).tocsr()
    nrows = A.shape[0]
    rng = np.random.default_rng(7 + dim + n)
    scale = abs(A).max()
    for nv in (1, 2, 3):
        x, b, _, mask = _random_case(rng, nrows, nv)
        X = x.reshape(nrows, nv)
        ref = (A @ X).ravel()
        for family in (CSR, DICT):
            out = ctx.kernel_apply(space, nv, x, a=a, b_coef=bc, family=family, epilogue=0)
            assert out["used_family"] == family and out["dict_entries"] > 0
            assert out["dict_exact"] == exact
            tol = 1e-13 if (exact or family == CSR) else 2.0 ** -38
            assert np.abs(out["y"] - ref).max() <= tol * scale * np.abs(x).max() * 65, (family, nv)
            # masked product: identity rows keep x, zero rows give 0
            for mode in (1, 2):
                out = ctx.kernel_apply(space, nv, x, a=a, b_coef=bc, family=family, epilogue=0, mask=mask, maskmode=mode)
                want = np.where(mask == 1, x if mode == 1 else 0.0, ref)
                assert np.abs(out["y"] - want).max() <= tol * scale * np.abs(x).max() * 65, (family, nv, mode)
            # residual: b - A x ; identity rows b - x ; zero rows 0.  Inexact dictionaries must NOT serve it
            out = ctx.kernel_apply(space, nv, x, a=a, b_coef=bc, family=family, epilogue=1, b=b, mask=mask, maskmode=1)
            if family == DICT and not exact:
                assert out["used_family"] != DICT, "a residual ran on a dictionary that is not bitwise exact"
            want = np.where(mask == 1, b - x, b - ref)
            assert np.abs(out["y"] - want).max() <= 1e-13 * scale * np.abs(x).max() * 65
        # a dictionary that is not bitwise exact serves no product unless the caller allows it (dict_ok)
        out = ctx.kernel_apply(space, nv, x, a=a, b_coef=bc, family=0, epilogue=0, dict_ok=False)
        assert (out["used_family"] == DICT) == exact
    ctx.close()


@pytest.mark.parametrize("dim,n,order", [(3, 16, "parity")])
def test_sell_kernel_matches_the_oracle(dim, n, order):
    """SELL-64 (tetrahedral P2 operators, >= 20 entries per row, parity-class numbering: consecutive rows of equal
    length): products, residuals and a smoothing step against the oracle"""
    mesh, dm, s = _lattice_case(dim, n, order)
    ctx = context(mesh, dm)
    A = (0.9 * s.mass_p2() + 0.4 * s.stiffness_p2()).tocsr()
    nrows = A.shape[0]
    rng = np.random.default_rng(3)
    for nv in (1, 2, 3):
        x, b, d, mask = _random_case(rng, nrows, nv)
        ref = (A @ x.reshape(nrows, nv)).ravel()
        out = ctx.kernel_apply(0, nv, x, a=0.9, b_coef=0.4, family=SELL, epilogue=0)
        assert out["used_family"] == SELL
        assert rel(out["y"], ref) < 1e-13
        out = ctx.kernel_apply(0, nv, x, a=0.9, b_coef=0.4, family=SELL, epilogue=1, b=b, mask=mask, maskmode=2)
        assert np.abs(out["y"] - np.where(mask == 1, 0.0, b - ref)).max() < 1e-12 * np.abs(ref).max()
        for c1 in (0.0, 0.37):
            xr, dr = _cheb_reference(A, nv, x, b, d, mask, [c1], [0.61])
            out = ctx.kernel_apply(0, nv, x, a=0.9, b_coef=0.4, family=SELL, epilogue=3, steps=1, b=b, d=d, mask=mask,
                                   maskmode=2, c1=[c1], c2=[0.61])
            assert out["used_family"] == SELL
            assert rel(out["y"], xr) < 1e-13 and rel(out["d"], dr) < 1e-13
    ctx.close()


@pytest.mark.parametrize("dim,n,exact", [(2, 64, True), (2, 48, False), (3, 16, True)])
@pytest.mark.parametrize("space", [0, 1])
def test_smoothing_steps_of_the_dictionary_kernels_match_the_oracle(dim, n, exact, space):
    """one Chebyshev-Jacobi step per launch (k_spmv_dict_w8<1|2,3>, k_spmv_dict<3,3>): first step (c1 = 0), later
    steps, zero-row masks, identity rows on the last step, frozen / zeroed ghost rows"""
    mesh, dm, s = _lattice_case(dim, n)
    ctx = context(mesh, dm)
    a, bc = 1500.0, 0.01          # the mass-dominated combination alpha0 / k M + nu K of a time step
    A = (a * (s.mass_p2() if space == 0 else s.mass_p1()) + bc * (s.stiffness_p2() if space == 0 else s.stiffness_p1())).tocsr()
    nrows = A.shape[0]
    rng = np.random.default_rng(21)
    tol = 1e-13 if exact else 2.0 ** -36
    for nv in (1, 2, 3):
        x, b, d, mask = _random_case(rng, nrows, nv)
        for c1s, c2s in (([0.0], [0.7]), ([0.25], [0.55]), ([0.0, 0.3, 0.2], [0.7, 0.6, 0.5])):
            for ident in (False, True):
                xr, dr = _cheb_reference(A, nv, x, b, d, mask, c1s, c2s, ident=ident)
                out = ctx.kernel_apply(space, nv, x, a=a, b_coef=bc, family=DICT, epilogue=3, steps=len(c1s), b=b, d=d,
                                       mask=mask, maskmode=2, c1=c1s, c2=c2s, ident=ident)
                assert out["used_family"] == DICT
                assert rel(out["y"], xr) < tol and rel(out["d"], dr) < tol, (nv, c1s, ident)
        # ghost rows (flag 2): ghost = 0 writes 0, ghost = 1 carries x through (frozen ghosts of the relaxed mode)
        gmask = mask.copy()
        gmask[rng.random(gmask.size) < 0.05] = 2
        xr, dr = _cheb_reference(A, nv, x, b, d, (gmask == 1).astype(np.uint8), [0.2], [0.6])
        for ghost in (0, 1):
            out = ctx.kernel_apply(space, nv, x, a=a, b_coef=bc, family=DICT, epilogue=3, steps=1, b=b, d=d, mask=gmask,
                                   maskmode=2, c1=[0.2], c2=[0.6], ghost=ghost)
            want = np.where(gmask == 2, x if ghost == 1 else 0.0, xr)
            assert rel(out["y"], want) < tol
            assert rel(out["d"], np.where(gmask == 2, 0.0, dr)) < tol
    ctx.close()


@pytest.mark.parametrize("n,exact", [(64, True), (48, False), (32, True)])
@pytest.mark.parametrize("space", [0, 1])
def test_multistep_lattice_kernel_matches_the_oracle(n, exact, space):
    """k_cheb_lattice: 1 - 4 Chebyshev-Jacobi steps of a 2D lattice operator in ONE launch (iterate staged in
    LDS, halo shrinking by the stencil reach per step) against the same steps computed with the oracle's matrix:
    from a given iterate and from zero, with a carried direction, with the fused residual, masks, identity
    rows; and against the one-step dictionary kernel (same arithmetic order: equal to round-off)."""
    mesh, dm, s = _lattice_case(2, n)
    ctx = context(mesh, dm)
    a, bc = 1500.0, 0.01
    A = (a * (s.mass_p2() if space == 0 else s.mass_p1()) + bc * (s.stiffness_p2() if space == 0 else s.stiffness_p1())).tocsr()
    nrows = A.shape[0]
    rng = np.random.default_rng(5)
    tol = 1e-13 if exact else 2.0 ** -36
    reach = 2 if space == 0 else 1
    for nv in (1, 2):
        x, b, d, mask = _random_case(rng, nrows, nv)
        c1_all, c2_all = [0.0, 0.31, 0.22, 0.17], [0.72, 0.63, 0.54, 0.45]
        for steps in (1, 2, 3, 4):
            for from_zero in (False, True):
                for with_res in (False, True):
                    mv = steps - (1 if from_zero else 0) + (1 if with_res else 0)
                    if mv > (3 if reach == 2 else 4) or steps > 4:
                        continue
                    for ident in (False, True):
                        c1s, c2s = c1_all[:steps], c2_all[:steps]
                        xr, dr = _cheb_reference(A, nv, x, b, None, mask, c1s, c2s, ident=ident, from_zero=from_zero)
                        out = ctx.kernel_apply(space, nv, x, a=a, b_coef=bc, family=LATTICE, epilogue=3, steps=steps,
                                               b=b, mask=mask, maskmode=2, c1=c1s, c2=c2s, ident=ident,
                                               from_zero=from_zero, with_residual=with_res)
                        assert out["used_family"] == LATTICE and out["lattice_w"] == (2 * n + 1 if space == 0 else n + 1)
                        key = (nv, steps, from_zero, with_res, ident)
                        assert rel(out["y"], xr) < tol, key
                        assert rel(out["d"], dr) < tol, key
                        if with_res:
                            rr = np.where(mask == 1, 0.0, b - (A @ xr.reshape(nrows, nv)).ravel())
                            assert np.abs(out["r"] - rr).max() < max(tol, 1e-12) * np.abs(b).max() * 30, key
        # carried direction (second launch of a long sequence): d_in given, c1 != 0 in the first step
        c1s, c2s = [0.3, 0.2, 0.1], [0.6, 0.5, 0.4]
        xr, dr = _cheb_reference(A, nv, x, b, d, mask, c1s, c2s)
        out = ctx.kernel_apply(space, nv, x, a=a, b_coef=bc, family=LATTICE, epilogue=3, steps=3, b=b, d=d, mask=mask,
                               maskmode=2, c1=c1s, c2=c2s)
        assert rel(out["y"], xr) < tol and rel(out["d"], dr) < tol
        # the one-step dictionary kernel on the same sequence
        one = ctx.kernel_apply(space, nv, x, a=a, b_coef=bc, family=DICT, epilogue=3, steps=3, b=b, d=d, mask=mask,
                               maskmode=2, c1=c1s, c2=c2s)
        assert rel(out["y"], one["y"]) < 1e-14 and rel(out["d"], one["d"]) < 1e-13
    ctx.close()


def test_unstructured_mesh_has_no_dictionary_and_no_lattice():
    import grid_generator as gg
    from fem_mesh import TaylorHoodDofMap
    mesh, _ = gg.dfg_channel(4, 2)
    dm = TaylorHoodDofMap(mesh)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    ctx = context(mesh, dm)
    x = np.random.default_rng(2).standard_normal(dm.n_p2 * 2)
    out = ctx.kernel_apply(0, 2, x, a=1.0, b_coef=0.5, family=0, epilogue=0)
    assert out["dict_entries"] == 0 and out["lattice_w"] == 0 and out["used_family"] == CSR
    A = (s.mass_p2() + 0.5 * s.stiffness_p2()).tocsr()
    assert rel(out["y"], (A @ x.reshape(-1, 2)).ravel()) < 1e-13
    with pytest.raises(nat.NativeError):
        ctx.kernel_apply(0, 2, x, family=LATTICE, epilogue=3, steps=1, b=x, c1=[0.0], c2=[0.5])
    ctx.close()


@pytest.mark.parametrize("n", [64, 48])
def test_cycles_on_the_lattice_kernel_equal_the_one_step_cycles(n, monkeypatch):
    """IPCS steps of the cavity with multigrid preconditioning three ways: one-step kernels only (NSFEM_LATTICE=0),
    the multi-step lattice kernel with explicit transfer launches (NSFEM_LATTICE_TRANSFERS=0), and with the
    prolongations / restrictions applied inside its staging (default).  Same arithmetic up to summation order:
    identical Newton counts, Krylov counts within one, fields equal far below the solver tolerance; and the
    converged fields against the LU oracle."""
    from gpu_common import box, cavity_bc
    from multigrid import attach_hierarchy
    mesh, dm, marks = box(n, n)
    mesh.structured = ((0.0, 0.0), (1.0, 1.0), n, n)
    bd, bv = cavity_bc(dm, marks)
    out = {}
    for tag, env in (("one-step", {"NSFEM_LATTICE": "0"}), ("lattice", {"NSFEM_LATTICE_TRANSFERS": "0"}), ("fused", {})):
        for k in ("NSFEM_LATTICE", "NSFEM_LATTICE_TRANSFERS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = context(mesh, dm)
        attach_hierarchy(ctx, mesh, coarsest=4)
        ctx.set_coeffs(1.0, 1.0, 0.01)
        ctx.set_dirichlet(nat.VELOCITY, bd.astype(np.int32), bv)
        ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
        ctx.mg_set_truncation(0.0, 0.1)            # full cycles: every level, every transfer
        opts = ctx.default_step_opts()
        for o in (opts.momentum, opts.poisson, opts.correction):
            o.rtol = 1e-12
        opts.momentum.precond = opts.poisson.precond = 1
        opts.correction.precond = 2
        its = []
        for step in range(3):
            ctx.set_bdf((1.0, -1.0, 0.0) if step == 0 else (1.5, -2.0, 0.5), 0.01)
            info = ctx.step_ipcs(opts)
            ctx.advance(0)
            its.append((info.newton_iterations, info.krylov_iterations_momentum, info.krylov_iterations_poisson))
        out[tag] = (ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD), its, ctx.smoother_info())
        ctx.close()
    assert not out["one-step"][3]["multistep_lattice_kernel"] and out["fused"][3]["multistep_lattice_kernel"]
    u0, p0, its0, _ = out["one-step"]
    for tag in ("lattice", "fused"):
        u, p, its, _ = out[tag]
        for a, b in zip(its0, its):
            assert a[0] == b[0] and abs(a[1] - b[1]) <= 1 and abs(a[2] - b[2]) <= 1, (tag, its0, its)
        assert rel(u, u0) < 1e-10 and rel(p - p.mean(), p0 - p0.mean()) < 1e-9, tag
    if n == 48:
        return
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    orc = fo.IPCSOracle(s, dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.01, body_force_term=None),
                        refactor_every_step=False)
    for step in range(3):
        orc.step(fo.bdf_alpha(step, 1.0), 0.01, (bd, bv))
        orc.advance()
    u, p = out["fused"][0], out["fused"][1]
    assert rel(u, orc.vel[1]) < 1e-8
    assert rel(p - p.mean(), orc.p_old - orc.p_old.mean()) < 1e-7


@pytest.mark.parametrize("nx,ny,form_id,form,picard", [
    (16, 16, 0, "standard", False), (80, 24, 0, "standard", False), (36, 52, 1, "rotational", False),
    (40, 40, 2, "divergence", False), (64, 32, 3, "skew_symmetric", False), (48, 20, 0, "standard", True)])
def test_lattice_jacobian_kernel_equals_the_launch_pair_bitwise_and_the_oracle(nx, ny, form_id, form, picard,
                                                                               monkeypatch):
    """Matrix-free action of the velocity Jacobian  y = L x + c_c [d conv(u)/du] x  (identity on Dirichlet rows) on
    lattice meshes: the one-launch kernel k_jac_lattice against the pair it replaces (k_conv_cell + dictionary product
    with fused node gather; NSFEM_JAC_LATTICE=0) bit for bit -- same products in the same order -- and both against
    the oracle's exact Gateaux derivative (source/ns_solver_base.py:370-390 linearised).  Tile edges: 1 x 3 up to
    3 x 8 workgroup tiles, partial tiles in both directions."""
    from gpu_common import box, cavity_bc
    mesh, dm, marks = box(nx, ny, p1=(nx / 16.0, ny / 16.0))          # binary spacing: exact dictionary
    bd, bv = cavity_bc(dm, marks)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    rng = np.random.default_rng(100 * nx + ny)
    u = rng.standard_normal(dm.n_velocity)
    x = rng.standard_normal(dm.n_velocity)
    out = {}
    for tag, env in (("pair", "0"), ("lattice", "1")):
        monkeypatch.setenv("NSFEM_JAC_LATTICE", env)
        ctx = context(mesh, dm)
        ctx.set_coeffs(0.8, 1.0, 0.02)
        ctx.set_bdf((1.5, -2.0, 0.5), 0.05)
        ctx.set_dirichlet(nat.VELOCITY, bd.astype(np.int32), bv)
        ctx.set_state(nat.USTAR, u)
        ctx.set_convective_form(form_id, picard=picard)
        info0 = ctx.jacobian_info()
        y = ctx.operator_apply(nat.OP_MOMENTUM_JAC_MF, x)
        info = ctx.jacobian_info()
        assert info0["path"] == info["path"] == ("lattice-kernel" if tag == "lattice" else "fused-gather")
        assert info["lattice_launches"] == (1 if tag == "lattice" else 0)
        out[tag] = y
        ctx.close()
    assert np.array_equal(out["pair"], out["lattice"])
    L = 1.5 / 0.05 * s.vector_mass() + 0.02 * s.vector_stiffness()
    J = L + 0.8 * (s.picard_convection(u, form) if picard else s.convection_jacobian(u, form))
    ref = J @ x
    ref[bd] = x[bd]
    assert rel(out["lattice"], ref) < 1e-13


@pytest.mark.parametrize("nx,ny,form_id,form", [(16, 16, 0, "standard"), (80, 24, 3, "skew_symmetric"),
                                                 (36, 52, 1, "rotational"), (40, 40, 2, "divergence")])
def test_lattice_residual_kernel_equals_the_four_launch_path_bitwise_and_the_oracle(nx, ny, form_id, form, monkeypatch):
    """Momentum residual  b = L u* + g + c_c conv(u*)  (Dirichlet rows u* - g_D; source/ns_ipcs_solver.py:126-135) on
    lattice meshes: k_jac_lattice<FORM, 0> in one launch against product + axpby + k_conv_cell + k_res_gather
    (NSFEM_JAC_LATTICE=0) bit for bit, and against the oracle's residual."""
    from gpu_common import box, cavity_bc
    mesh, dm, marks = box(nx, ny, p1=(nx / 16.0, ny / 16.0))
    bd, bv = cavity_bc(dm, marks)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    rng = np.random.default_rng(7 * nx + ny)
    u, u1, u2 = (rng.standard_normal(dm.n_velocity) for _ in range(3))
    p = rng.standard_normal(dm.n_p1)
    out = {}
    for tag, env in (("four", "0"), ("lattice", "1")):
        monkeypatch.setenv("NSFEM_JAC_LATTICE", env)
        ctx = context(mesh, dm)
        ctx.set_coeffs(0.8, 1.0, 0.02)
        ctx.set_bdf((1.5, -2.0, 0.5), 0.05)
        ctx.set_dirichlet(nat.VELOCITY, bd.astype(np.int32), bv)
        for slot, v in ((nat.U1, u1), (nat.U2, u2), (nat.USTAR, u), (nat.P_OLD, p)):
            ctx.set_state(slot, v)
        ctx.set_convective_form(form_id)
        n0 = ctx.jacobian_info()["lattice_launches"]
        ctx.assemble(nat.SYS_MOMENTUM, new_step=True)
        info = ctx.jacobian_info()
        assert info["path"] == ("lattice-kernel" if tag == "lattice" else "fused-gather")
        assert info["lattice_launches"] - n0 == (1 if tag == "lattice" else 0)
        out[tag] = ctx.get_rhs(nat.SYS_MOMENTUM)
        ctx.close()
    assert np.array_equal(out["four"], out["lattice"])
    M = s.vector_mass()
    b = (1.5 / 0.05 * M + 0.02 * s.vector_stiffness()) @ u + M @ ((-2.0 * u1 + 0.5 * u2) / 0.05) \
        - 1.0 * (s.divergence().T @ p) + 0.8 * s.convection_residual(u, form)      # (fem_oracle.IPCSOracle: const)
    b[bd] = u[bd] - bv
    assert rel(out["lattice"], b) < 1e-12


def test_lattice_jacobian_kernel_on_an_inexact_dictionary_serves_the_action_but_not_the_residual(monkeypatch):
    """Spacing 1/48 (not a binary fraction): the stencil dictionary equals the assembled matrix only to 2^-40 of its
    largest entry.  The Jacobian ACTION may use it (as the launch pair does: same bits from both), the Newton RESIDUAL
    must come from the assembled values: k_jac_lattice<FORM, 0> is not launched and the right-hand side equals the
    four-launch path bit for bit."""
    from gpu_common import box, cavity_bc
    mesh, dm, marks = box(48, 48)
    bd, bv = cavity_bc(dm, marks)
    rng = np.random.default_rng(48)
    u, x = rng.standard_normal(dm.n_velocity), rng.standard_normal(dm.n_velocity)
    out = {}
    for tag, env in (("pair", "0"), ("lattice", "1")):
        monkeypatch.setenv("NSFEM_JAC_LATTICE", env)
        ctx = context(mesh, dm)
        ctx.set_coeffs(1.0, 1.0, 0.01)
        ctx.set_bdf((1.5, -2.0, 0.5), 0.01)
        ctx.set_dirichlet(nat.VELOCITY, bd.astype(np.int32), bv)
        ctx.set_state(nat.USTAR, u)
        ctx.set_state(nat.U1, u)
        ctx.assemble(nat.SYS_MOMENTUM, new_step=True)
        n_res = ctx.jacobian_info()["lattice_launches"]
        y = ctx.operator_apply(nat.OP_MOMENTUM_JAC_MF, x)
        info = ctx.jacobian_info()
        assert n_res == 0                                             # no residual through the dictionary
        assert info["lattice_launches"] == (1 if tag == "lattice" else 0)
        out[tag] = (y, ctx.get_rhs(nat.SYS_MOMENTUM))
        ctx.close()
    assert np.array_equal(out["pair"][0], out["lattice"][0])
    assert np.array_equal(out["pair"][1], out["lattice"][1])


def test_full_size_properties_of_the_one_launch_jacobian_and_residual(monkeypatch):
    """BASELINE configs[1] at full size (512 x 512, 2.1 M velocity dofs), where the oracle cannot follow: the
    one-launch kernel equals the launch pair bit for bit (action and residual), the action is linear to round-off,
    and it is the Gateaux derivative of the residual it shares its element kernel with:
    (F(u + eps x) - F(u)) / eps -> J(u) x."""
    from gpu_common import box, cavity_bc
    n = 512
    mesh, dm, marks = box(n, n)
    bd, bv = cavity_bc(dm, marks)
    rng = np.random.default_rng(512)
    X = dm.p2_coords
    u = np.stack([np.sin(3 * X[:, 0]) * np.cos(2 * X[:, 1]), np.cos(X[:, 0]) * np.sin(4 * X[:, 1])], axis=1).ravel()
    x = rng.standard_normal(dm.n_velocity)
    y = rng.standard_normal(dm.n_velocity)
    free = np.ones(dm.n_velocity, dtype=bool)
    free[bd] = False
    res = {}
    for tag, env in (("pair", "0"), ("lattice", "1")):
        monkeypatch.setenv("NSFEM_JAC_LATTICE", env)
        ctx = context(mesh, dm)
        ctx.set_coeffs(1.0, 1.0, 0.01)
        ctx.set_bdf((1.5, -2.0, 0.5), 1e-3)
        ctx.set_dirichlet(nat.VELOCITY, bd.astype(np.int32), bv)
        ctx.set_state(nat.USTAR, u)
        ctx.assemble(nat.SYS_MOMENTUM, new_step=True)
        out = [ctx.get_rhs(nat.SYS_MOMENTUM), ctx.operator_apply(nat.OP_MOMENTUM_JAC_MF, x)]
        if tag == "lattice":
            assert ctx.jacobian_info()["path"] == "lattice-kernel" and ctx.jacobian_info()["lattice_launches"] == 2
            jy = ctx.operator_apply(nat.OP_MOMENTUM_JAC_MF, y)
            jc = ctx.operator_apply(nat.OP_MOMENTUM_JAC_MF, 0.75 * x - 1.5 * y)
            assert rel(jc, 0.75 * out[1] - 1.5 * jy) < 1e-14
            xs = np.stack([np.cos(2 * X[:, 0] + X[:, 1]), np.sin(X[:, 0] - 3 * X[:, 1])], axis=1).ravel()   # smooth direction
            js = ctx.operator_apply(nat.OP_MOMENTUM_JAC_MF, xs)
            eps = 1e-6
            ctx.set_state(nat.USTAR, u + eps * xs)
            ctx.assemble(nat.SYS_MOMENTUM)
            fd = (ctx.get_rhs(nat.SYS_MOMENTUM) - out[0]) / eps
            assert rel(fd[free], js[free]) < 1e-4
        res[tag] = out
        ctx.close()
    assert np.array_equal(res["pair"][0], res["lattice"][0])
    assert np.array_equal(res["pair"][1], res["lattice"][1])


@pytest.mark.parametrize("n", [50, 27])
def test_hierarchies_continue_below_odd_levels_with_non_nested_meshes(n):
    """A structured mesh whose cell count becomes odd on the way down (50 -> 25 -> 13 -> 7 -> 4; 27 -> 14 -> 7 -> 4; BASELINE's
    333 x 333) used to end its hierarchy there -- a two-level method with a huge smoothed coarsest level.  Now the next
    level is the NON-NESTED mesh of ceil(n / 2) cells with linear interpolation between the meshes
    (multigrid.interpolation_prolongation); the Dirichlet rows go down through the fine node a coarse hat function weighs
    most.  IPCS steps with multigrid-preconditioned BiCGStab / CG: converged fields against the LU oracle and iteration
    counts of a working multigrid."""
    from gpu_common import box, cavity_bc
    from multigrid import attach_hierarchy, structured_hierarchy
    mesh, dm, marks = box(n, n)
    mesh.structured = ((0.0, 0.0), (1.0, 1.0), n, n)
    sizes = [m.structured[2] for m, _ in structured_hierarchy(*mesh.structured, coarsest=4)]
    assert sizes == ([25, 13, 7, 4] if n == 50 else [14, 7, 4])
    bd, bv = cavity_bc(dm, marks)
    ctx = context(mesh, dm)
    assert attach_hierarchy(ctx, mesh, coarsest=4) == len(sizes)
    ctx.set_coeffs(1.0, 1.0, 0.01)
    ctx.set_dirichlet(nat.VELOCITY, bd.astype(np.int32), bv)
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    ctx.mg_set_truncation(0.0, 0.1)            # full cycles: every level, every transfer
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = 1e-12
    opts.momentum.precond = opts.poisson.precond = 1
    opts.correction.precond = 2
    its = []
    for step in range(3):
        ctx.set_bdf((1.0, -1.0, 0.0) if step == 0 else (1.5, -2.0, 0.5), 0.01)
        info = ctx.step_ipcs(opts)
        ctx.advance(0)
        its.append((info.newton_iterations, info.krylov_iterations_momentum, info.krylov_iterations_poisson))
    u, p = ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD)
    ctx.close()
    assert all(k[1] <= 12 * k[0] and k[2] <= 16 for k in its), its      # (a two-level method needs 60+ CG iterations)
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    orc = fo.IPCSOracle(s, dict(convective_term=1.0, pressure_term=1.0, viscous_term=0.01, body_force_term=None),
                        refactor_every_step=False)
    for step in range(3):
        orc.step(fo.bdf_alpha(step, 1.0), 0.01, (bd, bv))
        orc.advance()
    assert rel(u, orc.vel[1]) < 1e-8
    assert rel(p - p.mean(), orc.p_old - orc.p_old.mean()) < 1e-7
