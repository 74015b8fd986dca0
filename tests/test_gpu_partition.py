"""Partitioned (multi-GPU) path on ONE GPU: several contexts in one process, one host thread per
rank, coupled by the in-process communicator that implements the same interface as the RCCL
one (csrc/comm.hip).  The strip-partitioned IPCS step -- halo exchange before every SpMV,
all-reduced partial dot products, partitioned multigrid with a replicated global coarse solve
-- must reproduce the single-context run: same Newton/Krylov iteration counts, fields equal to
round-off, ghost copies consistent with their owners.  A second test drives bench.py through
RCCL itself with one rank (NSFEM_FORCE_COMM routes the single-rank run through ncclAllReduce)."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import _native as nat
from gpu_common import box, context, rel
from multigrid import attach_hierarchy
from partition import StripPartition

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cavity_bc(dm, height=1.0):
    X = dm.p2_coords
    on = (np.abs(X[:, 0]) < 1e-12) | (np.abs(X[:, 0] - 1) < 1e-12) | (np.abs(X[:, 1]) < 1e-12) | \
        (np.abs(X[:, 1] - height) < 1e-12)
    nodes = np.nonzero(on)[0]
    lid = np.abs(X[nodes, 1] - height) < 1e-12
    return (np.concatenate([2 * nodes, 2 * nodes + 1]).astype(np.int32),
            np.concatenate([np.where(lid, 1.0, 0.0), np.zeros(nodes.size)]))


def _run(ctx, dm, nsteps, k, use_mg, out, key, cheb=False):
    ctx.set_coeffs(1.0, 1.0, 0.01)
    ctx.set_dirichlet(nat.VELOCITY, *_cavity_bc(dm))
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = 1e-12
    if use_mg:
        opts.momentum.precond = opts.poisson.precond = 1
    if cheb:
        opts.correction.precond = 2
    infos = []
    for step in range(nsteps):
        ctx.set_bdf((1.0, -1.0, 0.0) if step == 0 else (1.5, -2.0, 0.5), k)
        infos.append(ctx.step_ipcs(opts))
        ctx.advance(0)
    out[key] = (ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD), infos)
    out[("comm", key)] = ctx.comm_stats()


@pytest.mark.parametrize("n,size,relaxed", [(32, 2, False), (64, 4, True), (48, 3, False)])
def test_partitioned_fast_diagonalisation_projection_equals_single_context(n, size, relaxed):
    """Projection step by fast diagonalisation on strips (nsfem_poisson_set_fast_diag_rows, csrc/fastdiag.hip): every
    rank keeps the rows of V_y of its own lattice lines, the contraction over y is ONE all-reduce of (n + 1)^2 doubles;
    the pressure comes back with valid ghost rows.  Same fields as the single context running the same direct solve
    (source/ns_ipcs_solver.py:160-171: the reference's LU), one pass per step, and the projection step costs no halo
    exchange at all: the exchange count of a step drops by what the multigrid-CG solve needed."""
    import poisson_fd as pf
    nsteps, k = 3, 0.01
    mesh, dm, _ = box(n, n)
    mesh.structured = ((0.0, 0.0), (1.0, 1.0), n, n)
    xs = np.linspace(0.0, 1.0, n + 1)
    factors = pf.factors(xs, xs, np.zeros(0, np.int64))

    def run(ctx, d, out, key, fd):
        ctx.set_coeffs(1.0, 1.0, 0.01)
        ctx.set_dirichlet(nat.VELOCITY, *_cavity_bc(d))
        ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
        opts = ctx.default_step_opts()
        for o in (opts.momentum, opts.poisson, opts.correction):
            o.rtol = 1e-12
        opts.momentum.precond = opts.poisson.precond = 1
        if fd:
            opts.poisson.precond = 3
        opts.correction.precond = 2
        infos = []
        for step in range(nsteps):
            ctx.set_bdf((1.0, -1.0, 0.0) if step == 0 else (1.5, -2.0, 0.5), k)
            infos.append(ctx.step_ipcs(opts))
            ctx.advance(0)
        out[key] = (ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD), infos, ctx.comm_stats())

    ref = {}
    ctx0 = context(mesh, dm)
    attach_hierarchy(ctx0, mesh, coarsest=2)
    ctx0.poisson_set_fast_diag(factors)
    run(ctx0, dm, ref, 0, True)
    u_ref, p_ref, inf_ref, _ = ref[0]
    ctx0.close()

    results = {}
    for fd in (False, True):
        group = nat.local_group_create(size)
        parts = [StripPartition((0.0, 0.0), (1.0, 1.0), n, n, r, size, coarsest=2) for r in range(size)]
        ctxs = []
        for r, part in enumerate(parts):
            pdm = part.dofmap
            c = nat.NsfemContext(part.mesh.coords, part.mesh.cells, pdm.p2_dofmap, pdm.p1_dofmap, pdm.n_p2, pdm.n_p1)
            c.attach_local_comm(group, r)
            ctxs.append(c)
        out, errors = {}, []

        def worker(r):
            try:
                part = parts[r]
                part.attach(ctxs[r])
                ctxs[r].mg_set_halo_mode(relaxed)
                if fd:
                    first = int(part.p1_global[0]) // (n + 1)
                    assert part.dofmap.n_p1 % (n + 1) == 0
                    ctxs[r].poisson_set_fast_diag(factors, first_line=first)
                run(ctxs[r], part.dofmap, out, r, fd)
            except BaseException as exc:                     # a dead rank would deadlock the others
                errors.append((r, repr(exc)))
                os._exit(17)

        threads = [threading.Thread(target=worker, args=(r,)) for r in range(size)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors
        results[fd] = (parts, out)
        for c in ctxs:
            c.close()
        nat.local_group_destroy(group)

    parts, out = results[True]
    u = np.zeros_like(u_ref)
    p = np.zeros_like(p_ref)
    for r, part in enumerate(parts):
        ul, pl, infos, _ = out[r]
        u.reshape(-1, 2)[part.p2_global[part.p2_owned]] = ul.reshape(-1, 2)[part.p2_owned]
        p[part.p1_global[part.p1_owned]] = pl[part.p1_owned]
        for a, b in zip(infos, inf_ref):
            assert a.newton_iterations == b.newton_iterations
            assert a.krylov_iterations_poisson == b.krylov_iterations_poisson == 1
    assert rel(u, u_ref) < (1e-9 if relaxed else 1e-11)
    assert rel(p - p.mean(), p_ref - p_ref.mean()) < (1e-8 if relaxed else 1e-10)
    for r, part in enumerate(parts):                         # ghost rows of the pressure: copies of the owners' values
        pl = out[r][1]
        assert np.abs(pl - p[part.p1_global]).max() < 1e-12 * max(1.0, np.abs(p).max())
    ex_fd = results[True][1][0][3]["exchanges"]
    ex_mg = results[False][1][0][3]["exchanges"]
    ar_fd = results[True][1][0][3]["allreduce_calls"]
    ar_mg = results[False][1][0][3]["allreduce_calls"]
    print("\n[n = %d, %d ranks] exchanges per step %.1f -> %.1f, all-reduces %.1f -> %.1f" % (
        n, size, ex_mg / nsteps, ex_fd / nsteps, ar_mg / nsteps, ar_fd / nsteps))
    assert ex_fd < 0.8 * ex_mg and ar_fd < ar_mg


@pytest.mark.parametrize("n,size,use_mg,tail,cheb,relaxed,overlap", [
    (16, 2, False, False, False, False, False), (32, 4, True, False, False, False, False),
    (64, 2, True, False, False, False, False), (64, 2, True, True, False, False, False),
    (32, 2, True, False, True, False, False), (64, 4, True, False, True, True, False),
    (64, 2, True, True, False, True, False),
    (64, 2, False, False, False, False, True), (64, 2, True, False, True, False, True),
    (64, 4, True, True, True, True, True)])
def test_partitioned_ipcs_equals_single_context(n, size, use_mg, tail, cheb, relaxed, overlap):
    """tail: the partitioned levels stop at 16 cells across and the rest of the hierarchy
    (16 -> 8 -> 4 -> 2) is the replicated global one -- still the serial algorithm.
    cheb: the velocity correction uses the dot-product-free Chebyshev mass solve.
    relaxed: nsfem_mg_set_halo_mode(1) -- frozen ghost values inside the smoothing sequences: a
    different (block-Jacobi across ranks) but equally good preconditioner, so the converged fields
    agree to solver tolerance, the iteration counts to within a few, and the number of halo
    exchanges drops by about a third.
    overlap: nsfem_set_overlap(1) -- the halo exchanges of the Krylov operators and smoothing steps
    run on the communicator's own stream under the row blocks that touch no ghost column, the
    halo-adjacent row blocks follow after an event wait: same arithmetic, same results."""
    nsteps, k, coarsest = 3, 0.01, 2
    mesh, dm, _ = box(n, n)
    ref = {}
    ctx0 = context(mesh, dm)
    if use_mg:
        attach_hierarchy(ctx0, mesh, coarsest=coarsest)
    _run(ctx0, dm, nsteps, k, use_mg, ref, 0, cheb)
    u_ref, p_ref, inf_ref = ref[0]
    ctx0.close()

    group = nat.local_group_create(size)
    parts = [StripPartition((0.0, 0.0), (1.0, 1.0), n, n, r, size, coarsest=16 if tail else coarsest,
                            global_coarsest=coarsest if tail else None)
             for r in range(size)]
    assert bool(parts[0].global_tail) == tail
    ctxs = []
    for r, part in enumerate(parts):
        pdm = part.dofmap
        c = nat.NsfemContext(part.mesh.coords, part.mesh.cells, pdm.p2_dofmap, pdm.p1_dofmap,
                             pdm.n_p2, pdm.n_p1)
        c.attach_local_comm(group, r)
        c.set_overlap(overlap)
        ctxs.append(c)
    out, errors = {}, []

    def worker(r):
        try:
            part = parts[r]
            if use_mg:
                part.attach(ctxs[r])
                ctxs[r].mg_set_halo_mode(relaxed)
            else:
                ctxs[r].set_partition(r, size, part.p2_ghost, part.p1_ghost, part.p2_halo,
                                      part.p1_halo, (2 * n + 1) ** 2, (n + 1) ** 2)
            _run(ctxs[r], part.dofmap, nsteps, k, use_mg, out, r, cheb)
            out[("overlapped", r)] = ctxs[r].comm_overlapped()
            if use_mg:
                out[("kernels", r)] = (ctxs[r].jacobian_info(), ctxs[r].mg_lattice_info(1), ctxs[r].mg_lattice_info(0))
        except BaseException as exc:                     # a dead rank would deadlock the others
            errors.append((r, repr(exc)))
            os._exit(17)

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    u = np.zeros_like(u_ref)
    p = np.zeros_like(p_ref)
    for r, part in enumerate(parts):
        ul, pl, infos = out[r]
        u.reshape(-1, 2)[part.p2_global[part.p2_owned]] = ul.reshape(-1, 2)[part.p2_owned]
        p[part.p1_global[part.p1_owned]] = pl[part.p1_owned]
        for a, b in zip(infos, inf_ref):                 # the partitioned algorithm IS the serial one
            assert a.newton_iterations == b.newton_iterations
            if relaxed:
                assert abs(a.krylov_iterations_momentum - b.krylov_iterations_momentum) <= 3
                assert abs(a.krylov_iterations_poisson - b.krylov_iterations_poisson) <= 3
                continue
            assert a.krylov_iterations_momentum == b.krylov_iterations_momentum
            assert a.krylov_iterations_poisson == b.krylov_iterations_poisson
            if cheb:        # a-priori bounds: the step count is predicted, identical everywhere
                assert a.krylov_iterations_correction == b.krylov_iterations_correction <= 60
    assert rel(u, u_ref) < (1e-9 if relaxed else 1e-11)
    assert rel(p - p.mean(), p_ref - p_ref.mean()) < (1e-8 if relaxed else 1e-10)
    for r, part in enumerate(parts):                     # ghosts are copies of the owners' values
        ul, _, _ = out[r]
        assert np.abs(ul.reshape(-1, 2) - u.reshape(-1, 2)[part.p2_global]).max() < 1e-13
    # communication counters (nsfem_comm_stats): none without a communicator; with one, every rank
    # exchanged halos and all-reduced dot products, interior ranks send to two neighbours
    assert ref[("comm", 0)] == dict(allreduce_calls=0, allreduce_bytes=0, exchanges=0, exchange_bytes=0)
    st = [out[("comm", r)] for r in range(size)]
    assert all(x["exchanges"] == st[0]["exchanges"] > 0 and x["allreduce_calls"] == st[0]["allreduce_calls"] > 0
               for x in st)
    if size > 2:
        assert st[1]["exchange_bytes"] > st[0]["exchange_bytes"]
    # overlapped exchanges: none unless enabled; when enabled, the bulk of them (every Krylov
    # operator application and smoothing step of a level that has interior row blocks)
    for r in range(size):
        if overlap:
            assert 0.3 * st[r]["exchanges"] < out[("overlapped", r)] <= st[r]["exchanges"]
        else:
            assert out[("overlapped", r)] == 0
    # every rank runs the single-GPU kernel set: the whole Jacobian action / momentum residual in
    # ONE launch of k_jac_lattice on its strip (exact: one halo exchange of the input first), and -- in relaxed halo
    # mode, bench.py's N > 1 default -- every smoothing sequence of the partitioned lattice levels in the multi-step
    # kernel k_cheb_lattice with the ghost lines frozen; exact halo mode keeps the one-step kernels
    if use_mg and n >= 32:
        for r in range(size):
            ji, lat_v, lat_p = out[("kernels", r)]
            assert ji["path"] == "lattice-kernel" and ji["lattice_launches"] > 0, (r, ji)
            if relaxed:
                assert lat_v["lattice_levels"] >= 1 and lat_v["lattice_launches"] > 0, (r, lat_v)
                assert lat_p["lattice_levels"] >= 1 and lat_p["lattice_launches"] > 0, (r, lat_p)
                assert lat_v["ghost_lines"] == ((0 if r == 0 else 1), (0 if r == size - 1 else 2)), (r, lat_v)
            else:
                assert lat_v["lattice_launches"] == 0 and lat_p["lattice_launches"] == 0
    for c in ctxs:
        c.close()
    nat.local_group_destroy(group)


@pytest.mark.parametrize("n,size,tail,relaxed", [(32, 2, False, False), (64, 4, True, False), (64, 4, True, True)])
def test_partitioned_monolithic_bdf_equals_single_context(n, size, tail, relaxed):
    """The monolithic BDF step on strips: halo-exchanged mixed operator (matrix-free velocity
    block), partitioned block preconditioner (Schur Laplacian and velocity V-cycles, pressure-mass
    smoother), all-reduced dots -- same Newton / BiCGStab counts and fields as one context."""
    nsteps, k, coarsest = 3, 0.01, 2
    mesh, dm, _ = box(n, n)

    def run(ctx, dmap, out, key):
        ctx.set_coeffs(1.0, 1.0, 0.01)
        ctx.set_dirichlet(nat.VELOCITY, *_cavity_bc(dmap))
        ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
        ctx.set_dirichlet(nat.PRESSURE_PRECOND, np.zeros(0, np.int32), np.zeros(0))
        opts = ctx.default_step_opts()
        opts.momentum.rtol, opts.momentum.precond = 1e-12, 1
        infos = []
        for step in range(nsteps):
            ctx.set_bdf((1.0, -1.0, 0.0) if step == 0 else (1.5, -2.0, 0.5), k)
            infos.append(ctx.step_bdf(opts))
            ctx.advance(1)
        out[key] = (ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD), infos)

    ref = {}
    ctx0 = context(mesh, dm)
    attach_hierarchy(ctx0, mesh, coarsest=coarsest)
    run(ctx0, dm, ref, 0)
    u_ref, p_ref, inf_ref = ref[0]
    ctx0.close()
    group = nat.local_group_create(size)
    parts = [StripPartition((0.0, 0.0), (1.0, 1.0), n, n, r, size, coarsest=16 if tail else coarsest,
                            global_coarsest=coarsest if tail else None) for r in range(size)]
    ctxs = []
    for r, part in enumerate(parts):
        pdm = part.dofmap
        c = nat.NsfemContext(part.mesh.coords, part.mesh.cells, pdm.p2_dofmap, pdm.p1_dofmap,
                             pdm.n_p2, pdm.n_p1)
        c.attach_local_comm(group, r)
        ctxs.append(c)
    out, errors = {}, []

    def worker(r):
        try:
            parts[r].attach(ctxs[r])
            ctxs[r].mg_set_halo_mode(relaxed)
            run(ctxs[r], parts[r].dofmap, out, r)
        except BaseException as exc:
            errors.append((r, repr(exc)))
            os._exit(17)

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    u = np.zeros_like(u_ref)
    p = np.zeros_like(p_ref)
    for r, part in enumerate(parts):
        ul, pl, infos = out[r]
        u.reshape(-1, 2)[part.p2_global[part.p2_owned]] = ul.reshape(-1, 2)[part.p2_owned]
        p[part.p1_global[part.p1_owned]] = pl[part.p1_owned]
        for a, b in zip(infos, inf_ref):
            assert a.newton_iterations == b.newton_iterations
            assert abs(a.krylov_iterations_momentum - b.krylov_iterations_momentum) <= (max(3, 0.2 * b.krylov_iterations_momentum) if relaxed else 1)
    assert rel(u, u_ref) < (1e-8 if relaxed else 1e-10)
    assert rel(p - p.mean(), p_ref - p_ref.mean()) < (1e-7 if relaxed else 1e-9)
    for c in ctxs:
        c.close()
    nat.local_group_destroy(group)


@pytest.mark.parametrize("scheme,problem", [("ipcs", "cavity"), ("bdf", "cavity"), ("ipcs", "channel"),
                                            ("bdf", "channel")])
def test_partitioned_3d_slabs_equal_single_context(scheme, problem):
    """3D: slabs of cube layers along z (SlabPartition), partitioned multigrid with the replicated
    global tail -- two in-process ranks reproduce the single-context 3D IPCS / monolithic runs.
    problem "channel" (BASELINE configs[4] in small): inflow at x = 0, open outlet at x = 1 that
    every slab touches -- pressure Dirichlet values there (IPCS) / natural outflow with the
    geometric Schur Laplacian pinned at the outlet (monolithic)."""
    from fem_mesh import TaylorHoodDofMap, box_mesh
    from partition import SlabPartition
    n, size, nsteps, k = 8, 2, 2, 0.02
    mesh = box_mesh((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), n, n, n)
    dm = TaylorHoodDofMap(mesh)

    def bc(dmap):
        X = dmap.p2_coords
        on = np.zeros(dmap.n_p2, dtype=bool)
        for a in range(3):
            on |= (np.abs(X[:, a]) < 1e-12)
            if not (problem == "channel" and a == 0):
                on |= (np.abs(X[:, a] - 1.0) < 1e-12)
        nodes = np.nonzero(on)[0]
        if problem == "channel":        # parabolic inflow at x = 0, walls elsewhere, x = 1 open
            inlet = (np.abs(X[nodes, 0]) < 1e-12)
            ux = np.where(inlet, 16.0 * X[nodes, 1] * (1 - X[nodes, 1]) * X[nodes, 2] * (1 - X[nodes, 2]), 0.0)
        else:
            ux = np.where(np.abs(X[nodes, 2] - 1.0) < 1e-12, 1.0, 0.0)
        return (np.concatenate([3 * nodes, 3 * nodes + 1, 3 * nodes + 2]).astype(np.int32),
                np.concatenate([ux, np.zeros(2 * nodes.size)]))

    def run(ctx, dmap, out, key):
        ctx.set_coeffs(1.0, 1.0, 0.02)
        ctx.set_dirichlet(nat.VELOCITY, *bc(dmap))
        outlet = np.nonzero(np.abs(dmap.p1_coords[:, 0] - 1.0) < 1e-12)[0].astype(np.int32) \
            if problem == "channel" else np.zeros(0, np.int32)
        ctx.set_dirichlet(nat.PRESSURE, outlet if scheme == "ipcs" else np.zeros(0, np.int32),
                          np.zeros(outlet.size if scheme == "ipcs" else 0))
        ctx.set_dirichlet(nat.PRESSURE_PRECOND, outlet, np.zeros(outlet.size))
        opts = ctx.default_step_opts()
        for o in (opts.momentum, opts.poisson, opts.correction):
            o.rtol = 1e-12
        opts.momentum.precond = opts.poisson.precond = 1
        infos = []
        for step in range(nsteps):
            ctx.set_bdf((1.0, -1.0, 0.0) if step == 0 else (1.5, -2.0, 0.5), k)
            infos.append(ctx.step_ipcs(opts) if scheme == "ipcs" else ctx.step_bdf(opts))
            ctx.advance(0 if scheme == "ipcs" else 1)
        out[key] = (ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD), infos)

    ref = {}
    ctx0 = context(mesh, dm)
    attach_hierarchy(ctx0, mesh, coarsest=2)
    run(ctx0, dm, ref, 0)
    u_ref, p_ref, inf_ref = ref[0]
    ctx0.close()
    group = nat.local_group_create(size)
    parts = [SlabPartition((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), n, n, n, r, size, coarsest=4,
                           global_coarsest=2) for r in range(size)]
    assert len(parts[0].levels) == 1 and len(parts[0].global_tail) == 1
    ctxs = []
    for r, part in enumerate(parts):
        pdm = part.dofmap
        c = nat.NsfemContext(part.mesh.coords, part.mesh.cells, pdm.p2_dofmap, pdm.p1_dofmap,
                             pdm.n_p2, pdm.n_p1)
        c.attach_local_comm(group, r)
        ctxs.append(c)
    out, errors = {}, []

    def worker(r):
        try:
            parts[r].attach(ctxs[r])
            run(ctxs[r], parts[r].dofmap, out, r)
        except BaseException as exc:
            errors.append((r, repr(exc)))
            os._exit(17)

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    u = np.zeros_like(u_ref)
    p = np.zeros_like(p_ref)
    for r, part in enumerate(parts):
        ul, pl, infos = out[r]
        u.reshape(-1, 3)[part.p2_global[part.p2_owned]] = ul.reshape(-1, 3)[part.p2_owned]
        p[part.p1_global[part.p1_owned]] = pl[part.p1_owned]
        for a, b in zip(infos, inf_ref):
            assert a.newton_iterations == b.newton_iterations
            assert abs(a.krylov_iterations_momentum - b.krylov_iterations_momentum) <= (0 if scheme == "ipcs" else 1)
            assert a.krylov_iterations_poisson == b.krylov_iterations_poisson
    assert rel(u, u_ref) < 1e-10
    if problem == "channel":            # the pressure level is fixed by the outlet
        assert rel(p, p_ref) < 1e-9
    else:
        assert rel(p - p.mean(), p_ref - p_ref.mean()) < 1e-9
    for c in ctxs:
        c.close()
    nat.local_group_destroy(group)


@pytest.mark.parametrize("dim,size", [(3, 2), (3, 4), (2, 2), (2, 4)])
def test_partitioned_algebraic_schur_laplacian_equals_single_context(dim, size, capsys):
    """BASELINE configs[4] in small on several ranks, with the preconditioner the single-GPU channel
    runs use: the ALGEBRAIC Schur Laplacian D diag(M)^-1 D^T (open outlet: no pressure Dirichlet
    set anywhere).  Every rank holds only its additive part (columns of the velocity dofs it owns)
    and its Galerkin coarsenings; products run as forward exchange -> local rows -> reverse (add)
    exchange.  Checks: (1) the parts sum to the single-context operator on every level it can be
    compared on, (2) the partitioned monolithic steps equal the single-context ones, with the same
    iteration counts; the halo / all-reduce counts per step are printed."""
    from fem_mesh import TaylorHoodDofMap, box_mesh, rectangle_mesh
    from multigrid import attach_schur_laplacian
    from partition import SlabPartition, StripPartition
    n, nsteps, k = (8, 2, 0.02) if dim == 3 else (32, 2, 0.02)
    if dim == 3:
        mesh = box_mesh((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), n, n, n)
    else:
        mesh = rectangle_mesh((0.0, 0.0), (1.0, 1.0), n, n)
    dm = TaylorHoodDofMap(mesh)

    def bc(dmap):
        X = dmap.p2_coords
        on = np.zeros(dmap.n_p2, dtype=bool)
        for a in range(dim):
            on |= (np.abs(X[:, a]) < 1e-12)
            if a != 0:
                on |= (np.abs(X[:, a] - 1.0) < 1e-12)          # x = 1 stays open
        nodes = np.nonzero(on)[0]
        inlet = (np.abs(X[nodes, 0]) < 1e-12)
        prof = 4.0 * X[nodes, 1] * (1 - X[nodes, 1])
        if dim == 3:
            prof = prof * 4.0 * X[nodes, 2] * (1 - X[nodes, 2])
        ux = np.where(inlet, prof, 0.0)
        return (np.concatenate([dim * nodes + a for a in range(dim)]).astype(np.int32),
                np.concatenate([ux] + [np.zeros(nodes.size)] * (dim - 1)))

    def run(ctx, dmap, out, key, part=None):
        ctx.set_coeffs(1.0, 1.0, 0.02)
        dofs, vals = bc(dmap)
        ctx.set_dirichlet(nat.VELOCITY, dofs, vals)
        singular = attach_schur_laplacian(ctx, dofs, part=part)
        assert not singular
        opts = ctx.default_step_opts()
        opts.momentum.rtol = 1e-12
        opts.momentum.precond = 1
        infos = []
        ctx.comm_stats(reset=True)
        for step in range(nsteps):
            ctx.set_bdf((1.0, -1.0, 0.0) if step == 0 else (1.5, -2.0, 0.5), k)
            infos.append(ctx.step_bdf(opts))
            ctx.advance(1)
        out[key] = (ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD), infos, ctx.comm_stats())

    ref = {}
    ctx0 = context(mesh, dm)
    attach_hierarchy(ctx0, mesh, coarsest=4 if dim == 3 else 8)
    run(ctx0, dm, ref, 0)
    u_ref, p_ref, inf_ref, _ = ref[0]
    ctx0.close()
    group = nat.local_group_create(size)
    if dim == 3:
        parts = [SlabPartition((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), n, n, n, r, size, coarsest=4) for r in range(size)]
    else:
        parts = [StripPartition((0.0, 0.0), (1.0, 1.0), n, n, r, size, coarsest=8) for r in range(size)]
    assert len(parts[0].levels) >= 1
    ctxs = []
    for r, part in enumerate(parts):
        pdm = part.dofmap
        c = nat.NsfemContext(part.mesh.coords, part.mesh.cells, pdm.p2_dofmap, pdm.p1_dofmap,
                             pdm.n_p2, pdm.n_p1)
        c.attach_local_comm(group, r)
        ctxs.append(c)
    out, errors = {}, []

    def worker(r):
        try:
            parts[r].attach(ctxs[r])
            run(ctxs[r], parts[r].dofmap, out, r, part=parts[r])
        except BaseException as exc:
            import traceback
            traceback.print_exc()
            errors.append((r, repr(exc)))
            os._exit(17)

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    u = np.zeros_like(u_ref)
    p = np.zeros_like(p_ref)
    for r, part in enumerate(parts):
        ul, pl, infos, stats = out[r]
        u.reshape(-1, dim)[part.p2_global[part.p2_owned]] = ul.reshape(-1, dim)[part.p2_owned]
        p[part.p1_global[part.p1_owned]] = pl[part.p1_owned]
        for a, b in zip(infos, inf_ref):
            assert a.newton_iterations == b.newton_iterations
            assert abs(a.krylov_iterations_momentum - b.krylov_iterations_momentum) <= 1
        with capsys.disabled():
            its = sum(i.krylov_iterations_momentum for i in infos)
            print(f"\n[algebraic Schur, {dim}D, rank {r}/{size}] per step: "
                  f"{stats['allreduce_calls'] / nsteps:.0f} all-reduces ({stats['allreduce_bytes'] / nsteps / 1e3:.1f} kB), "
                  f"{stats['exchanges'] / nsteps:.0f} halo exchanges incl. reverse adds "
                  f"({stats['exchange_bytes'] / nsteps / 1e6:.2f} MB), {its / nsteps:.1f} BiCGStab iterations")
    assert rel(u, u_ref) < 1e-10
    assert rel(p, p_ref) < 1e-9
    for c in ctxs:
        c.close()
    nat.local_group_destroy(group)


@pytest.mark.parametrize("scheme,size,overlap", [("ipcs", 2, False), ("ipcs", 4, True), ("bdf", 2, False), ("bdf", 3, True)])
def test_rcb_partitioned_dfg_channel_equals_single_context(scheme, size, overlap):
    """BASELINE configs[2] geometry (DFG channel with the cylinder, unstructured triangles, two
    red refinements) cut by recursive coordinate bisection of the coarsest cells
    (partition.GraphPartition): index-list halos with any number of neighbours
    (nsfem_set_halo_lists), rank-local prolongations of the refinement hierarchy, the replicated
    coarsest mesh addressed through an index list -- 2 / 3 / 4 in-process ranks reproduce the
    single-context IPCS and monolithic BDF-2 steps (open outlet: pressure Dirichlet values in IPCS,
    additive parts of the algebraic Schur Laplacian in the monolithic scheme)."""
    import grid_generator as gg
    from fem_mesh import TaylorHoodDofMap
    from multigrid import attach_schur_laplacian
    from partition import GraphPartition
    mesh, marks = gg.dfg_channel(4, 2)
    dm = TaylorHoodDofMap(mesh)
    M = gg.DFGBoundaryMarkers
    nsteps, k, H = 2, 0.05, 4.1

    def bc(dmap, mk):
        inlet = np.unique(dmap.facet_p2_nodes(mk.facets_with_id(M.inlet.value)))
        walls = np.unique(np.concatenate([dmap.facet_p2_nodes(mk.facets_with_id(m.value)).ravel()
                                          for m in (M.bottom, M.top, M.cylinder)]))
        y = dmap.p2_coords[inlet, 1]
        prof = 6.0 * y * (H - y) / H ** 2
        dofs = np.concatenate([2 * inlet, 2 * inlet + 1, 2 * walls, 2 * walls + 1])
        vals = np.concatenate([prof, np.zeros(inlet.size + 2 * walls.size)])
        # walls win on shared nodes (list order of DirichletBC.apply)
        _, first = np.unique(dofs[::-1], return_index=True)
        keep = dofs.size - 1 - first
        return dofs[keep].astype(np.int32), vals[keep]

    def run(ctx, dmap, mk, out, key, part=None):
        ctx.set_coeffs(1.0, 1.0, 0.05)
        dofs, vals = bc(dmap, mk)
        ctx.set_dirichlet(nat.VELOCITY, dofs, vals)
        outlet = np.unique(dmap.facet_p1_nodes(mk.facets_with_id(M.outlet.value))).astype(np.int32)
        if scheme == "ipcs":
            ctx.set_dirichlet(nat.PRESSURE, outlet, np.zeros(outlet.size))
        else:
            ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
            ctx.set_dirichlet(nat.PRESSURE_PRECOND, np.zeros(0, np.int32), np.zeros(0))
            assert not attach_schur_laplacian(ctx, dofs, part=part)
        if part is not None:
            ctx.set_overlap(overlap)
        opts = ctx.default_step_opts()
        for o in (opts.momentum, opts.poisson, opts.correction):
            o.rtol = 1e-12
        opts.momentum.precond = opts.poisson.precond = 1
        infos = []
        ctx.comm_stats(reset=True)
        for step in range(nsteps):
            ctx.set_bdf((1.0, -1.0, 0.0) if step == 0 else (1.5, -2.0, 0.5), k)
            infos.append(ctx.step_ipcs(opts) if scheme == "ipcs" else ctx.step_bdf(opts))
            ctx.advance(0 if scheme == "ipcs" else 1)
        out[key] = (ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD), infos, ctx.comm_stats(),
                    ctx.comm_overlapped() if part is not None else 0)

    ref = {}
    ctx0 = context(mesh, dm)
    assert attach_hierarchy(ctx0, mesh) == 2
    run(ctx0, dm, marks, ref, 0)
    u_ref, p_ref, inf_ref, _, _ = ref[0]
    ctx0.close()
    group = nat.local_group_create(size)
    parts = [GraphPartition(mesh, r, size, marks) for r in range(size)]
    assert sum(int(p.p2_owned.sum()) for p in parts) == dm.n_p2
    assert sum(int(p.p1_owned.sum()) for p in parts) == dm.n_p1
    assert max(len(p.p2_lists["neighbour"]) for p in parts) >= (2 if size > 2 else 1)
    ctxs = []
    for r, part in enumerate(parts):
        pdm = part.dofmap
        c = nat.NsfemContext(part.mesh.coords, part.mesh.cells, pdm.p2_dofmap, pdm.p1_dofmap,
                             pdm.n_p2, pdm.n_p1)
        c.attach_local_comm(group, r)
        ctxs.append(c)
    out, errors = {}, []

    def worker(r):
        try:
            assert parts[r].attach(ctxs[r]) == 2
            run(ctxs[r], parts[r].dofmap, parts[r].markers, out, r, part=parts[r])
        except BaseException as exc:
            import traceback
            traceback.print_exc()
            errors.append((r, repr(exc)))
            os._exit(17)

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    u = np.zeros_like(u_ref)
    p = np.zeros_like(p_ref)
    for r, part in enumerate(parts):
        ul, pl, infos, stats, overlapped = out[r]
        g2 = part.p2_global(dm)
        u.reshape(-1, 2)[g2[part.p2_owned]] = ul.reshape(-1, 2)[part.p2_owned]
        p[part.p1_global[part.p1_owned]] = pl[part.p1_owned]
        for a, b in zip(infos, inf_ref):
            assert a.newton_iterations == b.newton_iterations
            # (BiCGStab on the mixed system to rtol 1e-12: ~30 iterations per Newton step, whose count
            #  moves by one or two with the summation order of the partitioned dot products)
            assert abs(a.krylov_iterations_momentum - b.krylov_iterations_momentum) <= max(1, 0.05 * b.krylov_iterations_momentum)
            assert abs(a.krylov_iterations_poisson - b.krylov_iterations_poisson) <= 1
        assert stats["exchanges"] > 0
        assert (overlapped > 0) == overlap
    assert rel(u, u_ref) < 1e-9
    assert rel(p, p_ref) < 1e-8
    for c in ctxs:
        c.close()
    nat.local_group_destroy(group)


def test_channel_bench_thread_ranks_match_the_single_rank_run():
    """bench.py --workload channel3d-bdf (BASELINE configs[4]) on 1 rank and on 2 / 4 thread ranks
    (--local-ranks: the N-rank code path of the bench -- slabs, additive Schur parts, the reductions
    of its invariants -- through the in-process communicator on one GPU), strong scaling = the
    same mesh: same Newton counts, the same boundary fluxes, invariants green on every run."""
    def run(extra):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "channel3d-bdf", "--cells", "8",
               "--steps", "3", "--warmup", "1", "--krylov-rtol", "1e-10", "--newton-forcing", "0",
               "--no-cpu-baseline"] + extra
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-2000:]
        return json.loads(res.stdout.strip().splitlines()[-1])
    one = run([])
    assert one["n_gpus"] == 1 and one["config"]["invariants"]["mass_balance_rel"] < 1e-6
    for ranks in (2, 4):
        many = run(["--local-ranks", str(ranks), "--scaling", "strong"])
        assert many["n_gpus"] == ranks and many["scaling"] == "strong"
        assert many["config"]["n_dofs"] == one["config"]["n_dofs"]
        assert many["config"]["newton_its_per_step"] == one["config"]["newton_its_per_step"]
        fa, fb = many["config"]["invariants"]["flux"], one["config"]["invariants"]["flux"]
        for side in fb:
            assert abs(fa[side] - fb[side]) < 1e-8 * abs(fb["inlet"]), side
        stats = many["config"]["comm_per_step_rank0"]
        assert stats["exchanges"] > 0 and stats["allreduce_calls"] > 0


def test_dfg_bench_thread_ranks_match_the_single_rank_run():
    """bench.py --workload dfg-bdf (BASELINE configs[2]) on 1 rank and on 2 / 3 thread ranks
    (recursive-bisection parts, index-list halos, additive Schur parts): the drag / lift
    coefficients -- integrated per rank over the cylinder facets of its own cells and summed --
    and the net boundary mass flux agree with the single-rank run."""
    def run(extra):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "dfg-bdf", "--dfg-refine", "2",
               "--steps", "3", "--warmup", "1", "--krylov-rtol", "1e-10", "--newton-forcing", "0",
               "--no-cpu-baseline"] + extra
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-2000:]
        return json.loads(res.stdout.strip().splitlines()[-1])
    one = run([])
    assert one["n_gpus"] == 1
    for ranks in (2, 3):
        many = run(["--local-ranks", str(ranks)])
        assert many["n_gpus"] == ranks and many["scaling"] == "strong"
        assert many["config"]["n_dofs"] == one["config"]["n_dofs"]
        assert many["config"]["newton_its_per_step"] == one["config"]["newton_its_per_step"]
        for key in ("drag_lift_reference_formula", "drag_lift_newtonian_stress"):
            assert np.abs(np.array(many["config"][key]) - np.array(one["config"][key])).max() < 1e-7
        assert abs(many["config"]["cylinder_perimeter_of_the_mesh"] - one["config"]["cylinder_perimeter_of_the_mesh"]) < 1e-12
        assert abs(many["config"]["net_boundary_mass_flux"]) < 1e-9
        assert many["config"]["comm_per_step_rank0"]["exchanges"] > 0


def test_bench_through_rccl_single_rank():
    """bench.py with the RCCL communicator attached (1 rank): ncclCommInitRank, the all-reduces
    of every dot product and the torch.distributed(gloo) bootstrap all run for real."""
    env = dict(os.environ, NSFEM_FORCE_COMM="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    # (the projection step is the direct fast-diagonalisation solve in both runs; with the communicator its
    # contraction over y goes through ncclAllReduce: FastDiag::apply_strip)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--cells", "64", "--steps", "2", "--warmup",
           "1", "--no-cpu-baseline", "--no-other-configs"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["metric"] == "dof_updates_per_sec" and line["value"] > 0
    # the same run without a communicator gives the same iteration counts
    env2 = {k: v for k, v in os.environ.items() if k != "NSFEM_FORCE_COMM"}
    res2 = subprocess.run(cmd, env=env2, capture_output=True, text=True, timeout=600)
    assert res2.returncode == 0, res2.stderr[-2000:]
    line2 = json.loads(res2.stdout.strip().splitlines()[-1])
    for key in ("newton_its_per_step", "bicgstab_its_per_step", "poisson_cg_its_per_step"):
        assert line["config"][key] == line2["config"][key]
    assert line["config"]["poisson_cg_its_per_step"] == 1.0 and "fast diagonalisation" in line["config"]["poisson_solver"]
    assert line["config"]["comm_per_step_rank0"]["allreduce_bytes"] > 65 * 65 * 8       # (the H x W array of the solve)


def test_bench_spawns_its_own_rank_processes_on_a_shared_gpu():
    """`python bench.py --gpus 2` AS TYPED (no launcher, no WORLD_SIZE): the parent never touches the GPU, spawns one
    process per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run would set them), relays rank
    0's JSON line and exits 0.  On this one-GPU box the two rank PROCESSES share the device (NSFEM_SHARE_GPU), so the
    collectives go through the host-staged shared-memory communicator (RCCL refuses two ranks on one GPU); everything
    else -- gloo rendezvous, partitions, halo exchanges, all-reduces, the weak run AND the strong-scaling run of the
    same job -- is the code path of a multi-GPU node."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["NSFEM_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cells", "128", "--steps", "4", "--warmup", "2",
           "--halo-mode", "exact", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    # round 4: the HEADLINE of an N > 1 line is north_star's strong-scaling quantity -- ONE mesh cut into 2 strips --,
    # the weak-scaling figure (the same cells PER RANK) rides in the same line
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["n_dofs"] == 2 * 257 * 257 + 129 * 129
    assert line["config"]["validation"]["max_rel_diff_velocity_vs_exact"] < 1e-6
    assert line["config"]["comm_per_step_rank0"]["exchanges"] > 0
    wk = line["weak"]
    assert wk["cells"] == 128 and wk["n_dofs"] > line["config"]["n_dofs"] and wk["value"] > 0 and wk["comm_per_step"]["exchanges"] > 0
    # the strong run is the single-rank mesh: same iteration counts as one rank (exact halo mode; the projection step
    # is the direct fast-diagonalisation solve on one rank and on the strips alike)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cells", "128", "--steps", "4", "--warmup", "2",
                          "--timed-only"], env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    ref = json.loads(one.stdout.strip().splitlines()[-1])
    for key in ("newton_its_per_step", "bicgstab_its_per_step", "poisson_cg_its_per_step"):
        assert abs(line["config"][key] - ref["config"][key]) <= 0.26, key
    # a rank that dies takes the job down with a non-zero exit code instead of leaving the others in a barrier
    bad = subprocess.run(cmd + ["--mg-truncation", "not-a-number"], env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "rank" in bad.stderr


def test_message_counts_of_a_strong_scaling_step_on_eight_ranks():
    """BASELINE's strong-scaling mesh (960 x 960, 8.3 M dofs) on 8 thread ranks, the N > 1 defaults of bench.py (relaxed
    halo mode, levels thinner than 16 cell rows per rank replicated, no exchange after a globally solved child): the
    communicator's own counters per time step.  Round 2: 194 exchanges + 45 all-reduces; round 3: 127 + 43; round 4:
    the projection step is the direct fast-diagonalisation solve on the strips (one all-reduce of the 961 x 961
    transformed array instead of ~5 multigrid-CG iterations with ~10 exchanges each).  The iteration counts must stay
    those of the coupled cycle."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--local-ranks", "8", "--scaling", "strong", "--cells", "960",
           "--steps", "4", "--warmup", "2", "--timed-only"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([l for l in res.stdout.strip().splitlines() if l.startswith("{")][-1])
    comm = line["config"]["comm_per_step_rank0"]
    print("\n[strong 960^2, 8 ranks] per step: %.1f halo exchanges (%.2f MB), %.1f all-reduces (%.2f MB); its %s" % (
        comm["exchanges"], comm["exchange_bytes"] / 1e6, comm["allreduce_calls"], comm["allreduce_bytes"] / 1e6,
        [line["config"][k] for k in ("newton_its_per_step", "bicgstab_its_per_step", "poisson_cg_its_per_step")]))
    assert comm["exchanges"] <= 85 and comm["allreduce_calls"] <= 32          # (measured: 80.8 and 31.0)
    assert line["config"]["newton_its_per_step"] <= 2.01 and line["config"]["bicgstab_its_per_step"] <= 5.5
    assert line["config"]["poisson_cg_its_per_step"] == 1.0


def _visible_gpus():
    import torch
    return torch.cuda.device_count()          # (does not initialise the GPU on this image)


@pytest.mark.skipif(_visible_gpus() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
@pytest.mark.parametrize("scaling,overlap", [("strong", "on"), ("weak", "off")])
def test_two_rccl_ranks_on_two_gpus_reproduce_the_single_gpu_run(scaling, overlap):
    """Real multi-rank RCCL (grouped ncclSend / ncclRecv halo exchange + ncclAllReduce), one
    process per GPU through torch.distributed.run, exactly as the driver launches bench.py:
    same iteration counts as the one-GPU run of the same global mesh (strong scaling) and a
    validated line.  Skipped on one-GPU boxes -- there the partitioned algorithm is covered by the
    in-process ranks above."""
    port = "29563" if scaling == "strong" else "29564"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--cells", "128", "--steps", "4", "--warmup", "2", "--scaling", scaling, "--overlap", overlap,
           "--halo-mode", "exact", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([l for l in res.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == scaling
    assert line["config"]["validation"]["max_rel_diff_velocity_vs_exact"] < 1e-6
    assert line["config"]["comm_per_step_rank0"]["exchanges"] > 0
    if scaling == "strong":
        one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cells", "128", "--steps", "4",
                              "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
        assert one.returncode == 0, one.stderr[-3000:]
        ref = json.loads(one.stdout.strip().splitlines()[-1])
        for key in ("newton_its_per_step", "bicgstab_its_per_step", "poisson_cg_its_per_step"):
            assert abs(line["config"][key] - ref["config"][key]) <= 0.26, key


@pytest.mark.parametrize("dim", [2, 3])
def test_chebyshev_mass_solve_matches_cg(dim):
    """correction.precond = 2: Chebyshev iteration on diag(M)^-1 M with Wathen's element bounds
    (no dot products) must give the Jacobi-CG result of the velocity-correction step."""
    if dim == 2:
        mesh, dm, _ = box(24, 24)
        bc = _cavity_bc(dm)
    else:
        from fem_mesh import box_mesh, TaylorHoodDofMap
        mesh = box_mesh((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), 6, 6, 6)
        dm = TaylorHoodDofMap(mesh)
        X = dm.p2_coords
        on = ((np.abs(X) < 1e-12) | (np.abs(X - 1) < 1e-12)).any(axis=1)
        nodes = np.nonzero(on)[0]
        lid = np.abs(X[nodes, 2] - 1) < 1e-12
        bc = (np.concatenate([3 * nodes, 3 * nodes + 1, 3 * nodes + 2]).astype(np.int32),
              np.concatenate([np.where(lid, 1.0, 0.0), np.zeros(2 * nodes.size)]))
    res = {}
    for cheb in (False, True):
        ctx = context(mesh, dm)
        ctx.set_coeffs(1.0, 1.0, 0.01)
        ctx.set_dirichlet(nat.VELOCITY, *bc)
        ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
        opts = ctx.default_step_opts()
        for o in (opts.momentum, opts.poisson, opts.correction):
            o.rtol = 1e-12
        opts.correction.precond = 2 if cheb else 0
        its = []
        for step in range(3):
            ctx.set_bdf((1.0, -1.0, 0.0) if step == 0 else (1.5, -2.0, 0.5), 0.01)
            info = ctx.step_ipcs(opts)
            its.append(info.krylov_iterations_correction)
            ctx.advance(0)
        res[cheb] = (ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD), its)
        ctx.close()
    assert rel(res[True][0], res[False][0]) < 1e-10
    assert rel(res[True][1], res[False][1]) < 1e-8
    assert max(res[True][2]) <= (60 if dim == 2 else 130)


@pytest.mark.parametrize("dim,size,tail,relaxed", [(3, 2, False, False), (3, 4, True, False), (2, 2, False, False),
                                                   (2, 4, True, False), (3, 4, True, True), (2, 4, True, True)])
def test_partitioned_triple_periodic_taylor_green_equals_single_context(dim, size, tail, relaxed):
    """BASELINE configs[3] in small: Taylor-Green vortex on the triple-periodic cube, IPCS, slabs
    along z whose halo exchange WRAPS AROUND (PeriodicSlabPartition: x, y periodic inside every
    slab through the dof maps, z periodic across the ranks), periodic multigrid levels with the
    replicated periodic global coarse problem -- in-process ranks against the single context with
    the triple-periodic dof map."""
    import dlfn_compat as dlfn
    from fem_mesh import TaylorHoodDofMap, box_mesh, periodic_entity_map, rectangle_mesh
    from partition import PeriodicSlabPartition, PeriodicStripPartition

    class TriplePeriodic(dlfn.SubDomain):          # (doubly periodic in 2D)
        def inside(self, x, on_boundary):
            return bool(on_boundary and any(dlfn.near(x[a], 0.0) for a in range(dim)))

        def map(self, x_slave, x_master):
            for a in range(dim):
                if dlfn.near(x_slave[a], 1.0):
                    x_master[:] = x_slave
                    x_master[a] -= 1.0
                    return
            x_master[:] = -10.0

    n, nsteps, k, g = (8 if dim == 3 else 16), 3, 0.02, 2.0 * np.pi

    def fields(dmap):
        X, Y = dmap.p2_coords, dmap.p1_coords
        comps = [np.cos(g * X[:, 0]) * np.sin(g * X[:, 1]), -np.sin(g * X[:, 0]) * np.cos(g * X[:, 1])]
        if dim == 3:
            comps.append(0.3 * np.sin(g * X[:, 2]) * np.cos(g * X[:, 0]))
        return np.stack(comps, axis=1).ravel(), -0.25 * (np.cos(2 * g * Y[:, 0]) + np.cos(2 * g * Y[:, 1]))

    def run(ctx, dmap, out, key):
        u0, p0 = fields(dmap)
        for slot in (nat.U0, nat.U1, nat.U2):
            ctx.set_state(slot, u0)
        for slot in (nat.P, nat.P_OLD):
            ctx.set_state(slot, p0)
        ctx.set_coeffs(1.0, 1.0, 0.02)
        ctx.set_dirichlet(nat.VELOCITY, np.zeros(0, np.int32), np.zeros(0))
        ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
        opts = ctx.default_step_opts()
        for o in (opts.momentum, opts.poisson, opts.correction):
            o.rtol = 1e-12
        opts.momentum.precond = opts.poisson.precond = 1
        opts.correction.precond = 2
        infos = []
        means = []
        for step in range(nsteps):
            ctx.set_bdf((1.0, -1.0, 0.0) if step == 0 else (1.5, -2.0, 0.5), k)
            infos.append(ctx.step_ipcs(opts))
            means.append(ctx.shift_mean_pressure(0.0))        # global mean, all-reduced over the ranks
            ctx.advance(0)
        out[key] = (ctx.get_state(nat.U1), ctx.get_state(nat.P_OLD), infos)
        out[("means", key)] = means

    lo, hi = (0.0, ) * dim, (1.0, ) * dim
    mesh = box_mesh(lo, hi, n, n, n) if dim == 3 else rectangle_mesh(lo, hi, n, n)
    domain = TriplePeriodic()
    dm = TaylorHoodDofMap(mesh, periodic_map=periodic_entity_map(mesh, domain))
    ref = {}
    ctx0 = context(mesh, dm)
    attach_hierarchy(ctx0, mesh, coarsest=2, periodic=(domain, dm.p1_vertex_node))
    run(ctx0, dm, ref, 0)
    u_ref, p_ref, inf_ref = ref[0]
    ctx0.close()

    group = nat.local_group_create(size)
    if dim == 3:
        parts = [PeriodicSlabPartition(lo, hi, n, n, n, r, size, coarsest=4 if tail else 2,
                                       global_coarsest=2 if tail else None) for r in range(size)]
    else:
        parts = [PeriodicStripPartition(lo, hi, n, n, r, size, coarsest=8 if tail else 4,
                                        global_coarsest=2 if tail else None) for r in range(size)]
    ctxs = []
    for r, part in enumerate(parts):
        pdm = part.dofmap
        c = nat.NsfemContext(part.mesh.coords, part.mesh.cells, pdm.p2_dofmap, pdm.p1_dofmap,
                             pdm.n_p2, pdm.n_p1)
        c.attach_local_comm(group, r)
        ctxs.append(c)
    out, errors = {}, []

    def worker(r):
        try:
            parts[r].attach(ctxs[r])
            ctxs[r].mg_set_halo_mode(relaxed)          # (bench.py's default on several GPUs)
            run(ctxs[r], parts[r].dofmap, out, r)
        except BaseException as exc:
            import traceback
            traceback.print_exc()
            sys.stderr.flush()
            errors.append((r, repr(exc)))
            os._exit(17)

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    # compare by coordinates (mod 1): key -> reference dof
    key = lambda X: [tuple(r) for r in (np.round(X * 64).astype(np.int64) % 64)]
    ref2 = {kk: i for i, kk in enumerate(key(dm.p2_coords))}
    ref1 = {kk: i for i, kk in enumerate(key(dm.p1_coords))}
    u = np.full_like(u_ref, np.nan)
    p = np.full_like(p_ref, np.nan)
    for r, part in enumerate(parts):
        ul, pl, infos = out[r]
        own2, own1 = np.nonzero(part.p2_owned)[0], np.nonzero(part.p1_owned)[0]
        i2 = np.array([ref2[kk] for kk in key(part.dofmap.p2_coords[own2])])
        i1 = np.array([ref1[kk] for kk in key(part.dofmap.p1_coords[own1])])
        u.reshape(-1, dim)[i2] = ul.reshape(-1, dim)[own2]
        p[i1] = pl[own1]
        for a, b in zip(infos, inf_ref):
            assert a.newton_iterations == b.newton_iterations
            # (relaxed smoothing on slabs only two cube layers thick: a fifth more iterations at most)
            slack = (lambda ref_its: max(3, 0.2 * ref_its)) if relaxed else (lambda ref_its: 1)
            assert abs(a.krylov_iterations_momentum - b.krylov_iterations_momentum) <= slack(b.krylov_iterations_momentum)
            assert abs(a.krylov_iterations_poisson - b.krylov_iterations_poisson) <= slack(b.krylov_iterations_poisson)
    assert np.isfinite(u).all() and np.isfinite(p).all()          # every dof is owned by exactly one rank
    assert rel(u, u_ref) < 1e-9
    assert rel(p, p_ref) < 1e-8                                   # both shifted to zero mean
    for r in range(size):     # every rank computes the same GLOBAL mean (the level of the singular
        assert np.abs(np.array(out[("means", r)]) - np.array(out[("means", 0)])).max() < 1e-13
        if not relaxed:       # Poisson solution itself depends on the preconditioner: exact mode only)
            assert np.abs(np.array(out[("means", r)]) - np.array(ref[("means", 0)])).max() < 1e-10
    for c in ctxs:
        c.close()
    nat.local_group_destroy(group)
