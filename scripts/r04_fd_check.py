"""Fast-diagonalisation projection step (csrc/fastdiag.hip) on the GPU: z = A^+ r against the numpy reference and the
oracle's matrix, then IPCS steps with precond = 3 against the multigrid-CG path.  Usage: python scripts/r04_fd_check.py [n ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "navierstokes-with-fenics_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import _native as nat  # noqa: E402
import poisson_fd as pf  # noqa: E402
from gpu_common import box, cavity_bc, context, rel  # noqa: E402
from multigrid import attach_hierarchy  # noqa: E402


def run(nx, ny, fd, outlet, steps, k=1e-3):
    ext = (nx / float(max(nx, ny)), ny / float(max(nx, ny)))
    mesh, dm, marks = box(nx, ny, p1=ext)
    mesh.structured = ((0.0, 0.0), ext, nx, ny)
    bd, bv = cavity_bc(dm, marks)
    ctx = context(mesh, dm)
    attach_hierarchy(ctx, mesh)
    ctx.set_coeffs(1.0, 1.0, 0.01)
    ctx.set_dirichlet(nat.VELOCITY, bd.astype(np.int32), bv)
    nodes = np.zeros(0, np.int32)
    if outlet:
        nodes = np.where(np.abs(mesh.coords[:, 0] - mesh.coords[:, 0].max()) < 1e-12)[0].astype(np.int32)
    ctx.set_dirichlet(nat.PRESSURE, nodes, np.zeros(nodes.size))
    out = {}
    xs, ys = pf.lattice_lines(mesh)
    f = pf.factors(xs, ys, nodes)
    ctx.poisson_set_fast_diag(f)
    rng = np.random.default_rng(3)
    r = rng.standard_normal(dm.n_p1)
    r[nodes] = 0.0
    if not outlet:
        r -= r.mean()
    z = ctx.mg_apply(2, r)
    out["gemm_vs_numpy"] = rel(z, pf.apply_reference(f, r))
    opts = ctx.default_step_opts()
    for o in (opts.momentum, opts.poisson, opts.correction):
        o.rtol = 1e-8
    opts.momentum.precond = 1
    opts.poisson.precond = 3 if fd else 1
    opts.correction.precond = 2
    opts.newton_forcing = 1e-4
    opts.pressure_extrapolation = 1
    its = []
    for step in range(steps + 3):
        if step == 3:
            ctx.synchronize()
            t0 = time.perf_counter()
        ctx.set_bdf((1.0, -1.0, 0.0) if step == 0 else (1.5, -2.0, 0.5), k)
        info = ctx.step_ipcs(opts)
        ctx.advance(0)
        its.append((info.newton_iterations, info.krylov_iterations_momentum, info.krylov_iterations_poisson))
    ctx.synchronize()
    out["ms"] = (time.perf_counter() - t0) / steps * 1e3
    out["its"] = its[-2:]
    out["u"] = ctx.get_state(nat.U1)
    out["p"] = ctx.get_state(nat.P_OLD)
    ctx.close()
    return out


if __name__ == "__main__":
    for n in [int(a) for a in sys.argv[1:]] or [32, 64]:
        for (nx, ny, outlet) in ((n, n, False), (n, n // 2 + 3, True)):
            steps = 30 if n >= 256 else 4
            a = run(nx, ny, False, outlet, steps)
            b = run(nx, ny, True, outlet, steps)
            print("n %d x %d outlet %d: gemm vs numpy %.2e | mg-cg %.3f ms %s | fast-diag %.3f ms %s | du %.1e dp %.1e" % (
                nx, ny, outlet, b["gemm_vs_numpy"], a["ms"], a["its"], b["ms"], b["its"], rel(b["u"], a["u"]),
                rel(b["p"] - b["p"].mean(), a["p"] - a["p"].mean())))
            sys.stdout.flush()
