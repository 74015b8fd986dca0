"""Fused multigrid legs (csrc/mglegs.hip) against the separate launches: one preconditioner application z = M^-1 r
of the pressure and the velocity hierarchy on the same seeded vector, NSFEM_MG_LEGS=0 vs 1, plus timing of the cycles
inside IPCS steps.  Usage: python scripts/r04_legs_check.py [n ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "navierstokes-with-fenics_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import _native as nat  # noqa: E402
if os.environ.get("NSFEM_LIB"):          # (experiment builds, e.g. build/knockouts/libnsfem_hip.so)
    nat.load_library(os.environ["NSFEM_LIB"])
from gpu_common import box, cavity_bc, context, rel  # noqa: E402
from multigrid import attach_hierarchy  # noqa: E402


def run(nx, ny, legs, k=1e-3, outlet=False, steps=0, group=None):
    os.environ["NSFEM_MG_LEGS"] = "1" if legs else "0"
    if group is not None:
        os.environ["NSFEM_LEG_GROUP"] = str(group)
    mesh, dm, marks = box(nx, ny, p1=(nx / float(max(nx, ny)), ny / float(max(nx, ny))))
    mesh.structured = ((0.0, 0.0), (nx / float(max(nx, ny)), ny / float(max(nx, ny))), nx, ny)
    bd, bv = cavity_bc(dm, marks)
    ctx = context(mesh, dm)
    attach_hierarchy(ctx, mesh)
    ctx.set_coeffs(1.0, 1.0, 0.01)
    ctx.set_dirichlet(nat.VELOCITY, bd.astype(np.int32), bv)
    if outlet:
        pn = np.unique(marks.mesh.cells[0:0]) if False else None
        nodes = np.where(np.abs(mesh.coords[:, 0] - mesh.coords[:, 0].max()) < 1e-12)[0].astype(np.int32)
        ctx.set_dirichlet(nat.PRESSURE, nodes, np.zeros(nodes.size))
    else:
        ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    ctx.set_bdf((1.5, -2.0, 0.5), k)
    rng = np.random.default_rng(7)
    rp = rng.standard_normal(dm.n_p1)
    rv = rng.standard_normal(dm.n_velocity)
    out = dict(zp=ctx.mg_apply(0, rp), zv=ctx.mg_apply(1, rv), ip=ctx.mg_info(0), iv=ctx.mg_info(1))
    if steps:
        opts = ctx.default_step_opts()
        for o in (opts.momentum, opts.poisson, opts.correction):
            o.rtol = 1e-8
        opts.momentum.precond = opts.poisson.precond = 1
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
        out["its"] = its[-3:]
        out["u"] = ctx.get_state(nat.U1)
        out["p"] = ctx.get_state(nat.P_OLD)
    ctx.close()
    return out


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [64, 128]
    for n in sizes:
        for (nx, ny, outlet) in ((n, n, False), (n, n // 2, True)):
            steps = 20 if n >= 256 else 3
            a = run(nx, ny, False, outlet=outlet, steps=steps)
            b = run(nx, ny, True, outlet=outlet, steps=steps)
            print("n %d x %d outlet %d: poisson legs %s velocity legs %s" % (nx, ny, outlet, b["ip"], b["iv"]))
            print("   cycle  rel diff  pressure %.2e   velocity %.2e" % (rel(b["zp"], a["zp"]), rel(b["zv"], a["zv"])))
            if steps:
                print("   steps: separate %.3f ms %s | legs %.3f ms %s | du %.1e dp %.1e" % (
                    a["ms"], a["its"], b["ms"], b["its"], rel(b["u"], a["u"]),
                    rel(b["p"] - b["p"].mean(), a["p"] - a["p"].mean())))
            sys.stdout.flush()
