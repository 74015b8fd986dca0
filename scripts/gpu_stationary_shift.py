"""Scratch: stationary lid-driven cavity / channel at higher Reynolds numbers -- Krylov iteration
counts of the stationary Newton solves as a function of the preconditioner mass shift."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import _native as nat
from gpu_common import box, cavity_bc, context
from multigrid import attach_hierarchy
n, Re = int(sys.argv[1]), float(sys.argv[2])
shifts = [float(v) for v in sys.argv[3:]]
mesh, dm, marks = box(n, n)
for shift in shifts:
    ctx = context(mesh, dm)
    attach_hierarchy(ctx, mesh)
    ctx.set_coeffs(1.0, 1.0, 1.0 / Re)
    ctx.set_dirichlet(nat.VELOCITY, *cavity_bc(dm, marks))
    ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
    ctx.set_dirichlet(nat.PRESSURE_PRECOND, np.zeros(0, np.int32), np.zeros(0))
    ctx.set_bdf((0.0, 0.0, 0.0), 1.0)
    ctx.set_preconditioner_shift(shift)
    o = ctx.default_step_opts()
    o.momentum.rtol, o.momentum.precond, o.momentum.max_iter = 1e-8, 1, 3000
    o.newton_forcing = 1e-3
    t0 = time.time()
    try:
        o.picard, o.newton_atol, o.newton_max_iter, o.allow_nonconvergence = 1, 1e-2, 10, 1
        a = ctx.step_bdf(o)
        o.picard, o.newton_atol, o.newton_max_iter, o.allow_nonconvergence = 0, 1e-10, 30, 1
        b = ctx.step_bdf(o)
        print("n %d Re %g shift %g: picard %d its / %d kry, newton %d its / %d kry, final |F| %.2e, %.2fs" % (
            n, Re, shift, a.newton_iterations, a.krylov_iterations_momentum, b.newton_iterations,
            b.krylov_iterations_momentum, b.newton_residuals[b.newton_iterations], time.time() - t0))
    except Exception as e:
        print("n %d Re %g shift %g: EXC %s" % (n, Re, shift, e))
    ctx.close()
