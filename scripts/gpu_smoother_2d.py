import os, sys
sys.path[:0] = ["/root/repo/navierstokes-with-fenics_amd", "/root/repo/tests"]
import numpy as np, _native as nat
from gpu_common import box, context
from multigrid import attach_hierarchy
for n in (512, 1024):
    mesh, dm, _ = box(n, n)
    ctx = context(mesh, dm); attach_hierarchy(ctx, mesh)
    ctx.set_coeffs(1.0, 1.0, 0.01); ctx.set_bdf((1.5, -2.0, 0.5), 1e-3)
    ms, nb = ctx.time_spmv(nat.OP_MOMENTUM_SMOOTHER, 100)
    info = ctx.smoother_info()
    print("2D n %d: %s smoother (flush-interleaved) %.1f us, %.2f TB/s of its own algorithmic bytes (%.0f MB); "
          "CSR-equivalent %.0f MB = %.2f TB/s" % (n, info["kind"], ms * 1e3, nb / ms / 1e9, nb / 1e6,
                                                   info["csr_bytes"] / 1e6, info["csr_bytes"] / ms / 1e9))
    ctx.close()
