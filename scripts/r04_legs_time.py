"""Kernel times of the fused multigrid legs: 30 cycles of each hierarchy at n (under rocprofv3: scripts/r04_trace_script.sh).
NSFEM_LIB=build/knockouts/libnsfem_hip.so + NSFEM_LEG_MAXOPS=k: only the first k operations of every leg (wrong results)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "navierstokes-with-fenics_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import _native as nat  # noqa: E402
if os.environ.get("NSFEM_LIB"):
    nat._lib = nat.load_library(os.path.join(ROOT, os.environ["NSFEM_LIB"]))
from gpu_common import box, cavity_bc, context  # noqa: E402
from multigrid import attach_hierarchy  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
k = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
mesh, dm, marks = box(n, n)
mesh.structured = ((0.0, 0.0), (1.0, 1.0), n, n)
bd, bv = cavity_bc(dm, marks)
ctx = context(mesh, dm)
attach_hierarchy(ctx, mesh)
ctx.set_coeffs(1.0, 1.0, 0.01)
ctx.set_dirichlet(nat.VELOCITY, bd.astype(np.int32), bv)
ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
ctx.set_bdf((1.5, -2.0, 0.5), k)
rng = np.random.default_rng(7)
rp = rng.standard_normal(dm.n_p1)
rv = rng.standard_normal(dm.n_velocity)
for _ in range(30):
    ctx.mg_apply(0, rp)
    ctx.mg_apply(1, rv)
print(ctx.mg_info(0), ctx.mg_info(1))
ctx.close()
