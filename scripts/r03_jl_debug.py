"""debug: where does k_jac_lattice differ from the launch pair?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("tests", "navierstokes-with-fenics_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, d))
import _native as nat
from gpu_common import box, cavity_bc, context
nx, ny = int(sys.argv[1]), int(sys.argv[2])
mesh, dm, marks = box(nx, ny, p1=(nx / 16.0, ny / 16.0))
bd, bv = cavity_bc(dm, marks)
rng = np.random.default_rng(1)
u = rng.standard_normal(dm.n_velocity); x = rng.standard_normal(dm.n_velocity)
out = {}
for tag, env in (("pair", "0"), ("lattice", "1")):
    os.environ["NSFEM_JAC_LATTICE"] = env
    ctx = context(mesh, dm)
    ctx.set_coeffs(float(sys.argv[3]) if len(sys.argv) > 3 else 0.8, 1.0, 0.02)
    ctx.set_bdf((1.5, -2.0, 0.5), 0.05)
    ctx.set_dirichlet(nat.VELOCITY, bd.astype(np.int32), bv)
    ctx.set_state(nat.USTAR, u)
    print(tag, ctx.jacobian_info())
    out[tag] = ctx.operator_apply(nat.OP_MOMENTUM_JAC_MF, x)
    ctx.close()
a, b = out["pair"].reshape(-1, 2), out["lattice"].reshape(-1, 2)
W = 2 * nx + 1
d = np.abs(a - b).max(axis=1)
print("max abs diff", d.max(), "rel", d.max() / np.abs(a).max(), "nodes differing", int((d > 0).sum()), "of", d.size)
idx = np.nonzero(d > 0)[0]
j, i = idx // W, idx % W
for cls in range(4):
    sel = ((i & 1) + 2 * (j & 1)) == cls
    print("class", cls, "differing", int(sel.sum()), "max", d[idx[sel]].max() if sel.any() else 0.0)
print("first differing (i, j):", list(zip(i[:12].tolist(), j[:12].tolist())))
