"""Scratch: cold-cache (flush-interleaved) time of the finest-level smoothing launch for one
numbering / kernel variant.  usage: gpu_sell_tune.py DIM N lex|parity   (env NSFEM_SELL, NSFEM_SELL_VARIANT)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd")]
import numpy as np
import _native as nat
from fem_mesh import TaylorHoodDofMap, box_mesh, rectangle_mesh
from multigrid import attach_hierarchy
dim, n, order = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
mesh = rectangle_mesh((0.0, 0.0), (1.0, 1.0), n, n) if dim == 2 else box_mesh((0, 0, 0), (1, 1, 1), n, n, n)
if os.environ.get("PARITY_BLOCK"):
    mesh.parity_block = int(os.environ["PARITY_BLOCK"])
dm = TaylorHoodDofMap(mesh, reorder="parity" if order == "parity" else True)
ctx = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1)
attach_hierarchy(ctx, mesh)
ctx.set_coeffs(1.0, 1.0, 0.01)
ctx.set_bdf((1.5, -2.0, 0.5), 1e-3 if dim == 2 else 0.25 / n)
ms, nb = ctx.time_spmv(nat.OP_MOMENTUM_SMOOTHER, 100)
print("block %s balance %s " % (os.environ.get("PARITY_BLOCK", "-"), os.environ.get("NSFEM_SELL_BALANCE", "1")), end="")
info = ctx.smoother_info()
print("dim %d n %d %-6s kernel %s (NSFEM_DICT=%s NSFEM_SELL=%s): smoother cold %.1f us  %.2f TB/s of its own algorithmic bytes "
      "(%.0f MB; CSR-equivalent %.0f MB = %.2f TB/s)" % (
          dim, n, order, info["kind"], os.environ.get("NSFEM_DICT", "1"), os.environ.get("NSFEM_SELL", "1"), ms * 1e3,
          nb / ms / 1e9, nb / 1e6, info["csr_bytes"] / 1e6, info["csr_bytes"] / ms / 1e9))
ctx.close()
