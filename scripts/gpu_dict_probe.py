"""which lattice sizes get a stencil dictionary: python scripts/gpu_dict_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd")]
import _native as nat
from fem_mesh import TaylorHoodDofMap, box_mesh, rectangle_mesh, preferred_p2_order
from multigrid import attach_hierarchy
for dim, n in ((2, 48), (2, 64), (2, 512), (3, 12), (3, 16), (3, 20), (3, 24), (3, 32), (3, 48), (3, 64)):
    mesh = rectangle_mesh((0.0, 0.0), (1.0, 1.0), n, n) if dim == 2 else box_mesh((0, 0, 0), (1, 1, 1), n, n, n)
    mesh.structured = ((0.0,) * dim, (1.0,) * dim) + (n,) * dim
    for order in ((True, "parity") if dim == 3 else (True,)):
        dm = TaylorHoodDofMap(mesh, reorder=order)
        ctx = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1)
        attach_hierarchy(ctx, mesh)
        ctx.set_coeffs(1.0, 1.0, 0.01)
        ctx.set_bdf((1.5, -2.0, 0.5), 0.01)
        print(dim, n, order, dm.n_p2, ctx.smoother_info(), flush=True)
        ctx.close()
