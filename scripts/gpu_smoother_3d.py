"""Scratch: finest-level smoother launch (timed between cache-flushing launches) on the 3D Kuhn mesh, non-periodic vs
triple-periodic dof numbering."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd")]
import numpy as np
import _native as nat
import dlfn_compat as dlfn
from fem_mesh import TaylorHoodDofMap, box_mesh, periodic_entity_map
from multigrid import attach_hierarchy
n = int(sys.argv[1])
class TP(dlfn.SubDomain):
    def inside(self, x, on_boundary):
        return bool(on_boundary and (dlfn.near(x[0], 0.0) or dlfn.near(x[1], 0.0) or dlfn.near(x[2], 0.0)))
    def map(self, xs, xm):
        for a in range(3):
            if dlfn.near(xs[a], 1.0):
                xm[:] = xs; xm[a] -= 1.0; return
        xm[:] = -10.0
mesh = box_mesh((0, 0, 0), (1, 1, 1), n, n, n)
for periodic in (False, True):
    dm = TaylorHoodDofMap(mesh, periodic_map=periodic_entity_map(mesh, TP()) if periodic else None)
    ctx = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1)
    attach_hierarchy(ctx, mesh, periodic=(TP(), dm.p1_vertex_node) if periodic else None)
    ctx.set_coeffs(1.0, 1.0, 0.01)
    ctx.set_bdf((1.5, -2.0, 0.5), 0.25 / n)
    ms, nb = ctx.time_spmv(nat.OP_MOMENTUM_SMOOTHER, 100)
    A = ctx.operator_csr(nat.OP_STIFF_P2)
    bw = np.abs(A.indices - np.repeat(np.arange(A.shape[0]), np.diff(A.indptr)))
    print("periodic %s: n_p2 %d nnz/row %.1f  smoother %.1f us  %.2f TB/s  (bytes %.0f MB)  col distance median %d p99 %d max %d" % (
        periodic, dm.n_p2, A.nnz / A.shape[0], ms * 1e3, nb / ms / 1e9, nb / 1e6, np.median(bw), np.percentile(bw, 99), bw.max()))
    ctx.close()
