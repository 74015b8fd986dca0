"""Scratch GPU check: device operators and IPCS steps vs the CPU oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
import _native as nat
from fem_mesh import rectangle_mesh, TaylorHoodDofMap, FacetMarkers
import fem_oracle as fo

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = rectangle_mesh((0, 0), (1, 1), n, n)
dm = TaylorHoodDofMap(m)
t0 = time.time()
ctx = nat.NsfemContext(m.coords, m.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1)
print("create %.2fs  n_p2=%d n_p1=%d" % (time.time() - t0, dm.n_p2, dm.n_p1))
s = fo.Space(m.coords, m.cells, dm.p2_dofmap, dm.p1_dofmap)

def cmp(name, A, B):
    d = abs(A - B).max() if A.nnz or B.nnz else 0.0
    print("%-12s max|diff| %.3e  (max|ref| %.3e)" % (name, d, abs(B).max()))
    return d

cmp("mass_p2", ctx.operator_csr(nat.OP_MASS_P2), s.mass_p2())
cmp("stiff_p2", ctx.operator_csr(nat.OP_STIFF_P2), s.stiffness_p2())
cmp("stiff_p1", ctx.operator_csr(nat.OP_STIFF_P1), s.stiffness_p1())
cmp("mass_p1", ctx.operator_csr(nat.OP_MASS_P1), s.mass_p1())
cmp("div", ctx.operator_csr(nat.OP_DIV), s.divergence())
cmp("grad", ctx.operator_csr(nat.OP_GRAD), s.pressure_gradient())
cmp("divT", ctx.operator_csr(nat.OP_DIVT), s.divergence().T.tocsr())
rng = np.random.default_rng(0)
x = rng.standard_normal(2 * dm.n_p2)
y = ctx.operator_apply(nat.OP_DIV, x)
print("spmv div   ", abs(y - s.divergence() @ x).max())
xp = rng.standard_normal(dm.n_p1)
print("spmv grad  ", abs(ctx.operator_apply(nat.OP_GRAD, xp) - s.pressure_gradient() @ xp).max())
print("spmv stiff1", abs(ctx.operator_apply(nat.OP_STIFF_P1, xp) - s.stiffness_p1() @ xp).max())

# ---- cavity IPCS steps
marks = FacetMarkers(m)
near = lambda v: (lambda X: np.abs(X - v) < 1e-12)
marks.mark(lambda X: near(0.0)(X[:, 0]), 1); marks.mark(lambda X: near(1.0)(X[:, 0]), 2)
marks.mark(lambda X: near(0.0)(X[:, 1]), 3); marks.mark(lambda X: near(1.0)(X[:, 1]), 4)
dofs, vals = [], []
for mid, val in ((1, (0., 0.)), (2, (0., 0.)), (3, (0., 0.)), (4, (1., 0.))):
    nodes = np.unique(dm.facet_p2_nodes(marks.facets_with_id(mid)))
    for a in range(2):
        dofs.append(2 * nodes + a); vals.append(np.full(nodes.size, val[a]))
dofs = np.concatenate(dofs); vals = np.concatenate(vals)
last = {}
for d, v in zip(dofs, vals): last[int(d)] = v
bd = np.array(sorted(last)); bv = np.array([last[d] for d in bd])
Re, k = 100.0, 0.01
coef = dict(convective_term=1.0, pressure_term=1.0, viscous_term=1.0 / Re, body_force_term=None)
orc = fo.IPCSOracle(s, coef, refactor_every_step=False)
ctx.set_coeffs(1.0, 1.0, 1.0 / Re)
ctx.set_dirichlet(nat.VELOCITY, bd, bv)
ctx.set_dirichlet(nat.PRESSURE, np.zeros(0, np.int32), np.zeros(0))
for step in range(3):
    alpha = fo.bdf_alpha(step, 1.0)
    ctx.set_bdf(alpha, k)
    t0 = time.time(); info = ctx.step_ipcs(); tg = time.time() - t0
    t0 = time.time(); orc.step(alpha, k, (bd, bv)); tc = time.time() - t0
    u = ctx.get_state(nat.U0); p = ctx.get_state(nat.P); us = ctx.get_state(nat.USTAR)
    pm = p - p.mean(); po = orc.p - orc.p.mean()
    print("step %d newton %d (oracle %d) kry mom %d poi %d cor %d | gpu %.3fs cpu %.3fs" % (
        step, info.newton_iterations, orc.newton_its[-1], info.krylov_iterations_momentum,
        info.krylov_iterations_poisson, info.krylov_iterations_correction, tg, tc))
    print("   newton res gpu", [float("%.3e" % info.newton_residuals[i]) for i in range(info.newton_iterations + 1)])
    print("   newton res cpu", [float("%.3e" % r) for r in orc.newton_history[-1]])
    print("   rel err u* %.3e  u %.3e  p %.3e" % (
        np.linalg.norm(us - orc.ustar) / np.linalg.norm(orc.ustar),
        np.linalg.norm(u - orc.vel[0]) / np.linalg.norm(orc.vel[0]),
        np.linalg.norm(pm - po) / np.linalg.norm(po)))
    ctx.advance(0); orc.advance()
ms, nb = ctx.time_spmv(nat.OP_MOMENTUM_JAC, 20)
print("spmv J: %.4f ms, %.1f MB, %.1f GB/s" % (ms, nb / 1e6, nb / ms / 1e6))
