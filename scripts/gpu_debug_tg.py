import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import test_reference_style_solvers as T
import fem_oracle as fo
p = T.TaylorGreenVortex()
p._n_max_steps = 2
p.solve_problem()
s = p._get_solver()
dm = s._dofmap
u = s.solution.split()[0].nodal_values()
X = dm.p2_coords
g = 2*np.pi
for t in (0.0, 0.1, 0.2):
    ue = np.exp(-2*g*g*t/100.0) * np.stack([np.cos(g*X[:,0])*np.sin(g*X[:,1]), -np.sin(g*X[:,0])*np.cos(g*X[:,1])], axis=1)
    print("t", t, "max|u|", np.abs(u).max(), "max|ue|", np.abs(ue).max(), "max err", np.abs(u-ue).max())
print("newton", s.last_step_info.newton_iterations, [s.last_step_info.newton_residuals[i] for i in range(4)])
import _native as nat
for slot,name in ((nat.U0,"U0"),(nat.U1,"U1"),(nat.U2,"U2")):
    v = s._ctx.get_state(slot).reshape(-1,2); print(name, np.abs(v).max())
