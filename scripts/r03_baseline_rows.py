"""BASELINE.md section 3, rows C1 / C1' on the GPU box: the reference-algorithm CPU figure (oracle: full re-assembly +
SuperLU per Newton iteration, one core) and the device figure through the solver classes, same mesh / dt / tolerances.
usage (gpurun): python scripts/r03_baseline_rows.py > gpurun_out/<tag>_baseline_rows.json"""
import contextlib, io, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "navierstokes-with-fenics_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import _native as nat
import fem_oracle as fo
from problem_specs import build_problem, unique_dirichlet

out = {"host_cpus": os.cpu_count()}
log = io.StringIO()
# ---- C1: cavity n = 64, Re = 100, BDF-2 monolithic, dt = 0.01, 10 steps
spec = dict(name="Cavity", mesh=("cube", 2, 64), scheme="bdf", numbers=dict(Re=100.0), clock=dict(dt=0.01, steps=3),
            start={"velocity": (0.0, 0.0), "pressure": 0.0},
            bcs=[("no_slip", "left"), ("no_slip", "right"), ("no_slip", "bottom"), ("velocity", "top", (1.0, 0.0))])
with contextlib.redirect_stdout(log):
    p = build_problem(dict(spec)); p.compute_cfl = False
    p.solve_problem()                         # set-up + 3 warm-up steps
    solver, ts = p._get_solver(), p._time_stepping
    solver._ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        ts.update_coefficients(); solver.solve(); ts.advance_time(); solver.advance_time()
    solver._ctx.synchronize(); dt_gpu = (time.perf_counter() - t0) / 10
    n_dofs = solver._n_dofs
    dm = solver._dofmap
    s = fo.Space(dm.mesh.coords, dm.mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    bd, bv = unique_dirichlet(*solver._dirichlet_bcs["velocity"])
    orc = fo.BDFOracle(s, solver._equation_coefficients)
    orc.step(fo.bdf_alpha(0, 1.0), 0.01, (bd, bv)); orc.advance()
    t0 = time.perf_counter(); n_cpu = 3
    for k in range(1, n_cpu + 1):
        orc.step(fo.bdf_alpha(k, 1.0), 0.01, (bd, bv)); orc.advance()
    dt_cpu = (time.perf_counter() - t0) / n_cpu
out["C1"] = {"config": "cavity n=64, Re=100, BDF-2 monolithic, dt=0.01", "n_dofs": n_dofs,
             "cpu_reference_algorithm_steps_per_s": 1.0 / dt_cpu, "cpu_dof_updates_per_s": n_dofs / dt_cpu, "cpu_steps_timed": n_cpu,
             "gpu_steps_per_s": 1.0 / dt_gpu, "gpu_dof_updates_per_s": n_dofs / dt_gpu, "gpu_steps_timed": 10,
             "gpu_settings": "solver-class defaults: Krylov rtol 1e-12, exact Newton"}
# ---- C1': demo/cavity_flow.py as shipped: stationary, n = 25, Re = 10, Picard -> Newton
spec = dict(name="Cavity", stationary=True, mesh=("cube", 2, 25), numbers=dict(Re=10.0),
            bcs=[("no_slip", "left"), ("no_slip", "right"), ("no_slip", "bottom"), ("velocity", "top", (1.0, 0.0))])
with contextlib.redirect_stdout(log):
    times = []
    for _ in range(3):
        p = build_problem(dict(spec))
        t0 = time.perf_counter(); p.solve_problem(); p._get_solver()._ctx.synchronize(); times.append(time.perf_counter() - t0)
out["C1prime"] = {"config": "demo/cavity_flow.py as shipped: stationary, n=25, Re=10, Picard -> Newton", "n_dofs": p._get_solver()._n_dofs,
                  "gpu_solves_per_s_including_set_up": 1.0 / min(times), "runs": times}
print(json.dumps(out))
