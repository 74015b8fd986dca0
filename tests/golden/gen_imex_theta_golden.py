"""Generates tests/golden/imex_theta_tables.json by IMPORTING the reference's pure-Python
modules source/imex_time_stepping.py and source/theta_time_stepping.py (no dolfin import) in
the build container; only the JSON travels to the GPU box.  Inputs: scheme type, start/end time,
sequence of desired step sizes.  Outputs per step: the coefficient sets after
``update_coefficients()``."""
import json
import os
import random
import sys

sys.path.insert(0, "/root/reference/source")
from imex_time_stepping import IMEXTimeStepping, IMEXType  # noqa: E402  (reference modules)
from theta_time_stepping import GeneralThetaTimeStepping, ThetaTimeSteppingType  # noqa: E402


def imex_case(kind, start, end, first, sizes):
    ts = IMEXTimeStepping(start, end, IMEXType[kind], desired_start_time_step=first)
    rows = []
    for k in sizes:
        if ts.is_at_end():
            break
        ts.set_desired_next_step_size(k)
        ts.update_coefficients()
        rows.append(dict(step=ts.step_number, next_step=ts.get_next_step_size(), alpha=list(ts.alpha),
                         beta=list(ts.beta), gamma=list(ts.gamma), eta=list(ts.eta),
                         changed=ts.coefficients_changed))
        ts.advance_time()
    return dict(kind=kind, start=start, end=end, first_step=first, sizes=sizes, rows=rows,
                n_levels=ts.n_levels, n_substeps=ts.n_substeps)


def theta_case(kind, start, end, first, sizes):
    ts = GeneralThetaTimeStepping(start, end, ThetaTimeSteppingType[kind], desired_start_time_step=first)
    rows = []
    for k in sizes:
        if ts.is_at_end():
            break
        ts.set_desired_next_step_size(k)
        ts.update_coefficients()
        rows.append(dict(step=ts.step_number, steps=list(ts.intermediate_timesteps),
                         times=[list(r) for r in ts.intermediate_times]))
        ts.advance_time()
    return dict(kind=kind, start=start, end=end, first_step=first, sizes=sizes, rows=rows,
                theta=[list(t) for t in ts.theta], n_levels=ts.n_levels, n_steps=ts.n_steps)


random.seed(7)
var = [round(random.uniform(0.05, 0.6), 3) for _ in range(14)]
imex = [imex_case(k, 0.0, 6.0, f, s) for k in ("CNAB", "mCNAB", "CNLF", "SBDF2")
        for f, s in ((0.0, [1.0, 1.0, 2.0, 2.0, 1.0, 1.0]), (0.1, var))]
theta = [theta_case(k, 0.0, 3.0, f, s) for k in ("ForwardEuler", "BackwardEuler", "CrankNicolson",
                                                 "FractionalStep01", "FractionalStep02")
         for f, s in ((0.25, [0.25] * 5), (0.1, var[:8]))]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "imex_theta_tables.json")
with open(out, "w") as fh:
    json.dump(dict(imex=imex, theta=theta), fh, indent=1)
print("wrote", out, len(imex), len(theta))
