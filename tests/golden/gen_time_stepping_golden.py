"""Generates tests/golden/bdf_tables.json by IMPORTING the reference's pure-Python
time stepping modules (source/discrete_time.py, source/bdf_time_stepping.py -- they
do not import dolfin) in the build container.  /root/reference does not exist on the
GPU box: only the JSON travels.  Inputs: start/end time, order, a sequence of desired
step sizes.  Outputs per step: times, step number, alpha(1), alpha(2), changed flags.
Also stores the golden alpha tables written in the reference's
tests/test_bdf_time_stepping.py:67-114 (data, not code)."""
import json
import os
import random
import sys

sys.path.insert(0, "/root/reference/source")
from bdf_time_stepping import BDFTimeStepping  # noqa: E402  (reference module)


def trajectory(start, end, order, first_step, sizes):
    ts = BDFTimeStepping(start, end, order=order, desired_start_time_step=first_step)
    rows = []
    i = 0
    while not ts.is_at_end() and i < len(sizes):
        ts.set_desired_next_step_size(sizes[i])
        ts.update_coefficients()
        rows.append(dict(step=ts.step_number, current=ts.current_time, next=ts.next_time,
                         previous=ts.previous_time, next_step=ts.get_next_step_size(),
                         alpha1=list(ts.coefficients(1)), alpha2=list(ts.coefficients(2)),
                         changed1=ts.coefficients_changed(1), changed2=ts.coefficients_changed(2),
                         text=str(ts)))
        ts.advance_time()
        i += 1
    return dict(start=start, end=end, order=order, first_step=first_step, sizes=sizes, rows=rows,
                at_end=ts.is_at_end())


random.seed(20261004)
cases = []
for order in (1, 2):
    cases.append(trajectory(0.0, 9.0, order, 0.0, [1.0, 1.0, 2.0, 2.0, 1.0, 1.0, 1.0]))
    cases.append(trajectory(0.0, 1.0, order, 0.01, [0.01] * 12))
    cases.append(trajectory(0.0, 5.0, order, 0.1, [round(random.uniform(0.05, 0.9), 3) for _ in range(25)]))
    cases.append(trajectory(0.5, 2.0, order, 0.25, [0.25, 0.25, 0.5, 0.125, 0.125, 0.125, 0.3, 0.3]))
# golden tables written in the reference test (order 2, step sizes [1,1,2,2,1,1,1])
table = {
    "step_sizes": [1.0, 1.0, 2.0, 2.0, 1.0, 1.0, 1.0],
    "order1_alpha2": [[1.0, -2.0, 1.0], [1.0, -2.0, 1.0], [4.0 / 3.0, -4.0, 8.0 / 3.0], [1.0, -2.0, 1.0],
                      [2.0 / 3.0, -1.0, 1.0 / 3.0], [1.0, -2.0, 1.0], [1.0, -2.0, 1.0]],
    "order2_alpha1": [[1.0, -1.0, 0.0], [1.5, -2.0, 0.5], [5.0 / 3.0, -3.0, 4.0 / 3.0], [1.5, -2.0, 0.5],
                      [4.0 / 3.0, -1.5, 1.0 / 6.0], [1.5, -2.0, 0.5], [1.5, -2.0, 0.5]],
    "order2_alpha2": [[1.0, -2.0, 1.0, 0.0], [2.0, -5.0, 4.0, -1.0], [3.0, -14.0, 16.0, -5.0],
                      [11.0 / 5.0, -6.0, 7.0, -16.0 / 5.0], [6.0 / 5.0, -2.0, 1.0, -1.0 / 5.0],
                      [7.0 / 4.0, -4.0, 5.0 / 2.0, -1.0 / 4.0], [2.0, -5.0, 4.0, -1.0]],
    "order2_changed1": [True, True, True, True, True, True, False],
    "order2_changed2": [True, True, True, True, True, True, True],
    "order1_changed1": [True, False, False, False, False, False, False],
    "order1_changed2": [True, True, True, True, True, True, False],
}
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bdf_tables.json")
with open(out, "w") as fh:
    json.dump(dict(cases=cases, reference_test_table=table), fh, indent=1)
print("wrote", out, len(cases), "cases")
