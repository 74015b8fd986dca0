"""Host time-stepping logic against golden vectors captured from the reference's own
modules (tests/golden/gen_time_stepping_golden.py) and the tables written in the
reference's tests/test_bdf_time_stepping.py:67-114.  Exact equality, as in the reference."""
import json
import os

import numpy as np
import pytest

from bdf_time_stepping import BDFTimeStepping
from discrete_time import DiscreteTime, calculate_next_time

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bdf_tables.json")))


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: "order%d_%dsteps" % (c["order"], len(c["sizes"])))
def test_trajectory_matches_reference(case):
    ts = BDFTimeStepping(case["start"], case["end"], order=case["order"],
                         desired_start_time_step=case["first_step"])
    for row, size in zip(case["rows"], case["sizes"]):
        ts.set_desired_next_step_size(size)
        ts.update_coefficients()
        assert ts.step_number == row["step"]
        assert ts.current_time == row["current"] and ts.next_time == row["next"]
        assert ts.previous_time == row["previous"]
        assert ts.get_next_step_size() == row["next_step"]
        assert list(ts.coefficients(1)) == row["alpha1"]
        assert list(ts.coefficients(2)) == row["alpha2"]
        assert ts.coefficients_changed(1) == row["changed1"]
        assert ts.coefficients_changed(2) == row["changed2"]
        assert str(ts) == row["text"]
        ts.advance_time()
    assert ts.is_at_end() == case["at_end"]


@pytest.mark.parametrize("order", [1, 2])
def test_reference_golden_tables_and_restart(order):
    t = GOLD["reference_test_table"]
    ts = BDFTimeStepping(0.0, 9.0, order=order)
    for sweep in range(2):
        while not ts.is_at_end():
            i = ts.step_number
            ts.set_desired_next_step_size(t["step_sizes"][i])
            ts.update_coefficients()
            if order == 2:
                assert list(ts.coefficients(1)) == t["order2_alpha1"][i]
                assert list(ts.coefficients(2)) == t["order2_alpha2"][i]
            else:
                assert list(ts.coefficients(1)) == [1.0, -1.0]
                assert list(ts.coefficients(2)) == t["order1_alpha2"][i]
            assert ts.coefficients_changed(1) == t["order%d_changed1" % order][i]
            assert ts.coefficients_changed(2) == t["order%d_changed2" % order][i]
            ts.advance_time()
        assert ts.is_at_end()
        ts.restart()
    assert ts.n_levels() == order and ts.n_levels(2) == order + 1 and ts.n_substeps == 1


def test_discrete_time_random_walk_reaches_end():
    # mirrors the reference's tests/test_discrete_time.py (random steps, asserts is_at_end)
    rng = np.random.default_rng(7)
    ts = DiscreteTime(0.0, 5.0)
    assert ts.is_at_start() and ts.next_time == 0.0
    for _ in range(2):
        while not ts.is_at_end():
            ts.set_desired_next_step_size(float(rng.random()) + 1e-3)
            ts.advance_time()
        assert ts.current_time == 5.0
        ts.restart()
        assert ts.is_at_start()
    while not ts.is_at_end():
        ts.set_desired_next_step_size(0.7)
        ts.advance_time()
    ts.set_end_time(10.0)
    while not ts.is_at_end():
        ts.set_desired_next_step_size(float(rng.random()) + 1e-3)
        ts.advance_time()
    assert ts.current_time == 10.0


def test_end_snapping():
    assert calculate_next_time(0.0, 0.5, 1.0) == 0.5
    assert calculate_next_time(0.5, 0.49, 1.0) == 1.0      # remainder < 5 % of the step
    assert calculate_next_time(0.5, 0.6, 1.0) == 1.0
