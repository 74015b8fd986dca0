"""Time-level bookkeeping of the transient solvers.

Same public surface and semantics as the reference's ``source/discrete_time.py``
(:5-184): the next time is snapped onto the end time when it comes within 5 % of
a step of it, ``advance_time`` repeats the last step size, ``restart`` rewinds.
Host-side only; nothing here touches the device.
"""

_END_SNAP_FRACTION = 0.05


def calculate_next_time(current_time, step_size, end_time):
    """t + dt, replaced by ``end_time`` when the remainder would be < 5 % of dt
    (reference: source/discrete_time.py:5-26)."""
    for value in (current_time, step_size, end_time):
        assert isinstance(value, float)
    assert step_size >= 0.0 and end_time >= current_time
    candidate = current_time + step_size
    return end_time if candidate > end_time - _END_SNAP_FRACTION * step_size else candidate


class DiscreteTime:
    """Current / next / previous time levels and the step counter."""

    def __init__(self, start_time, end_time, desired_start_time_step=0.0):
        for value in (start_time, end_time, desired_start_time_step):
            assert isinstance(value, float)
        assert start_time < end_time and desired_start_time_step >= 0.0
        self._start_time, self._end_time = start_time, end_time
        first = calculate_next_time(start_time, desired_start_time_step, end_time)
        self._start_step_size = first - start_time
        self._rewind()

    def _rewind(self):
        self._previous_time = self._current_time = self._start_time
        self._next_time = calculate_next_time(self._start_time, self._start_step_size,
                                              self._end_time)
        self._step_number = 0

    def __str__(self):
        return ("step number {0:8d}, current time {1:10.2e}, next step size {2:10.2e}"
                .format(self._step_number, self._current_time, self.get_next_step_size()))

    # read-only views -----------------------------------------------------------
    current_time = property(lambda self: self._current_time)
    next_time = property(lambda self: self._next_time)
    previous_time = property(lambda self: self._previous_time)
    start_time = property(lambda self: self._start_time)
    end_time = property(lambda self: self._end_time)
    step_number = property(lambda self: self._step_number)

    def is_at_start(self):
        return self._step_number == 0

    def is_at_end(self):
        return self._current_time == self._end_time

    def get_next_step_size(self):
        return self._next_time - self._current_time

    def get_previous_step_size(self):
        return self._current_time - self._previous_time

    # mutation --------------------------------------------------------------------
    def set_desired_next_step_size(self, next_step_size):
        assert isinstance(next_step_size, float) and next_step_size > 0.0
        self._next_time = calculate_next_time(self._current_time, next_step_size, self._end_time)

    def advance_time(self):
        assert self._next_time > self._current_time
        dt = self.get_next_step_size()
        self._previous_time, self._current_time = self._current_time, self._next_time
        self._step_number += 1
        self._next_time = calculate_next_time(self._current_time, dt, self._end_time)

    def restart(self):
        self._rewind()

    def set_end_time(self, new_end_time):
        assert isinstance(new_end_time, float)
        assert new_end_time > self._start_time and new_end_time > self._current_time
        self._end_time = new_end_time
        dt = self._start_step_size if self._step_number == 0 else self.get_previous_step_size()
        self._next_time = calculate_next_time(self._current_time, dt, self._end_time)
