"""Variable-step BDF-1/2 coefficients for first and second time derivatives.

Public surface and numbers of the reference's ``source/bdf_time_stepping.py``
(:7-186): ``coefficients(d)`` returns alpha with  d^d u/dt^d ~ sum_i alpha_i
u^{n+1-i} / k^d  for the step ratio  omega = k_{n+1}/k_n  (and the previous ratio
Omega for the second derivative); ``coefficients_changed(d)`` reports whether the
last ``update_coefficients`` call modified them.  The golden tables of the
reference's tests/test_bdf_time_stepping.py:67-114 are reproduced bit for bit
(tests/golden/bdf_tables.json).
"""
import math

from discrete_time import DiscreteTime


def _d1_weights(order, w):
    if order == 1:
        return [1.0, -1.0]
    return [(1.0 + 2.0 * w) / (1.0 + w), -(1.0 + w), w * w / (1.0 + w)]


def _d2_weights(order, w, W):
    if order == 1:
        return [2.0 * w / (1.0 + w), -2.0 * w, 2.0 * w * w / (1.0 + w)]
    s = 1.0 + W + w * W
    return [2.0 * w * (1.0 + (2.0 + 3.0 * w) * W) / ((1.0 + w) * s),
            -2.0 * w * (1.0 + 2.0 * (1.0 + w) * W) / (1.0 + W),
            2.0 * w ** 2 * (1.0 + W + 2.0 * w * W) / (1.0 + w),
            -2.0 * w ** 2 * (1.0 + 2.0 * w) * W ** 3 / ((1.0 + W) * s)]


class BDFTimeStepping(DiscreteTime):
    def __init__(self, start_time, end_time, order=2, desired_start_time_step=0.0):
        super().__init__(start_time, end_time, desired_start_time_step)
        assert isinstance(order, int) and order > 0
        if order > 2:  # pragma: no cover
            raise NotImplementedError()
        self._order = order
        self._reset_scheme()

    def _reset_scheme(self):
        # the very first step is always implicit Euler
        self._ratios = [1.0, 1.0]                 # (omega, Omega) of the last update
        self._changed = {1: True, 2: True}
        first = [1.0, -1.0] + [0.0] * (self._order - 1)
        second = [1.0, -2.0, 1.0] + [0.0] * (self._order - 1)
        self._alpha = {1: first, 2: second}

    def restart(self):
        super().restart()
        self._reset_scheme()

    def update_coefficients(self):
        if self.step_number == 0:
            return
        w = self.get_next_step_size() / self.get_previous_step_size()
        assert math.isfinite(w) and w > 0.0
        W = self._ratios[0]
        assert W > 0.0
        settled = self.step_number > 1
        same_w = self._ratios[0] == w
        same_W = self._ratios[1] == W
        if settled and same_w and (self._order == 1 or same_W):
            self._changed = {1: False, 2: False}
            return
        if self._order == 2 and settled and same_w:
            # only the older ratio moved: first-derivative weights are untouched
            self._ratios[1] = W
            self._changed[1] = False
        else:
            self._ratios = [w, W]
            self._alpha[1][:] = _d1_weights(self._order, w)
            self._changed[1] = self._order == 2
        self._alpha[2][:] = _d2_weights(self._order, w, W)
        self._changed[2] = True

    def coefficients(self, derivative):
        assert derivative in (1, 2)
        return tuple(self._alpha[derivative])

    def coefficients_changed(self, derivative):
        assert derivative in (1, 2)
        return self._changed[derivative]

    def n_levels(self, derivative=1):
        """Number of old time levels the derivative approximation reaches back."""
        assert derivative in (1, 2)
        return len(self._alpha[derivative]) - 1

    @property
    def n_substeps(self):
        return 1

    def print_coefficients(self):
        names = ("n + 1", "n", "n - 1", "n - 2")[: self._order + 2]
        width = len(names) + 1
        rule = "+-" + "-+-".join(width * (12 * "-",)) + "-+"
        print(rule)
        print("| {:12} | ".format("derivative") + " | ".join("{:12}".format(n) for n in names) + " |")
        for d, label in ((1, "1st"), (2, "2nd")):
            cells = ["{:12.2e}".format(a) for a in self._alpha[d]]
            cells += [12 * " "] * (len(names) - len(cells))
            print("| {:12} | ".format(label) + " | ".join(cells) + " |")
        print(rule)
