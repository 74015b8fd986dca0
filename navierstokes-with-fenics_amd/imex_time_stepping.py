"""Second-order implicit-explicit (IMEX) multistep coefficients on a variable step grid.

Same public surface and numbers as the reference's ``source/imex_time_stepping.py`` (:11-159).
For a scheme parameter pair (a, b) and the step ratio w = k_{n+1} / k_n the class provides

  alpha  weights of u^{n+1}, u^n, u^{n-1} in the discrete time derivative,
  beta   extrapolation weights of the explicit (convective) term at levels n, n-1,
  gamma  weights of the implicit (diffusive) term at levels n+1, n, n-1,
  eta    linear Taylor extrapolation to t_{n+1} from levels n, n-1,

with (a, b) = (1, 0) SBDF2, (1/2, 0) CNAB, (1/2, 1/8) modified CNAB, (0, 1) CNLF.  The very first
step is first order (implicit Euler / explicit convection).  No solver of this repository uses
the class yet (neither does the reference); it is pinned by golden trajectories produced by the
reference's own module (tests/golden/imex_theta_tables.json).
"""
import math
from enum import Enum, auto

from discrete_time import DiscreteTime


class IMEXType(Enum):
    CNAB = auto()
    mCNAB = auto()
    CNLF = auto()
    SBDF2 = auto()


_PARAMETERS = {IMEXType.SBDF2: (1.0, 0.0), IMEXType.CNAB: (0.5, 0.0),
               IMEXType.mCNAB: (0.5, 1.0 / 8.0), IMEXType.CNLF: (0.0, 1.0)}


class IMEXTimeStepping(DiscreteTime):
    def __init__(self, start_time, end_time, imex_type, desired_start_time_step=0.0):
        super().__init__(start_time, end_time, desired_start_time_step)
        assert isinstance(imex_type, IMEXType)
        self._type = imex_type
        self._first_order_state()

    def _first_order_state(self):
        self._imex_parameters = _PARAMETERS[self._type]
        self._coefficients_changed = True
        self._omega = -1.0                       # no ratio seen yet
        self._alpha = [1.0, -1.0, 0.0]
        self._beta = [1.0, 0.0]
        self._gamma = [1.0, 0.0, 0.0]
        self._eta = [1.0, 0.0]

    def restart(self):
        super().restart()
        self._first_order_state()

    def update_coefficients(self):
        if self._step_number == 0:               # the first step keeps the first-order scheme
            return
        w = self.get_next_step_size() / self.get_previous_step_size()
        assert math.isfinite(w) and w > 0.0
        if w == self._omega and self._step_number > 1:
            self._coefficients_changed = False
            return
        self._omega = w
        a, b = self._imex_parameters
        self._alpha[:] = [(1.0 + 2.0 * a * w) / (1.0 + w),
                          (1.0 - 2.0 * a) * w - 1.0,
                          (2.0 * a - 1.0) * w * w / (1.0 + w)]
        self._beta[:] = [1.0 + a * w, -a * w]
        self._gamma[:] = [a + b / (2.0 * w),
                          1.0 - a - (1.0 + 1.0 / w) * b / 2.0,
                          b / 2.0]
        self._eta[:] = [1.0 + w, -w]
        self._coefficients_changed = True

    def print_coefficients(self):
        """table of the four coefficient sets over the time levels n+1, n, n-1"""
        cell = "{:12.2e}".format
        blank = 12 * " "
        rows = [("coefficient", "n + 1", "n", "n - 1"),
                ("alpha", ) + tuple(cell(v) for v in self._alpha),
                ("beta", blank) + tuple(cell(v) for v in self._beta),
                ("gamma", ) + tuple(cell(v) for v in self._gamma),
                ("eta", blank) + tuple(cell(v) for v in self._eta)]
        print("+" + "+".join(4 * (14 * "-", )) + "+")
        for row in rows:
            print("| " + " | ".join("{:12}".format(c) for c in row) + " |")

    alpha = property(lambda self: self._alpha)
    beta = property(lambda self: self._beta)
    gamma = property(lambda self: self._gamma)
    eta = property(lambda self: self._eta)
    coefficients_changed = property(lambda self: self._coefficients_changed)

    @property
    def n_levels(self):
        return len(self._alpha) - 1

    @property
    def n_substeps(self):
        return 1
