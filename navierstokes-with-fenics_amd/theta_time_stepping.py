"""One-step theta schemes and the fractional-step theta scheme (V. John 2016, Tables 7.1 / 7.2).

Same public surface and numbers as the reference's ``source/theta_time_stepping.py`` (:8-123):
``theta`` is a list of 4-tuples, one per sub-step, weighting (implicit diffusion, explicit
diffusion, explicit right-hand side, implicit right-hand side); ``intermediate_timesteps`` are the
sub-step sizes and ``intermediate_times[0 | 1]`` the start / end times of the sub-steps of the
coming step.  Fractional step: theta = 1 - sqrt(2)/2, sub-steps (theta, 1 - 2 theta, theta) k.
No solver uses the class yet (as in the reference); pinned by golden trajectories produced by the
reference's own module (tests/golden/imex_theta_tables.json).
"""
import math
from enum import Enum, auto

from discrete_time import DiscreteTime


class ThetaTimeSteppingType(Enum):
    ForwardEuler = auto()
    BackwardEuler = auto()
    CrankNicolson = auto()
    FractionalStep01 = auto()
    FractionalStep02 = auto()


class GeneralThetaTimeStepping(DiscreteTime):
    _theta = 1.0 - math.sqrt(2.0) / 2.0
    _zeta = 1.0 - 2.0 * _theta
    _tau = _zeta / (1.0 - _theta)
    _eta = 1.0 - _tau

    def __init__(self, start_time, end_time, theta_type, desired_start_time_step=0.0):
        super().__init__(start_time, end_time, desired_start_time_step)
        assert isinstance(theta_type, ThetaTimeSteppingType)
        self._type = theta_type
        T = ThetaTimeSteppingType
        th, ze, ta, et = self._theta, self._zeta, self._tau, self._eta
        one_step = {T.ForwardEuler: (0.0, 1.0, 1.0, 0.0), T.BackwardEuler: (1.0, 0.0, 0.0, 1.0),
                    T.CrankNicolson: (0.5, 0.5, 0.5, 0.5)}
        if theta_type in one_step:
            self._Theta = [one_step[theta_type]]
        elif theta_type is T.FractionalStep01:
            outer = (ta * th, et * th, et * th, ta * th)
            self._Theta = [outer, (et * ze, ta * ze, ta * ze, et * ze), outer]
        else:
            outer = (ta * th, et * th, th, 0.0)
            self._Theta = [outer, (et * ze, ta * ze, 0.0, ze), outer]
        self._n_steps = len(self._Theta)
        self._clear_intermediates()

    def _clear_intermediates(self):
        self._intermediate_timesteps = [0.0] * self._n_steps
        self._intermediate_times = [[0.0] * self._n_steps for _ in range(2)]

    def restart(self):
        super().restart()
        self._clear_intermediates()

    def update_coefficients(self):
        k = self.get_next_step_size()
        assert math.isfinite(k)
        t0, t1 = self.current_time, self.next_time
        if self._n_steps == 3:
            th = self._theta
            self._intermediate_timesteps[:] = [th * k, self._zeta * k, th * k]
            self._intermediate_times[0][:] = [t0, t0 + th * k, t1 - th * k]
            self._intermediate_times[1][:] = [t0 + th * k, t1 - th * k, t1]
        else:
            self._intermediate_timesteps[0] = k
            self._intermediate_times[0][0] = t0
            self._intermediate_times[1][0] = t1

    theta = property(lambda self: self._Theta)
    intermediate_timesteps = property(lambda self: self._intermediate_timesteps)
    intermediate_times = property(lambda self: self._intermediate_times)

    @property
    def n_levels(self):
        return 1

    @property
    def n_steps(self):
        return self._n_steps
