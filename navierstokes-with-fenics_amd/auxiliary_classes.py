"""Dimensionless numbers -> equation coefficients.

Restates ``EquationCoefficientHandler`` of the reference
(source/auxiliary_classes.py:167-406; arithmetic at :251-306): convective 1,
pressure 1, viscous 1/Re (or Ek/Ro, Ek, 1 in rotating frames), body force 1/Fr^2,
Coriolis/Euler 1/Ro, 1/(Ek Re) or 1.  Pure host arithmetic.
"""
import math


class EquationCoefficientHandler:
    _names = (("Re", "Reynolds"), ("Fr", "Froude"), ("Ro", "Rossby"), ("Ek", "Ekman"))

    def __init__(self, **kwargs):
        self._dimensionless_numbers = dict()
        for key, long_name in self._names:
            assert not (key in kwargs and long_name in kwargs)
            value = kwargs.get(key, kwargs.get(long_name))
            if value is not None:
                self._store(key, value)
        self._closed = False

    def _store(self, key, value):
        assert math.isfinite(value) and value > 0.0
        self._dimensionless_numbers[key] = value

    def _compute_equation_coefficients(self):
        n = self._dimensionless_numbers
        rotating = "Ro" in n or "Ek" in n
        if not rotating:
            if "Re" not in n:  # pragma: no cover
                raise RuntimeError()
            rotation, viscous = None, 1.0 / n["Re"]
        else:
            if all(k in n for k in ("Ek", "Re", "Ro")):  # pragma: no cover
                raise RuntimeError("Overconstrained parameter set.")
            if "Ro" in n and "Re" in n:
                rotation, viscous = 1.0 / n["Ro"], 1.0 / n["Re"]
            elif "Ro" in n and "Ek" in n:
                rotation, viscous = 1.0 / n["Ro"], n["Ek"] / n["Ro"]
            elif "Ek" in n and "Re" in n:
                rotation, viscous = 1.0 / (n["Ek"] * n["Re"]), 1.0 / n["Re"]
            elif "Ek" in n:
                rotation, viscous = 1.0, n["Ek"]
            else:
                rotation, viscous = 1.0 / n["Ro"], 1.0
        self._equation_coefficients = dict(
            convective_term=1.0, coriolis_term=rotation, euler_term=rotation, pressure_term=1.0,
            viscous_term=viscous,
            body_force_term=(1.0 / n["Fr"] ** 2 if "Fr" in n else None))

    @property
    def equation_coefficients(self):
        self._compute_equation_coefficients()
        return self._equation_coefficients

    def close(self):
        self._closed = True

    def clear(self):
        self._closed = False
        self._dimensionless_numbers.clear()
        if hasattr(self, "_equation_coefficients"):
            self._equation_coefficients.clear()

    def modify_dimensionless_number(self, key, value):
        assert key in self._dimensionless_numbers
        assert isinstance(value, float)
        self._store(key, value)

    def get_file_suffix(self):
        assert len(self._dimensionless_numbers) > 0
        return "".join("_" + k + "{:1.3e}".format(v) for k, v in self._dimensionless_numbers.items())

    def __str__(self):
        lines = ["dimensionless numbers:"]
        lines += ["  {:4} = {:.3e}".format(k, v) for k, v in self._dimensionless_numbers.items()]
        # (as in the reference, :190: only what has been computed so far is shown -- printing an
        # empty handler must not raise)
        if getattr(self, "_equation_coefficients", None):
            lines.append("equation coefficients:")
            for k, v in self._equation_coefficients.items():
                lines.append("  {:16} = {}".format(k, "None" if v is None else "{:.3e}".format(v)))
        return "\n".join(lines)


def _number_property(key):
    def getter(self):
        return self._dimensionless_numbers.get(key)

    def setter(self, value):
        assert self._closed is False
        assert isinstance(value, float)
        self._store(key, value)
    return property(getter, setter)


for _key in ("Re", "Fr", "Ek", "Ro"):
    setattr(EquationCoefficientHandler, _key, _number_property(_key))


class FunctionTime:
    """Time-dependent value with an optional derivative (reference: source/auxiliary_classes.py:89-117)."""

    def __init__(self, value_size, current_time=0.0):
        assert isinstance(value_size, int) and value_size > 0
        assert isinstance(current_time, float)
        self._value_size = value_size
        self._current_time = 0.0

    def derivative(self):  # pragma: no cover
        raise NotImplementedError("You are calling a purely virtual method.")

    def set_time(self, current_time):
        assert isinstance(current_time, float)
        assert current_time >= self._current_time
        self._current_time = current_time

    def value(self):  # pragma: no cover
        raise NotImplementedError("You are calling a purely virtual method.")

    @property
    def value_size(self):
        return self._value_size


class AngularVelocityVector:
    """Angular velocity of the rotating frame of reference (reference:
    source/auxiliary_classes.py:12-86).  ``value`` / ``derivative`` are plain numbers here (the
    reference wraps them in dolfin Constants that enter the forms): a float about e_z in 2D, a
    3-tuple in 3D; the solver pushes them to the device before every solve (C ABI
    nsfem_set_angular_velocity / nsfem_set_angular_velocity_3d)."""

    def __init__(self, space_dim=2, function=None):
        assert isinstance(space_dim, int) and space_dim in (2, 3)
        self._space_dim = space_dim
        self._current_time = 0.0
        self._value_size = 1 if space_dim == 2 else 3
        if function is not None:
            self.set_angular_velocity_function(function)

    def _convert(self, value):
        if self._space_dim == 2:
            return float(value)
        value = tuple(float(v) for v in value)
        assert len(value) == 3
        return value

    def _modify_time(self):
        self._omega = self._convert(self._angular_velocity.value())
        if self._alpha is not None:
            self._alpha = self._convert(self._angular_velocity.derivative())

    def set_angular_velocity_function(self, function):
        assert isinstance(function, FunctionTime)
        assert function.value_size == self._value_size
        self._angular_velocity = function
        self._omega = self._convert(function.value())
        try:                                   # the derivative is optional (:37-50)
            self._alpha = self._convert(function.derivative())
        except (RuntimeError, NotImplementedError):
            self._alpha = None

    def set_time(self, current_time):
        assert isinstance(current_time, float)
        assert current_time >= self._current_time
        self._current_time = current_time
        self._angular_velocity.set_time(self._current_time)
        self._modify_time()

    @property
    def derivative(self):
        return self._alpha

    @property
    def space_dim(self):
        return self._space_dim

    @property
    def value(self):
        return self._omega
