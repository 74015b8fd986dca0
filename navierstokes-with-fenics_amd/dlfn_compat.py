"""Minimal dolfin-free stand-ins for the objects that cross the solver boundary.

The reference's demos/tests hand ``dolfin.Expression`` / ``Constant`` /
``SubDomain`` instances to the solver (e.g. tests/test_ipcs_solver.py:37-43,
tests/test_transient_solvers.py:20-46).  FEniCS is not available, so these small
classes provide the slice of that API the solver surface consumes: evaluation at
dof coordinates, ``.t`` / ``.time`` attributes, ``value_rank``.  C++ expression
strings are evaluated with numpy (``x[i]`` is the i-th coordinate array).
"""
import math

import numpy as np

pi = math.pi
DOLFIN_EPS = 3.0e-16
_LOG_LEVEL = [20]


def set_log_level(level):
    _LOG_LEVEL[0] = int(level)


def info(message):
    if _LOG_LEVEL[0] <= 20:
        print(message)


def near(a, b, eps=DOLFIN_EPS):
    return np.abs(np.asarray(a) - b) < eps


class Constant:
    def __init__(self, value, name=None):
        self._v = np.atleast_1d(np.asarray(value, dtype=np.float64)).copy()
        self._scalar = np.ndim(value) == 0
        self._name = name

    def assign(self, value):
        self._v[...] = np.asarray(float(value) if self._scalar else value, dtype=np.float64)

    def values(self):
        return self._v.copy()

    def value_rank(self):
        return 0 if self._scalar else 1

    @property
    def ufl_shape(self):
        return () if self._scalar else (self._v.size,)

    def rename(self, name, label):
        self._name = name

    def __float__(self):
        assert self._scalar
        return float(self._v[0])

    def eval_at(self, X):
        X = np.atleast_2d(X)
        if self._scalar:
            return np.full(X.shape[0], self._v[0])
        return np.tile(self._v, (X.shape[0], 1))


_NAMESPACE = dict(where=np.where, sin=np.sin, cos=np.cos, tan=np.tan, exp=np.exp, log=np.log, sqrt=np.sqrt,
                  pow=np.power, fabs=np.abs, abs=np.abs, tanh=np.tanh, sinh=np.sinh, cosh=np.cosh,
                  atan2=np.arctan2, atan=np.arctan, asin=np.arcsin, acos=np.arccos,
                  fmin=np.minimum, fmax=np.maximum, M_PI=math.pi, pi=math.pi, DOLFIN_PI=math.pi,
                  DOLFIN_EPS=DOLFIN_EPS)


def _cpp_to_python(code):
    """C++ expression -> numpy-evaluable Python: ``std::`` dropped, ``&&``/``||``/``!`` mapped to
    elementwise logic, and the conditional operator ``c ? a : b`` (right-associative, lowest
    precedence) rewritten as ``where(c, a, b)`` inside every parenthesis level."""
    code = code.replace("std::", "").replace("&&", " & ").replace("||", " | ")

    def ternary(expr):
        depth, q = 0, -1
        for i, ch in enumerate(expr):
            depth += ch in "([" 
            depth -= ch in ")]"
            if ch == "?" and depth == 0:
                q = i
                break
        if q < 0:
            return expr
        depth, nested = 0, 0
        for j in range(q + 1, len(expr)):
            ch = expr[j]
            depth += ch in "(["
            depth -= ch in ")]"
            if depth == 0 and ch == "?":
                nested += 1
            elif depth == 0 and ch == ":":
                if nested == 0:
                    return "where(%s, %s, %s)" % (expr[:q], ternary(expr[q + 1:j]), ternary(expr[j + 1:]))
                nested -= 1
        raise SyntaxError("unbalanced conditional operator in expression: " + expr)

    def walk(expr):                      # rewrite innermost parentheses first
        out, i = "", 0
        while i < len(expr):
            if expr[i] == "(":
                depth, j = 1, i + 1
                while depth:
                    depth += expr[j] == "("
                    depth -= expr[j] == ")"
                    j += 1
                inner = expr[i + 1:j - 1]
                parts, d, start = [], 0, 0          # split arguments at top-level commas
                for k, ch in enumerate(inner):
                    d += ch in "(["
                    d -= ch in ")]"
                    if ch == "," and d == 0:
                        parts.append(inner[start:k])
                        start = k + 1
                parts.append(inner[start:])
                out += "(" + ",".join(ternary(walk(a)) for a in parts) + ")"
                i = j
            else:
                out += expr[i]
                i += 1
        return out

    return ternary(walk(code)).strip()


class Expression:
    """``Expression("cpp string" | (strings...), degree=k, **parameters)``.

    Parameters become attributes and may be re-assigned (``expr.t = 0.3``), which is
    what ``InstationarySolverBase._set_time`` relies on (reference:
    source/ns_solver_base.py:1045-1051)."""

    def __init__(self, cpp_code, degree=None, element=None, **params):
        object.__setattr__(self, "_params", dict(params))
        self._code = (cpp_code,) if isinstance(cpp_code, str) else tuple(cpp_code)
        self._scalar = isinstance(cpp_code, str)
        self._degree = degree
        self._compiled = [compile(_cpp_to_python(c), "<expression>", "eval") for c in self._code]
        self._name = None

    def __getattr__(self, key):
        params = object.__getattribute__(self, "_params")
        if key in params:
            return params[key]
        raise AttributeError(key)

    def __setattr__(self, key, value):
        if key in self._params:
            self._params[key] = value
        else:
            object.__setattr__(self, key, value)

    def value_rank(self):
        return 0 if self._scalar else 1

    def value_dimension(self, i=0):
        return len(self._code)

    def rename(self, name, label):
        self._name = name

    def eval_at(self, X):
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        ns = dict(_NAMESPACE)
        ns.update(self._params)
        ns["x"] = X.T
        cols = [np.broadcast_to(np.asarray(eval(c, {"__builtins__": {}}, ns), dtype=np.float64),
                                (X.shape[0],)) for c in self._compiled]
        return cols[0].copy() if self._scalar else np.stack(cols, axis=1)


class UserExpression:
    """Python-callable expression: subclass and implement ``eval(values, x)`` and
    ``value_shape()`` as in dolfin."""

    def __init__(self, degree=None, **kwargs):
        self._degree = degree

    def value_shape(self):
        return ()

    def value_rank(self):
        return len(self.value_shape())

    def rename(self, name, label):
        pass

    def eval_at(self, X):
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        shape = self.value_shape()
        n = shape[0] if shape else 1
        out = np.zeros((X.shape[0], n))
        for i in range(X.shape[0]):
            self.eval(out[i], X[i])
        return out[:, 0] if not shape else out


class SubDomain:
    """Base class for periodic maps / boundary predicates (``inside``, ``map``)."""

    def inside(self, x, on_boundary):  # pragma: no cover
        raise NotImplementedError

    def map(self, x_slave, x_master):  # pragma: no cover
        raise NotImplementedError


def evaluate(value, X):
    """Evaluate a Constant / Expression / python callable / plain number(s) at X."""
    X = np.atleast_2d(X)
    if hasattr(value, "eval_at"):
        return value.eval_at(X)
    if callable(value):
        return np.asarray(value(X), dtype=np.float64)
    arr = np.asarray(value, dtype=np.float64)
    if arr.ndim == 0:
        return np.full(X.shape[0], float(arr))
    return np.tile(arr, (X.shape[0], 1))


def __getattr__(name):
    # dolfin.Function / FunctionSpace / project stand-ins live in fem_spaces (imported lazily:
    # fem_spaces itself evaluates values through this module)
    if name in ("Function", "FunctionSpace", "project"):
        import fem_spaces
        return getattr(fem_spaces, name)
    raise AttributeError(name)


def is_time_dependent(value):
    return hasattr(value, "_params") and any(k in value._params for k in ("t", "time"))
