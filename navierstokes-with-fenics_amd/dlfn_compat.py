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

from form_language import (FacetNormal, Measure, Operand, assemble, dot, ds, dx, grad, inner,  # noqa: F401
                           sqrt)

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


class Constant(Operand):
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


_NAMESPACE.update(logical_and=np.logical_and, logical_or=np.logical_or, logical_not=np.logical_not)


def _c_int_div(a, b):
    """C++ division of two integer operands truncates toward zero"""
    return int(a / b) if isinstance(a, (int, np.integer)) and isinstance(b, (int, np.integer)) else a / b


_NAMESPACE["c_int_div"] = _c_int_div

_BINARY = {"||": 1, "&&": 2, "==": 3, "!=": 3, "<": 4, ">": 4, "<=": 4, ">=": 4, "+": 5, "-": 5,
           "*": 6, "/": 6, "%": 6}


def _tokenize(code):
    import re
    token = re.compile(r"\s*(?:(\d+\.\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+|\d+)"
                       r"|([A-Za-z_][A-Za-z_0-9]*(?:::[A-Za-z_][A-Za-z_0-9]*)*)|(\|\||&&|==|!=|<=|>=|[-+*/%<>!?:(),\[\]]))")
    pos, out = 0, []
    code = code.strip()
    while pos < len(code):
        m = token.match(code, pos)
        if not m or m.end() == pos:
            raise SyntaxError("cannot parse the expression at: %r" % code[pos:pos + 20])
        num, name, op = m.groups()
        out.append(("num", num) if num else ("name", name.split("::")[-1]) if name else ("op", op))
        pos = m.end()
    return out


def _cpp_to_python(code):
    """C++ expression string -> numpy-evaluable Python source.  A small precedence parser (instead
    of textual replacement): ``a && b`` / ``a || b`` / ``!a`` become logical_and / logical_or /
    logical_not CALLS (Python's ``&`` / ``|`` bind tighter than comparisons), the conditional
    operator ``c ? a : b`` (right-associative, lowest precedence) becomes ``where(c, a, b)``, a
    quotient of two integer-valued operands truncates like C++ (``1/2*x[0]`` is 0), ``std::`` is
    dropped."""
    toks = _tokenize(code)
    pos = [0]

    def peek():
        return toks[pos[0]] if pos[0] < len(toks) else ("end", "")

    def take(kind=None, value=None):
        t = peek()
        if (kind and t[0] != kind) or (value and t[1] != value):
            raise SyntaxError("unexpected %r in expression %r" % (t[1], code))
        pos[0] += 1
        return t

    def primary():                               # -> (source, is_integer)
        kind, val = peek()
        if kind == "num":
            take()
            return val, val.isdigit()
        if kind == "name":
            take()
            src = val
            while peek() == ("op", "(") or peek() == ("op", "["):
                if peek()[1] == "(":
                    take()
                    args = []
                    if peek() != ("op", ")"):
                        args.append(ternary()[0])
                        while peek() == ("op", ","):
                            take()
                            args.append(ternary()[0])
                    take("op", ")")
                    src = "%s(%s)" % (src, ", ".join(args))
                else:
                    take()
                    idx = ternary()[0]
                    take("op", "]")
                    src = "%s[%s]" % (src, idx)
            return src, False
        if (kind, val) == ("op", "("):
            take()
            inner, is_int = ternary()
            take("op", ")")
            return "(%s)" % inner, is_int
        raise SyntaxError("unexpected %r in expression %r" % (val, code))

    def unary():
        kind, val = peek()
        if kind == "op" and val in "+-":
            take()
            operand, is_int = unary()
            return "(%s%s)" % (val, operand), is_int
        if (kind, val) == ("op", "!"):
            take()
            return "logical_not(%s)" % unary()[0], False
        return primary()

    def binary(min_prec):
        left, lint = unary()
        while peek()[0] == "op" and peek()[1] in _BINARY and _BINARY[peek()[1]] >= min_prec:
            op = take()[1]
            right, rint = binary(_BINARY[op] + 1)
            if op == "&&":
                left, lint = "logical_and(%s, %s)" % (left, right), False
            elif op == "||":
                left, lint = "logical_or(%s, %s)" % (left, right), False
            elif op == "/" and lint and rint:
                left, lint = "c_int_div(%s, %s)" % (left, right), True
            else:
                left, lint = "(%s %s %s)" % (left, op, right), lint and rint and op in "+-*%"
        return left, lint

    def ternary():
        cond, cint = binary(1)
        if peek() == ("op", "?"):
            take()
            a, _ = ternary()
            take("op", ":")
            b, _ = ternary()
            return "where(%s, %s, %s)" % (cond, a, b), False
        return cond, cint

    src, _ = ternary()
    if pos[0] != len(toks):
        raise SyntaxError("unexpected %r in expression %r" % (peek()[1], code))
    return src


class Expression(Operand):
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

    @property
    def T(self):                         # a parameter called T wins over the transpose of the mix-in
        params = object.__getattribute__(self, "_params")
        return params["T"] if "T" in params else Operand.T.fget(self)

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


class UserExpression(Operand):
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


def FunctionSpace(*args, **kwargs):
    """``FunctionSpace(dofmap, kind)`` of fem_spaces, or dolfin's ``FunctionSpace(mesh, "CG", k)``
    (target of ``project`` in the reference's post-processing hooks)"""
    import fem_spaces
    import form_language
    if len(args) >= 2 and isinstance(args[1], str) and hasattr(args[0], "num_cells"):
        return form_language.LagrangeSpaceRequest(args[0], args[1], args[2] if len(args) > 2 else kwargs.get("degree", 1))
    return fem_spaces.FunctionSpace(*args, **kwargs)


def project(value, space, function=None):
    """``dolfin.project``: onto the solver's spaces (fem_spaces.project: constants and functions the
    space reproduces) or, for a form expression and ``FunctionSpace(mesh, "CG", 1)``, the L2
    projection with the mass solve on the device"""
    import fem_spaces
    import form_language
    if isinstance(space, form_language.LagrangeSpaceRequest):
        assert function is None
        return form_language.project_expression(value, space)
    return fem_spaces.project(value, space, function)


def __getattr__(name):
    # the dolfin.Function stand-in lives in fem_spaces (imported lazily: fem_spaces itself evaluates
    # values through this module)
    if name == "Function":
        import fem_spaces
        return fem_spaces.Function
    raise AttributeError(name)


def is_time_dependent(value):
    return hasattr(value, "_params") and any(k in value._params for k in ("t", "time"))
