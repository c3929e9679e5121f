"""A small ``dolfin`` look-alike for case files (SURVEY §8f "next" row 4).

The reference's case files define boundaries and actuator profiles as **C++ expression strings**
that dolfin JIT-compiles (``CompiledSubDomain("on_boundary && near(x[0], xinfa, MESH_TOL)", xinfa=…)``,
``Expression(["0", "(x[0]-x0>=L || …) ? 0 : u_ctrl*…"], element=…, L=…, u_ctrl=0.0)``).
This module evaluates the grammar those strings actually use with numpy, vectorised over points,
so a reference case file ports by swapping ``import dolfin`` for
``from flowcontrol_amd import dolfin_compat as dolfin``:

    x[0] x[1] on_boundary pi DOLFIN_EPS  numbers identifiers(parameters)
    + - * / unary-  < <= > >= == !=  && || !  ?:  ( )
    near(a, b[, eps]) between(a, lo, hi)?  sin cos tan atan2 exp log sqrt pow fabs abs tanh

Provided: ``CompiledSubDomain``, ``Expression``, ``Constant``, ``DirichletBC``, ``DOLFIN_EPS``, ``pi``,
and the string helpers of the reference's ``utils/fem.py:53-70`` (``near_cpp`` …).
"""

from __future__ import annotations

import re

import numpy as np

from .actuator import ActuatorExpression
from .fem.boundary import DOLFIN_EPS, Constant, DirichletBC, SubDomain

pi = np.pi

_TOKEN = re.compile(
    r"\s*(?:(?P<num>(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?)|(?P<id>[A-Za-z_][A-Za-z_0-9]*)|(?P<op>&&|\|\||<=|>=|==|!=|[-+*/<>!?:(),\[\]]))"
)

_FUNCS = {
    "sin": np.sin, "cos": np.cos, "tan": np.tan, "atan2": np.arctan2, "exp": np.exp, "log": np.log, "sqrt": np.sqrt,
    "pow": np.power, "fabs": np.abs, "abs": np.abs, "tanh": np.tanh,
    "near": lambda a, b, eps=DOLFIN_EPS: (a >= b - eps) & (a <= b + eps),
}


class _Parser:
    """Recursive descent over C operator precedence; produces a closure ``f(env) -> ndarray``."""

    def __init__(self, text: str):
        self.toks = []
        pos = 0
        text = text.strip()
        while pos < len(text):
            m = _TOKEN.match(text, pos)
            if not m or m.end() == pos:
                raise ValueError(f"cannot tokenise C expression at: {text[pos:pos + 20]!r}")
            pos = m.end()
            kind = m.lastgroup
            self.toks.append((kind, m.group(kind)))
        self.i = 0

    def peek(self):
        return self.toks[self.i] if self.i < len(self.toks) else (None, None)

    def take(self, val=None):
        k, v = self.peek()
        if val is not None and v != val:
            raise ValueError(f"expected {val!r}, got {v!r}")
        self.i += 1
        return k, v

    def parse(self):
        f = self.ternary()
        if self.i != len(self.toks):
            raise ValueError(f"unexpected token {self.peek()[1]!r}")
        return f

    def ternary(self):
        c = self.lor()
        if self.peek()[1] == "?":
            self.take("?")
            a = self.ternary()
            self.take(":")
            b = self.ternary()
            return lambda e: np.where(np.asarray(c(e), dtype=bool), a(e), b(e))
        return c

    def _binary(self, sub, table):
        f = sub()
        while self.peek()[1] in table:
            op = table[self.take()[1]]
            g = sub()
            f = (lambda f, g, op: lambda e: op(f(e), g(e)))(f, g, op)
        return f

    def lor(self):
        return self._binary(self.land, {"||": lambda a, b: np.asarray(a, bool) | np.asarray(b, bool)})

    def land(self):
        return self._binary(self.equality, {"&&": lambda a, b: np.asarray(a, bool) & np.asarray(b, bool)})

    def equality(self):
        return self._binary(self.relational, {"==": np.equal, "!=": np.not_equal})

    def relational(self):
        return self._binary(self.additive, {"<": np.less, "<=": np.less_equal, ">": np.greater, ">=": np.greater_equal})

    def additive(self):
        return self._binary(self.multiplicative, {"+": np.add, "-": np.subtract})

    def multiplicative(self):
        return self._binary(self.unary, {"*": np.multiply, "/": np.divide})

    def unary(self):
        v = self.peek()[1]
        if v == "-":
            self.take()
            f = self.unary()
            return lambda e: -f(e)
        if v == "+":
            self.take()
            return self.unary()
        if v == "!":
            self.take()
            f = self.unary()
            return lambda e: ~np.asarray(f(e), bool)
        return self.primary()

    def primary(self):
        k, v = self.take()
        if k == "num":
            val = float(v)
            return lambda e: val
        if v == "(":
            f = self.ternary()
            self.take(")")
            return f
        if k == "id":
            if self.peek()[1] == "[":  # x[0], x[1]
                self.take("[")
                _, idx = self.take()
                self.take("]")
                j = int(float(idx))
                return lambda e: e[v][:, j]
            if self.peek()[1] == "(":
                self.take("(")
                args = []
                if self.peek()[1] != ")":
                    args.append(self.ternary())
                    while self.peek()[1] == ",":
                        self.take(",")
                        args.append(self.ternary())
                self.take(")")
                if v not in _FUNCS:
                    raise ValueError(f"unknown function {v!r} in C expression")
                fn = _FUNCS[v]
                return lambda e: fn(*[a(e) for a in args])
            return lambda e: e[v]
        raise ValueError(f"unexpected token {v!r}")


def compile_c_expression(text: str):
    """``f(env)`` evaluating a C expression string; ``env`` maps names (``x``, parameters…) to values."""
    return _Parser(str(text)).parse()


class CompiledSubDomain(SubDomain):
    """``dolfin.CompiledSubDomain(cpp_string, **parameters)``."""

    def __init__(self, code: str, **params):
        self.code = code
        self.params = dict(params)
        f = compile_c_expression(code)

        def inside(x, on_boundary):
            env = {"x": x, "on_boundary": on_boundary, "pi": np.pi, "DOLFIN_EPS": DOLFIN_EPS, "DOLFIN_PI": np.pi, **self.params}
            return np.broadcast_to(np.asarray(f(env), dtype=bool), (x.shape[0],))

        super().__init__(inside, name=code)


class Expression(ActuatorExpression):
    """``dolfin.Expression([cpp, cpp], element=…, **parameters)`` with mutable parameters.

    Actuator expressions are linear in ``u_ctrl``; ``profile`` evaluates the strings
    with ``u_ctrl = 1``."""

    def __init__(self, code, element=None, degree=None, **params):
        codes = [code] if isinstance(code, str) else list(code)
        self._fs = [compile_c_expression(c) for c in codes]
        self._param_names = list(params)
        u0 = params.pop("u_ctrl", 0.0)
        super().__init__(self._eval_unit, u_ctrl=u0, **params)

    def _env(self, x, u_ctrl):
        env = {"x": x, "pi": np.pi, "DOLFIN_EPS": DOLFIN_EPS, "u_ctrl": u_ctrl}
        for k in self._param_names:
            if k != "u_ctrl":
                env[k] = getattr(self, k)
        return env

    def _eval_unit(self, x):
        env = self._env(x, 1.0)
        return np.stack([np.broadcast_to(np.asarray(f(env), dtype=np.float64), (x.shape[0],)) for f in self._fs], axis=1)

    def __getitem__(self, i):
        return lambda x: self(x)[:, i]


# string helpers of the reference's utils/fem.py:53-70
def near_cpp(x: str, xnear, tol: str = "MESH_TOL") -> str:
    return f"near({x}, {xnear}, {tol})"


def between_cpp(x: str, xmin: str, xmax: str, tol: str = "0.0") -> str:
    return f"{x}>={xmin}-{tol} && {x}<={xmax}+{tol}"


def or_cpp() -> str:
    return " || "


def and_cpp() -> str:
    return " && "


def on_boundary_cpp() -> str:
    return "on_boundary"


__all__ = ["CompiledSubDomain", "Expression", "Constant", "DirichletBC", "DOLFIN_EPS", "pi", "compile_c_expression",
           "near_cpp", "between_cpp", "or_cpp", "and_cpp", "on_boundary_cpp"]
