"""
casadi_expr.py -- evaluates the expression strings CasADi printed into the reference's exported OCP
(acados_ocp_SNMPC.json: model.disc_dyn_expr / cost_y_expr / con_h_expr) numerically with numpy.

Used only by make_golden.py, here, to turn those exported expressions into input/output vectors (tests/golden/snmpc_expr.npz);
nothing of the reference travels. The printed form is fully parenthesised infix with `@k=` sub-expression definitions,
`(c?a:b)` selections, function calls, `{0}` output selectors, `'` transposes, slices and `(v)[i] = e` element assignments.
Matrix shapes are not printed (`reshape(A_pce)'`): they are supplied by the caller.
"""
import math
import re

import numpy as np

_TOK = re.compile(r"\s*(?:(\d+x\d+(?:,\d+nz)?)|(\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+)|(@\d+)|([A-Za-z_][A-Za-z_0-9]*)|(==|[-+*/<?:!(),\[\]{}'=]))")


def tokenize(s):
    out, i = [], 0
    while i < len(s):
        m = _TOK.match(s, i)
        if not m:
            if s[i:].strip() == "":
                break
            raise ValueError("cannot tokenize at: " + s[i:i + 40])
        i = m.end()
        if m.group(1):
            out.append(("dim", m.group(1)))
        elif m.group(2):
            out.append(("num", float(m.group(2))))
        elif m.group(3):
            out.append(("tmp", int(m.group(3)[1:])))
        elif m.group(4):
            out.append(("id", m.group(4)))
        else:
            out.append(("op", m.group(5)))
    return out


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ("eof", None)

    def eat(self, kind=None, val=None):
        tok = self.peek()
        if (kind and tok[0] != kind) or (val is not None and tok[1] != val):
            raise ValueError(f"expected {kind} {val}, got {tok} at {self.i}")
        self.i += 1
        return tok

    def is_op(self, v, k=0):
        return self.peek(k) == ("op", v)

    # precedence: ?: < == < '<' < + - < * / < unary < postfix
    def expr(self):
        c = self.eq()
        if self.is_op("?"):
            self.eat(); a = self.expr(); self.eat("op", ":"); b = self.expr()
            return ("sel", c, a, b)
        return c

    def eq(self):
        a = self.lt()
        while self.is_op("=="):
            self.eat(); a = ("bin", "==", a, self.lt())
        return a

    def lt(self):
        a = self.add()
        while self.is_op("<"):
            self.eat(); a = ("bin", "<", a, self.add())
        return a

    def add(self):
        a = self.mul()
        while self.is_op("+") or self.is_op("-"):
            op = self.eat()[1]; a = ("bin", op, a, self.mul())
        return a

    def mul(self):
        a = self.unary()
        while self.is_op("*") or self.is_op("/"):
            op = self.eat()[1]; a = ("bin", op, a, self.unary())
        return a

    def unary(self):
        if self.is_op("-"):
            self.eat(); return ("neg", self.unary())
        if self.is_op("!"):
            self.eat(); return ("not", self.unary())
        return self.postfix()

    def postfix(self):
        a = self.primary()
        while True:
            if self.is_op("{"):
                self.eat(); self.eat("num"); self.eat("op", "}")          # output selector {0}
            elif self.is_op("'"):
                self.eat(); a = ("T", a)
            elif self.is_op("["):
                self.eat()
                parts, cur = [], None
                while not self.is_op("]"):
                    if self.is_op(":"):
                        self.eat(); parts.append(cur); cur = None
                    else:
                        cur = int(self.eat("num")[1])
                parts.append(cur); self.eat("op", "]")
                if self.is_op("="):                                       # (v)[i] = e
                    self.eat(); e = self.expr(); a = ("set", a, parts[0], e)
                else:
                    a = ("idx", a, tuple(parts))
            else:
                return a

    def primary(self):
        k, v = self.peek()
        if k == "num":
            self.eat(); return ("num", v)
        if k == "tmp":
            self.eat(); return ("tmp", v)
        if k == "dim":
            self.eat(); return ("dim", v)
        if k == "id":
            self.eat()
            if self.is_op("("):
                self.eat(); args = []
                if not self.is_op(")"):
                    args.append(self.expr())
                    while self.is_op(","):
                        self.eat(); args.append(self.expr())
                self.eat("op", ")")
                return ("call", v, args)
            return ("id", v)
        if (k, v) == ("op", "("):
            self.eat(); e = self.expr(); self.eat("op", ")")
            return e
        raise ValueError(f"unexpected token {k} {v} at {self.i}")


def split_top(s):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur); cur = ""
        else:
            cur += ch
    parts.append(cur)
    return [p.strip() for p in parts]


def parse_program(s):
    """-> (definitions {k: ast}, result ast)"""
    defs, res = {}, None
    for part in split_top(s):
        m = re.match(r"@(\d+)=(.*)$", part, re.S)
        if m:
            defs[int(m.group(1))] = Parser(tokenize(m.group(2))).expr()
        else:
            p = Parser(tokenize(part)); res = p.expr()
            if p.peek()[0] != "eof":
                raise ValueError("trailing tokens in result expression")
    return defs, res


class Evaluator:
    """env: name -> value; funcs: name -> python callable; shapes: callable(reshape_arg_ast, value) -> matrix"""

    def __init__(self, defs, env, funcs, reshape):
        self.defs, self.env, self.funcs, self.reshape, self.memo = defs, env, funcs, reshape, {}

    def child(self, env):
        return Evaluator(self.defs, env, self.funcs, self.reshape)

    def ev(self, a):
        k = a[0]
        if k == "num":
            return a[1]
        if k == "id":
            return self.env[a[1]]
        if k == "tmp":
            if a[1] not in self.memo:
                self.memo[a[1]] = self.ev(self.defs[a[1]])
            return self.memo[a[1]]
        if k == "neg":
            return -self.ev(a[1])
        if k == "not":
            return not bool(self.ev(a[1]))
        if k == "bin":
            x, y = self.ev(a[2]), self.ev(a[3])
            return {"+": lambda: x + y, "-": lambda: x - y, "*": lambda: x * y, "/": lambda: x / y,
                    "<": lambda: x < y, "==": lambda: x == y}[a[1]]()
        if k == "sel":
            return self.ev(a[2]) if bool(self.ev(a[1])) else self.ev(a[3])
        if k == "T":
            return np.transpose(self.ev(a[1]))
        if k == "idx":
            v = np.asarray(self.ev(a[1])); sl = a[2]
            flat = v.reshape(-1, order="F")
            if len(sl) == 1:
                return flat[sl[0]]
            return flat[slice(*sl)]
        if k == "set":
            v = np.array(self.ev(a[1]), dtype=float, copy=True); v[a[2]] = self.ev(a[3]); return v
        if k == "call":
            name, args = a[1], a[2]
            if name == "reshape":
                return self.reshape(args[0], self.ev(args[0]))
            if name in ("zeros", "ones"):
                r, c = (int(q) for q in args[0][1].split(",")[0].split("x"))
                shp = (r, c) if (r > 1 and c > 1) else (max(r, c),)
                return np.zeros(shp) if name == "zeros" else np.ones(shp)
            vals = [self.ev(x) for x in args]
            if name == "vertcat":
                return np.concatenate([np.atleast_1d(np.asarray(v, dtype=float)) for v in vals])
            if name in ("dense", "project"):
                return vals[0]
            if name == "mac":
                return np.asarray(vals[0]) @ np.asarray(vals[1]) + vals[2]
            if name in self.funcs:
                return self.funcs[name](self, *vals)
            f = {"sqrt": math.sqrt, "sin": math.sin, "cos": math.cos, "atan": math.atan, "asin": math.asin,
                 "sq": lambda x: x * x, "pow": math.pow, "fmax": max, "fmin": min, "fmod": math.fmod}[name]
            return f(*vals)
        raise ValueError("bad node " + str(k))
