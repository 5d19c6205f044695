#!/usr/bin/env python3
"""Development-time tool (needs /root/reference, never runs on the GPU box): derives the IAU-1980 nutation series
as a TABLE — (multipliers of l, l', F, D, Omega; A + A' t) per term — from the behaviour of the reference's
hand-expanded `nutn80` (src/earth_orientation.rs:170-456), by symbolic expansion of its trigonometric algebra
(product-to-sum over the five fundamental arguments). The table is data (the published series, 106 terms); the
evaluator that uses it (oracle/earthrot.c, device/earthrot.hpp) is this repository's own.

Writes outfit_amd/csrc/device/nutation_table.hpp and oracle/nutation_table.h and prints the series evaluated at
J2000 next to the reference's known answer (earth_orientation.rs:606-610)."""
import math
import os
import re
import sys
from fractions import Fraction

SRC = "/root/reference/src/earth_orientation.rs"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
text = open(SRC).read().splitlines()
body = "\n".join(text[169:456])

# ---- symbolic values: dict {(kind, kvec): [a, b]} meaning sum (a + b t) * kind(k . args), plus pure numbers under key None
ZERO5 = (0, 0, 0, 0, 0)


def norm_key(kind, k):
    if kind == "c":
        if k < tuple(-x for x in k):
            k = tuple(-x for x in k)
        return ("c", k), 1
    if k == ZERO5:
        return None, 0
    if k < tuple(-x for x in k):
        return ("s", tuple(-x for x in k)), -1
    return ("s", k), 1


class Sym:
    def __init__(self, d=None):
        self.d = {} if d is None else d

    @staticmethod
    def num(a, b=0):
        return Sym({("c", ZERO5): [Fraction(a), Fraction(b)]})

    @staticmethod
    def trig(kind, k):
        key, sg = norm_key(kind, tuple(k))
        return Sym({} if key is None else {key: [Fraction(sg), Fraction(0)]})

    def add(self, o, sign=1):
        r = {k: v[:] for k, v in self.d.items()}
        for k, v in o.d.items():
            cur = r.setdefault(k, [Fraction(0), Fraction(0)])
            cur[0] += sign * v[0]
            cur[1] += sign * v[1]
        return Sym({k: v for k, v in r.items() if v[0] != 0 or v[1] != 0})

    def mul(self, o):
        r = Sym()
        for (k1, ka), (a1, b1) in self.d.items():
            for (k2, kb), (a2, b2) in o.d.items():
                if b1 != 0 and b2 != 0:
                    raise ValueError("t^2 term")
                a, b = a1 * a2, a1 * b2 + b1 * a2
                sm = tuple(x + y for x, y in zip(ka, kb))
                df = tuple(x - y for x, y in zip(ka, kb))
                if k1 == "c" and k2 == "c":
                    parts = [("c", df, 1), ("c", sm, 1)]
                elif k1 == "s" and k2 == "s":
                    parts = [("c", df, 1), ("c", sm, -1)]
                elif k1 == "s" and k2 == "c":
                    parts = [("s", sm, 1), ("s", df, 1)]
                else:  # c * s
                    parts = [("s", sm, 1), ("s", df, -1)]
                for kind, k, sg in parts:
                    key, s2 = norm_key(kind, k)
                    if key is None:
                        continue
                    cur = r.d.setdefault(key, [Fraction(0), Fraction(0)])
                    cur[0] += Fraction(sg * s2, 2) * a
                    cur[1] += Fraction(sg * s2, 2) * b
        return Sym({k: v for k, v in r.d.items() if v[0] != 0 or v[1] != 0})


# ---- tiny expression parser
TOK = re.compile(r"\s*(?:(\d[\d_]*\.?\d*(?:[eE][-+]?\d+)?)|([A-Za-z_]\w*)|(.))")


def tokenize(s):
    out = []
    for num, ident, op in TOK.findall(s):
        if num:
            out.append(("n", Fraction(num.replace("_", ""))))
        elif ident:
            out.append(("i", ident))
        elif op.strip():
            out.append(("o", op))
    return out


class Parser:
    def __init__(self, toks, env):
        self.t, self.i, self.env = toks, 0, env

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else ("e", None)

    def eat(self):
        tok = self.peek()
        self.i += 1
        return tok

    def expr(self):
        v = self.term()
        while self.peek() in (("o", "+"), ("o", "-")):
            op = self.eat()[1]
            v = v.add(self.term(), 1 if op == "+" else -1)
        return v

    def term(self):
        v = self.unary()
        while self.peek() == ("o", "*"):
            self.eat()
            v = v.mul(self.unary())
        return v

    def unary(self):
        if self.peek() == ("o", "-"):
            self.eat()
            return Sym().add(self.unary(), -1)
        tok = self.eat()
        if tok[0] == "n":
            return Sym.num(tok[1])
        if tok[0] == "i":
            return self.env[tok[1]]
        if tok == ("o", "("):
            v = self.expr()
            assert self.eat() == ("o", ")")
            return v
        raise ValueError(tok)


env = {"t": Sym.num(0, 1)}
basis = {"l": 0, "p": 1, "x": None, "d": 3, "n": 4}
for nm, idx in basis.items():
    k = [0, 0, 0, 0, 0]
    if nm == "x":
        k[2] = 2  # x = f + f
    else:
        k[idx] = 1
    env["s" + nm] = Sym.trig("s", k)
    env["c" + nm] = Sym.trig("c", k)

stmts = re.findall(r"(let(?:\s+mut)?\s+(\w+)\s*=|(\w+)\s*\+=)\s*(.*?);", body, re.S)
series = {}
for _, let_name, aug_name, rhs in stmts:
    name = let_name or aug_name
    if name in ("t", "t2", "t3", "l", "p", "f", "d", "n", "x") or "sc(" in rhs or "as_rad" in rhs:
        continue
    if rhs.strip().startswith("(") and "sin_cos" in rhs:
        continue
    val = Parser(tokenize(rhs), env).expr()
    if name in ("dpsi", "deps"):
        series[name] = series[name].add(val) if aug_name and name in series else val
    else:
        env[name] = val
        assert len(val.d) == 1, (name, val.d)  # every compound variable is a single sine or cosine

rows = {"dpsi": [], "deps": []}
for name, kind in (("dpsi", "s"), ("deps", "c")):
    for (kd, k), (a, b) in sorted(series[name].d.items(), key=lambda kv: -abs(kv[1][0])):
        assert kd == kind, (name, kd, k)
        rows[name].append((k, float(a), float(b)))
    print(name, len(rows[name]), "terms; largest:", rows[name][:3])

# ---- numeric check at J2000 against the reference's KAT
RADSEC = math.pi / 648000.0
args0 = [485866.733 * RADSEC, 1287099.804 * RADSEC, 335778.877 * RADSEC, 1072261.307 * RADSEC, 450160.280 * RADSEC]
dpsi = sum(a * math.sin(sum(ki * xi for ki, xi in zip(k, args0))) for k, a, b in rows["dpsi"]) * 1e-4
deps = sum(a * math.cos(sum(ki * xi for ki, xi in zip(k, args0))) for k, a, b in rows["deps"]) * 1e-4
print("dpsi(J2000) =", repr(dpsi), " reference KAT -13.923385169502602, diff", dpsi + 13.923385169502602)
print("deps(J2000) =", repr(deps), " reference KAT -5.773808263765919, diff", deps + 5.773808263765919)
if abs(dpsi + 13.923385169502602) > 1e-10 or abs(deps + 5.773808263765919) > 1e-10:
    sys.exit("derived table does not reproduce the known answers")


def emit(path, guard, static):
    with open(path, "w") as f:
        f.write("/* GENERATED by tools/derive_nutation_table.py — the IAU 1980 nutation series as a table:\n"
                " * multipliers of (l, l', F, D, Omega) and the coefficient A + A' t in units of 1e-4 arcsec,\n"
                " * largest terms first. dpsi: sum (A + A' t) sin(arg); deps: sum (A + A' t) cos(arg). */\n")
        f.write(f"#ifndef {guard}\n#define {guard}\n")
        for name in ("dpsi", "deps"):
            f.write(f"#define NUT80_N_{name.upper()} {len(rows[name])}\n")
            f.write(f"{static} const signed char NUT80_{name.upper()}_K[{len(rows[name])}][5] = {{\n")
            f.write("\n".join("  {%d, %d, %d, %d, %d}," % k for k, a, b in rows[name]))
            f.write(f"\n}};\n{static} const double NUT80_{name.upper()}_A[{len(rows[name])}][2] = {{\n")
            f.write("\n".join("  {%r, %r}," % (a, b) for k, a, b in rows[name]))
            f.write("\n};\n")
        f.write("#endif\n")


emit(os.path.join(ROOT, "oracle", "nutation_table.h"), "ORC_NUTATION_TABLE_H", "static")
emit(os.path.join(ROOT, "outfit_amd", "csrc", "device", "nutation_table.hpp"), "OUTFIT_NUTATION_TABLE_HPP", "static __device__")
print("tables written")
