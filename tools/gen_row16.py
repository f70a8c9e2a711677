#!/usr/bin/env python3
"""Generator + simulator of the ROW-OF-16 pairing check (round 6): csrc/elpasso_pair16_prog.h.

The check  e(sig1, K) e(-sig2, gg) == 1  (src/ps-verifier.cc:31-34, 132-137) for ONE item on ONE 16-lane row of a wave.  Twelve lanes hold one base-field
coefficient each of an Fp12 value f = sum_k f_k w^k (lane q = 2k + c); every step of the computation is ONE inner product per lane,
    dest[lane] = sum_t  coeff_t * slot[a_t] * slot[b_t]            (a single Montgomery reduction, quad.h fp_dot),
over operands that live in the row's LDS slots -- the same instruction stream on every lane, lane-dependent ADDRESSES and small integer coefficients from a table.
Linear operations (sums, xi-multiples, conjugations, the factor 3 b' of the twist) never run on the device: they are expanded symbolically HERE into the terms of the next
product.  The sequence of steps (Miller loop over the 65 digits of 6z + 2, final exponentiation by the Fuentes-Castaneda-Knapp-Rodriguez multiple) is a flat program.

This file (build tooling, runs in the build container; output committed):
  * a small symbolic engine (linear forms over slots, their products as quadratic forms),
  * the formulas, written once (homogeneous projective doubling / mixed addition with the lines of elp/pairing.h, Fp12 products, Granger-Scott squarings, Frobenius maps,
    the inversion of the easy part through the norms Fp12 -> Fp6 -> Fp2 -> Fp),
  * a SIMULATOR of the generated tables over big integers, checked against the big-int model's pairing (oracle/pymodel.py) on valid and invalid signatures before
    anything is written: the device executes exactly these tables,
  * the emitter of the tables and the program as C arrays.
Both curves: python tools/gen_row16.py [--curve bn254|bls12_381]  (BN254: D-type twist, 29-bit limbs, 13 limb products per accumulator column, 6z + 2 with two closing
Frobenius additions, FKR hard part; BLS12-381: M-type twist, 28-bit limbs, 35 products per column, loop over |z|, the Hayashida-Hayasaka-Teruya hard part taken to the third
power as elp/pairing.h does for verdicts).  The headroom is checked per lane below.
"""
import os
import random
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from oracle.pymodel import BLS12_381, BN254, Groups, Mcl  # noqa: E402

CURVE = "bn254"
if "--curve" in sys.argv:
    CURVE = sys.argv[sys.argv.index("--curve") + 1].lower()
assert CURVE in ("bn254", "bls12_381")
CV = BN254 if CURVE == "bn254" else BLS12_381
IS_BN = CV.is_bn
G = Groups(CV)
F = G.F
P = CV.p
NLIMB, LBITS = (9, 29) if IS_BN else (14, 28)        # params_<curve>.h NL, LB
HEADROOM = 13 if IS_BN else 35                       # limb products per accumulator column (params_<curve>.h HEADROOM - 1)
ALLOWED = (1, 2, 3, 4, 6, 12)

# ---------------------------------------------------------------------------------------------------------------- slots
NLANES = 16
# per-row slots (absolute ids below 128); constants shared by the rows of a workgroup from 128 on
_next = [0]
NAMES = {}


def alloc(name, n=1):
    b = _next[0]
    _next[0] += n
    NAMES[name] = b
    return b


T_ = alloc("T", 6)            # X.re X.im Y.re Y.im Z.re Z.im
Q_ = alloc("Q", 4)            # xq.re xq.im yq.re yq.im   (the Frobenius images replace it for the two closing additions)
P1 = alloc("P1", 2)           # x, y of sig1
P2 = alloc("P2", 2)           # x, y of -sig2
L1 = alloc("L1", 12)          # doubling level 1: U = XY, B = Y^2, E = 3b'Z^2, E3 = 3E, H = 2YZ, Lb = -3X^2
LV = alloc("LV", 4)           # evaluated variable line: a' = l.a y_P, b' = l.b x_P
LF = alloc("LF", 6)           # fixed line as stored (a, b, c)
LFS = alloc("LFS", 4)         # fixed line evaluated: a' = a y_P2, b' = b x_P2
AD = alloc("AD", 16)          # addition step: theta, mu, D, Ct, lc, E, F, G      (aliased by the temporaries of the inversion)
W0 = alloc("W0", 12)
X0 = alloc("X0", 12)          # xi * W0
W1 = alloc("W1", 12)
EQ = alloc("EQ", 2)           # BLS12-381: xi Z^2, the first link of the chain that makes E = 3 b' Z^2 = 12 xi Z^2 within the accumulator's headroom
NSLOT = _next[0]
assert NSLOT <= 120, NSLOT
CONST_BASE = 128
CONSTS = {}                   # name -> (index, integer value mod p)


def const(name, value):
    if name not in CONSTS:
        CONSTS[name] = (CONST_BASE + len(CONSTS), value % P)
    return CONSTS[name][0]


ZERO = const("zero", 0)
ONE = const("one", 1)
E_INIT = None      # (set once TB3 is known: E = 3 b', E3 = 9 b' of a point with Z = 1)

# ---------------------------------------------------------------------------------------------------------------- symbolic forms


class Lin(dict):
    """linear form over slots: slot -> integer coefficient"""

    def __add__(self, o):
        r = Lin(self)
        for k, v in o.items():
            r[k] = r.get(k, 0) + v
            if r[k] == 0:
                del r[k]
        return r

    def __neg__(self):
        return Lin({k: -v for k, v in self.items()})

    def __sub__(self, o):
        return self + (-o)

    def scale(self, c):
        return Lin({k: v * c for k, v in self.items()}) if c else Lin()


def S(slot):
    return Lin({slot: 1})


class Quad(dict):
    """quadratic form: (slot a, slot b) -> integer coefficient (a <= b: products commute)"""

    def __add__(self, o):
        r = Quad(self)
        for k, v in o.items():
            r[k] = r.get(k, 0) + v
            if r[k] == 0:
                del r[k]
        return r

    def __neg__(self):
        return Quad({k: -v for k, v in self.items()})

    def __sub__(self, o):
        return self + (-o)

    def scale(self, c):
        return Quad({k: v * c for k, v in self.items()}) if c else Quad()


def lmul(a, b):
    r = Quad()
    for sa, ca in a.items():
        for sb, cb in b.items():
            k = (sa, sb) if sa <= sb else (sb, sa)
            r[k] = r.get(k, 0) + ca * cb
            if r[k] == 0:
                del r[k]
    return r


def f2(base):                       # the Fp2 value stored at slots (base, base + 1)
    return (S(base), S(base + 1))


def f2add(x, y):
    return (x[0] + y[0], x[1] + y[1])


def f2sub(x, y):
    return (x[0] - y[0], x[1] - y[1])


def f2neg(x):
    return (-x[0], -x[1])


def f2scale(x, c):
    return (x[0].scale(c), x[1].scale(c))


def f2conj(x):
    return (x[0], -x[1])


def f2xi(x):                        # (1 + i)(a + b i) = (a - b) + (a + b) i
    return (x[0] - x[1], x[0] + x[1])


def f2mul(x, y):                    # -> pair of Quads
    return (lmul(x[0], y[0]) - lmul(x[1], y[1]), lmul(x[0], y[1]) + lmul(x[1], y[0]))


def q2add(x, y):
    return (x[0] + y[0], x[1] + y[1])


def q2sub(x, y):
    return (x[0] - y[0], x[1] - y[1])


def q2scale(x, c):
    return (x[0].scale(c), x[1].scale(c))


def f2mulfp(x, s):                  # Fp2 (linear) times a base-field slot (linear form s)
    return (lmul(x[0], s), lmul(x[1], s))


def v12(base, sign=None):           # the Fp12 value at 12 consecutive slots; sign[k] = -1 negates coefficient k (conjugation = odd k negated)
    out = []
    for k in range(6):
        x = f2(base + 2 * k)
        out.append(f2neg(x) if sign and sign[k] < 0 else x)
    return out


CONJ = [1, -1, 1, -1, 1, -1]


def mul12(a, xa, b):
    """h_k = sum_{i <= k} a_i b_{k-i} + sum_{i > k} (xi a_i) b_{k-i+6};   a, xa, b: lists of 6 Fp2 linear forms"""
    out = []
    for k in range(6):
        acc = (Quad(), Quad())
        for i in range(6):
            acc = q2add(acc, f2mul(a[i] if i <= k else xa[i], b[(k - i) % 6]))
        out.append(acc)
    return out


def sqr12(a, xa):
    out = []
    for k in range(6):
        acc = (Quad(), Quad())
        for i in range(6):
            for j in range(i, 6):
                if (i + j) % 6 == k:
                    t = f2mul(xa[i] if i + j >= 6 else a[i], a[j])
                    acc = q2add(acc, t if i == j else q2scale(t, 2))
        out.append(acc)
    return out


# ---------------------------------------------------------------------------------------------------------------- steps
class Step:
    """one table: per lane a quadratic form, a destination slot (or None), and the post-operation
         post = None                     dest = dot
         post = ("gs", slot, s)          dest = 3 * dot + s * slot[..]  (s = +-2), weakly reduced       (Granger-Scott squaring)
       wx: lanes 0..11 also write xi * (the Fp2 coefficient they form with their pair lane) to X0 + lane"""

    def __init__(self, name, wx=False, tag=0):
        self.name, self.wx, self.tag = name, wx, tag
        self.lanes = [None] * NLANES

    def out(self, lane, quad, dest, post=None):
        assert self.lanes[lane] is None, (self.name, lane)
        terms = []
        w = 0
        for (a, b), c in sorted(quad.items()):
            assert abs(c) in ALLOWED, (self.name, lane, c)
            terms.append((a, b, c))
            w += abs(c)
        assert w <= HEADROOM, "step %s lane %d: weight %d" % (self.name, lane, w)
        assert len(terms) <= 12, (self.name, lane, len(terms))
        self.lanes[lane] = (terms, dest, post)

    def out2(self, lane0, quads, dest):            # an Fp2 result on the lane pair (lane0, lane0 + 1) -> slots (dest, dest + 1)
        self.out(lane0, quads[0], dest)
        self.out(lane0 + 1, quads[1], dest + 1)

    def out12(self, quads, dest):                  # an Fp12 result on lanes 0..11
        for k in range(6):
            self.out2(2 * k, quads[k], dest + 2 * k)

    @property
    def nt(self):
        n = max([len(l[0]) for l in self.lanes if l] + [1])
        for cand in (1, 2, 3, 4, 6, 8, 12):
            if n <= cand:
                return cand
        raise AssertionError(n)


STEPS = []


def step(name, **kw):
    s = Step(name, **kw)
    s.id = len(STEPS)
    STEPS.append(s)
    return s


# program entries: ("dot", step id) | ("line", n) load the fixed line n into LF | ("inv", src, dst) | ("ld", reg, area) | ("st", reg) | ("pubx",) X0 = xi W0
# | ("check",)
PROG = []

# ---------------------------------------------------------------------------------------------------------------- the Miller loop
f, xf = v12(W0), v12(X0)
X, Y, Z = f2(T_), f2(T_ + 2), f2(T_ + 4)
xq, yq = f2(Q_), f2(Q_ + 2)
xP1, yP1, xP2, yP2 = S(P1), S(P1 + 1), S(P2), S(P2 + 1)
U, Bv, E, E3, H, Lb = (f2(L1 + 2 * i) for i in range(6))
lva, lvb = f2(LV), f2(LV + 2)
lfa, lfb, lfc = f2(LF), f2(LF + 2), f2(LF + 4)
lfsa, lfsb = f2(LFS), f2(LFS + 2)
TB3 = (3, -3) if IS_BN else (12, 12)
E_INIT = [const("e_init%d" % i_, v_) for i_, v_ in enumerate((TB3[0], TB3[1], 3 * TB3[0], 3 * TB3[1]))]
TB3 = TB3   # 3 b': BN254 b' = 2 / (1 + i) = 1 - i (D-type twist); BLS12-381 b' = 4 (1 + i) (M-type twist)
assert F.f2_muls(F.b2, 3) == (TB3[0] % P, TB3[1] % P), F.b2


def f2mulc(x, c):                    # Fp2 (linear) times the small constant c[0] + c[1] i
    return (x[0].scale(c[0]) - x[1].scale(c[1]), x[0].scale(c[1]) + x[1].scale(c[0]))


def line_product(la, lb, lc):
    """f * line for the line (l.a y_P, l.b x_P, l.c) of elp/pairing.h (ml_apply_line): D-type twist (BN254) line = l.a y_P + l.b x_P w + l.c w^3; M-type twist
    (BLS12-381) line = l.c + l.b x_P w^2 + l.a y_P w^3.  h_k = sum_pos coef_pos F_{k - pos},  F_j = f_j (j >= 0), xi f_{j+6} (j < 0)"""
    placed = ((0, la), (1, lb), (3, lc)) if IS_BN else ((0, lc), (2, lb), (3, la))
    out = []
    for k in range(6):
        acc = (Quad(), Quad())
        for pos, coef in placed:
            acc = q2add(acc, f2mul(f[k - pos] if k >= pos else xf[k - pos + 6], coef))
        out.append(acc)
    return out


# S1: f <- f^2 (all twelve lanes) and, on the four spare lanes, the fixed line evaluated at P2
s_sqr = step("sqr_f", wx=True)
s_sqr.out12(sqr12(f, xf), W0)
for j, (src, pt) in enumerate(((lfa[0], yP2), (lfa[1], yP2), (lfb[0], xP2), (lfb[1], xP2))):
    s_sqr.out(12 + j, lmul(src, pt), LFS + j)
# S2: doubling, level 1.  Homogeneous projective (elp/pairing.h ml_dbl_step) with the halvings dropped: the point comes out as (4 X', 4 Y', 4 Z').
# E = 3 b' Z^2 and E3 = 3 E of the CURRENT point are made on the spare lanes of the two line products that precede a doubling (E: 12 limb products per column,
# 3 E in one go would be 36: over the accumulator's headroom); the set-up writes them for the first doubling (Z = 1).
zz = f2mul(Z, Z)
e_q = (zz[0].scale(TB3[0]) - zz[1].scale(TB3[1]), zz[0].scale(TB3[1]) + zz[1].scale(TB3[0]))      # E = 3 b' Z^2


Eq = f2(EQ)


def spare_e(s_):          # on the spare lanes of the variable line's product: BN254 E itself (12 limb products per column); BLS12-381 xi Z^2 (E = 12 xi Z^2 would be 48)
    if IS_BN:
        s_.out2(12, e_q, L1 + 4)
    else:
        s_.out2(12, f2mul(f2xi(Z), Z), EQ)


def spare_e3(s_):         # on the spare lanes of the fixed line's product: BN254 3 E; BLS12-381 E = 12 (xi Z^2)
    if IS_BN:
        s_.out2(12, q2scale(f2mulfp(E, S(ONE)), 3), L1 + 6)
    else:
        s_.out2(12, q2scale(f2mulfp(Eq, S(ONE)), 12), L1 + 4)


s_d1 = step("dbl1")
s_d1.out2(0, f2mul(X, Y), L1 + 0)                          # U
s_d1.out2(2, f2mul(Y, Y), L1 + 2)                          # B
s_d1.out2(4, q2scale(f2mul(Y, Z), 2), L1 + 8)              # H = 2 Y Z
s_d1.out2(6, q2scale(f2mul(X, X), -3), L1 + 10)            # Lb = -3 X^2
if not IS_BN:
    s_d1.out2(8, q2scale(f2mulfp(E, S(ONE)), 3), L1 + 6)     # BLS12-381: 3 E, the last link of the chain (level 2 reads it)
# S3: doubling, level 2 (+ the variable line evaluated at P1 on the spare lanes)
s_d2 = step("dbl2")
s_d2.out2(0, q2scale(f2mul(U, f2sub(Bv, E3)), 2), T_ + 0)                                   # 4 X' = 2 U (B - 3E)
s_d2.out2(2, q2sub(q2add(f2mul(Bv, Bv), q2scale(f2mul(Bv, E3), 2)), f2mul(E, E3)), T_ + 2)  # 4 Y' = (B + 3E)^2 - 12 E^2 = B^2 + 2 B (3E) - E (3E)
s_d2.out2(4, q2scale(f2mul(Bv, H), 4), T_ + 4)                                              # 4 Z' = 4 B H
for j, (src, pt) in enumerate(((H[0], yP1), (H[1], yP1), (Lb[0], xP1), (Lb[1], xP1))):
    s_d2.out(12 + j, lmul(src, pt), LV + j)
# S4 / S5: f <- f * line
s_lv = step("line_v", wx=True, tag=1)
s_lv.out12(line_product(lva, lvb, f2sub(Bv, E)), W0)         # l.c = B - E, expanded (E: the OLD point's, read before this step's spare lanes replace it)
spare_e(s_lv)
s_lf = step("line_f", wx=True, tag=2)
s_lf.out12(line_product(lfsa, lfsb, lfc), W0)
spare_e3(s_lf)
# addition step (mixed, elp/pairing.h ml_add_step): four levels; `sy` = the sign of y_Q (the digit's sign), `ty` = the sign of T.Y (the closing additions of a
# negative z see -T.Y once)
theta, mu, Dv, Ct, lc, Ea, Fa, Ga = (f2(AD + 2 * i) for i in range(8))
one = S(ONE)


def add_steps(sy, ty, tagname):
    a1 = step("add1" + tagname)
    yqs = f2scale(yq, sy)
    Ys = f2scale(Y, ty)
    a1.out2(0, q2sub(f2mulfp(Ys, one), f2mul(yqs, Z)), AD + 0)          # theta = Y - yq Z
    a1.out2(2, q2sub(f2mulfp(X, one), f2mul(xq, Z)), AD + 2)            # mu = X - xq Z
    for j, (src, pt) in enumerate(((lfa[0], yP2), (lfa[1], yP2), (lfb[0], xP2), (lfb[1], xP2))):
        a1.out(12 + j, lmul(src, pt), LFS + j)                            # the fixed line of this step evaluated at P2
    a2 = step("add2" + tagname)
    a2.out2(0, f2mul(mu, mu), AD + 4)                                   # D
    a2.out2(2, f2mul(theta, theta), AD + 6)                             # Ct
    a2.out2(4, q2sub(f2mul(theta, xq), f2mul(mu, yqs)), AD + 8)         # l.c = theta xq - mu yq
    for j, (src, pt, sg) in enumerate(((mu[0], yP1, 1), (mu[1], yP1, 1), (theta[0], xP1, -1), (theta[1], xP1, -1))):
        a2.out(12 + j, lmul(src, pt).scale(sg), LV + j)                   # l.a y_P = mu y_P,  l.b x_P = -theta x_P
    a3 = step("add3" + tagname)
    a3.out2(0, f2mul(mu, Dv), AD + 10)                                  # E = mu^3
    a3.out2(2, f2mul(Z, Ct), AD + 12)                                   # F = Z theta^2
    a3.out2(4, f2mul(X, Dv), AD + 14)                                   # G = X mu^2
    a4 = step("add4" + tagname)
    Hh = f2sub(f2add(Ea, Fa), f2scale(Ga, 2))                           # H = E + F - 2G
    a4.out2(0, f2mul(mu, Hh), T_ + 0)
    a4.out2(2, q2sub(f2mul(theta, f2sub(Ga, Hh)), f2mul(Ea, Ys)), T_ + 2)
    a4.out2(4, f2mul(Z, Ea), T_ + 4)
    return [a1, a2, a3, a4]


ADD = {(1, 1): add_steps(1, 1, "_p"), (-1, 1): add_steps(-1, 1, "_n"), (1, -1): add_steps(1, -1, "_pf"), (-1, -1): add_steps(-1, -1, "_nf")}
s_lva = step("line_va", wx=True, tag=1)
s_lva.out12(line_product(lva, lvb, lc), W0)
spare_e(s_lva)
# f <- conj(f) (z < 0), and the Frobenius images of Q for the two closing additions
s_conj = step("conj_f", wx=True)
s_conj.out12([f2mulfp(x, one) for x in v12(W0, CONJ)], W0)
gam = F.gamma                        # gamma[k] = xi^(k (p-1)/6)


def cslot2(name, v):
    return (S(const(name + ".re", v[0])), S(const(name + ".im", v[1])))


s_q1 = step("frob_q1")               # Q <- (conj(x) gamma_2, conj(y) gamma_3)
s_q1.out2(0, f2mul(f2conj(xq), cslot2("g12", gam[2])), Q_ + 0)
s_q1.out2(2, f2mul(f2conj(yq), cslot2("g13", gam[3])), Q_ + 2)
# pi^2(Q) = (x gamma_2 conj(gamma_2)..): applied to the ORIGINAL Q it is x * g22, y * g23 with g2k = gamma_k * conj(gamma_k)^... ; here Q already holds pi(Q), so apply pi once more
s_q2 = step("frob_q2")
s_q2.out2(0, f2mul(f2conj(xq), cslot2("g12", gam[2])), Q_ + 0)
s_q2.out2(2, f2mul(f2conj(yq), cslot2("g13", gam[3])), Q_ + 2)

# NAF digits of 6z + 2, MSB first without the leading one (params_bn254.h ate_naf)


def naf(k):
    out = []
    while k:
        if k & 1:
            d = 2 - (k % 4)
            k -= d
        else:
            d = 0
        out.append(d)
        k >>= 1
    return out


DIG = list(reversed(naf(CV.ate_loop)[:-1]))
assert len(DIG) == (65 if IS_BN else 64), len(DIG)
line_no = 0
for i, d in enumerate(DIG):
    PROG.append(("line", line_no))
    PROG.append(("dot", s_sqr.id))
    PROG.append(("dot", s_d1.id))
    PROG.append(("dot", s_d2.id))
    PROG.append(("dot", s_lv.id))
    PROG.append(("dot", s_lf.id))
    line_no += 1
    if d:
        PROG.append(("line", line_no))
        for s_ in ADD[(d, 1)]:
            PROG.append(("dot", s_.id))
        PROG.append(("dot", s_lva.id))
        PROG.append(("dot", s_lf.id))
        line_no += 1
assert CV.z < 0
PROG.append(("dot", s_conj.id))
if IS_BN:
    # closing additions: T <- -T (sign carried by the tables of the first one), Q1 = pi(Q), Q2 = -pi^2(Q)
    PROG.append(("dot", s_q1.id))
    PROG.append(("line", line_no))
    for s_ in ADD[(1, -1)]:
        PROG.append(("dot", s_.id))
    PROG.append(("dot", s_lva.id))
    PROG.append(("dot", s_lf.id))
    line_no += 1
    PROG.append(("dot", s_q2.id))
    PROG.append(("line", line_no))
    for s_ in ADD[(-1, 1)]:
        PROG.append(("dot", s_.id))
    PROG.append(("dot", s_lva.id))
    PROG.append(("dot", s_lf.id))
    line_no += 1
NLINES = line_no

# ---------------------------------------------------------------------------------------------------------------- final exponentiation
# registers: each lane keeps its coefficient of up to NREG stored Fp12 values; ("st", r): reg r <- the lane's coefficient of W0; ("ld", r, area): area (0 = W0 with X0, 1 = W1) <- reg r
NREG = 7 if IS_BN else 4
w1 = v12(W1)
s_mul = step("mul_w0_w1", wx=True)                       # W0 <- W0 * W1
s_mul.out12(mul12(f, xf, w1), W0)
s_mulc = step("mul_w0_conj_w1", wx=True)                 # W0 <- W0 * conj(W1)
s_mulc.out12(mul12(f, xf, v12(W1, CONJ)), W0)
s_cyc = step("cyc_sqr", wx=True)                         # Granger-Scott squaring of W0 (cyclotomic subgroup)
XI = [0, 2, 1, 0, 2, 1]
YI = [3, 5, 4, 3, 5, 4]
for k in range(6):
    if k in (0, 2, 4):
        qd = q2add(f2mul(f[XI[k]], f[XI[k]]), f2mul(xf[YI[k]], f[YI[k]]))
    else:
        qd = q2scale(f2mul(xf[XI[k]] if k == 1 else f[XI[k]], f[YI[k]]), 2)
    for c in (0, 1):
        s_cyc.out(2 * k + c, qd[c], W0 + 2 * k + c, post=("gs", W0 + 2 * k + c, 2 if (k & 1) else -2))
# Frobenius maps W0 <- W0^(p^n): coefficient k conjugated n times and scaled by gamma_{n,k} (gamma_{n,k} = the product of the conjugates, as in the model)


def gamma_n(n, k):
    # f^(p^n): coefficient of w^k picks up w^(k (p^n - 1)) = xi^(k (p^n - 1) / 6)
    return F.f2_pow(CV.xi, k * (P ** n - 1) // 6)


FROB = {}
for n in (1, 2, 3):
    s_ = step("frob%d" % n, wx=True)
    outq = []
    for k in range(6):
        x = f[k] if n % 2 == 0 else f2conj(f[k])
        outq.append(f2mul(x, cslot2("g%d%d" % (n, k), gamma_n(n, k))))
    s_.out12(outq, W0)
    FROB[n] = s_
# the inversion of the easy part: N = f conj(f) in Fp6 (even coefficients), Fp6 -> Fp2 -> Fp norms, one base-field inversion
s_norm = step("norm6")                                   # t = (N_0, N_2, N_4) -> AD + 0..5
nq = mul12(f, xf, v12(W0, CONJ))
for j in range(3):
    s_norm.out2(2 * j, nq[2 * j], AD + 2 * j)
t0, t1, t2 = f2(AD), f2(AD + 2), f2(AD + 4)
s_i2 = step("inv6_c")                                    # c0 = t0^2 - xi t1 t2, c1 = xi t2^2 - t0 t1, c2 = t1^2 - t0 t2 -> AD + 6..11
s_i2.out2(0, q2sub(f2mul(t0, t0), f2mul(f2xi(t1), t2)), AD + 6)
s_i2.out2(2, q2sub(f2mul(f2xi(t2), t2), f2mul(t0, t1)), AD + 8)
s_i2.out2(4, q2sub(f2mul(t1, t1), f2mul(t0, t2)), AD + 10)
c0, c1, c2 = f2(AD + 6), f2(AD + 8), f2(AD + 10)
s_i3 = step("inv6_d")                                    # d = t0 c0 + xi (t2 c1 + t1 c2) -> AD + 12, 13
s_i3.out2(0, q2add(f2mul(t0, c0), q2add(f2mul(f2xi(t2), c1), f2mul(f2xi(t1), c2))), AD + 12)
dd = f2(AD + 12)
s_i4 = step("inv2_n")                                    # n = d.re^2 + d.im^2 -> AD + 14
s_i4.out(0, lmul(dd[0], dd[0]) + lmul(dd[1], dd[1]), AD + 14)
# ("inv", AD + 14, AD + 15)
ninv = S(AD + 15)
s_i6 = step("inv2_d")                                    # d^-1 = conj(d) / n -> AD + 12, 13
s_i6.out(0, lmul(dd[0], ninv), AD + 12)
s_i6.out(1, lmul(dd[1], ninv).scale(-1), AD + 13)
s_i7 = step("inv6_t")                                    # t^-1 = c d^-1 -> AD + 0..5
for j, cj in enumerate((c0, c1, c2)):
    s_i7.out2(2 * j, f2mul(cj, dd), AD + 2 * j)
s_i8 = step("inv12", wx=True)                            # f^-1 = conj(f) t^-1 (t^-1 = u0 + u1 w^2 + u2 w^4) -> W1... written to W0? no: to W1 (f stays in W0)
u = [f2(AD), f2(AD + 2), f2(AD + 4)]
cf, cxf = v12(W0, CONJ), v12(X0, CONJ)
outq = []
for k in range(6):
    acc = (Quad(), Quad())
    for j in range(3):
        i = k - 2 * j
        acc = q2add(acc, f2mul(cf[i] if i >= 0 else cxf[i + 6], u[j]))
    outq.append(acc)
s_i8.wx = False
s_i8.out12(outq, W1)
s_conj_keep = s_conj                                     # W0 <- conj(W0)
s_copy = step("w1_from_w0")                              # W1 <- W0
s_copy.out12([f2mulfp(x, one) for x in f], W1)
s_one = step("is_one")                                   # placeholder type for the final comparison (no table use)

def final_exp_entries(prog):
    """appends the easy part, the hard part (Fuentes-Castaneda-Knapp-Rodriguez multiple) and the comparison with 1; W0 holds the Miller value on entry"""
    def exp_z(base_reg):                                     # W0 <- W0^z with W0 = reg[base_reg] on entry (z < 0: conjugate at the end)
        prog.append(("ld", base_reg, 1))                     # W1 = base
        for b_ in bin(ZABS)[3:]:
            prog.append(("dot", s_cyc.id))
            if b_ == "1":
                prog.append(("dot", s_mul.id))
        prog.append(("dot", s_conj.id))

    def ld0(r):
        prog.append(("ld", r, 0))

    def st(r):
        prog.append(("st", r))

    def mul_by(r, conj=False):
        prog.append(("ld", r, 1))
        prog.append(("dot", (s_mulc if conj else s_mul).id))

    def d(s_):
        prog.append(("dot", s_.id))

    # easy part: f1 = conj(f) * f^-1; f2 = frob2(f1) * f1
    for s_ in (s_norm, s_i2, s_i3, s_i4):
        d(s_)
    prog.append(("inv", AD + 14, AD + 15))
    for s_ in (s_i6, s_i7, s_i8, s_conj, s_mul):
        d(s_)                                                # W1 = f^-1; W0 = conj(f); W0 = W0 * W1
    d(s_copy)                                                # W1 = f1
    d(FROB[2])                                               # W0 = f1^(p^2)
    d(s_mul)                                                 # W0 = f2: in the cyclotomic subgroup from here on
    if IS_BN:
        st(R_F)
        exp_z(R_F); st(R_FZ)                                     # fz
        d(s_cyc); st(R_F2Z)                                      # f2z
        d(s_cyc)                                                 # f4z
        mul_by(R_F2Z); st(R_F6Z)                                 # f6z
        exp_z(R_F6Z); st(R_F6Z2)                                 # f6z2
        d(s_cyc)                                                 # f12z2
        st(R_FZ)                                                 # (fz is dead: its register carries the base of the next power)
        exp_z(R_FZ)                                              # f12z3
        mul_by(R_F6Z2)
        mul_by(R_F6Z); st(R_A)                                   # a = f^l2
        mul_by(R_F2Z, conj=True); st(R_B)                        # b = f^l1
        ld0(R_A)
        mul_by(R_F6Z2)
        mul_by(R_F); st(R_FZ)                                    # r = f^l0            (kept in the free register)
        ld0(R_B); d(FROB[1]); d(s_copy); ld0(R_FZ); d(s_mul); st(R_FZ)      # r *= b^p
        ld0(R_A); d(FROB[2]); d(s_copy); ld0(R_FZ); d(s_mul); st(R_FZ)      # r *= a^(p^2)
        ld0(R_B); mul_by(R_F, conj=True)                         # f^l3 = b conj(f)
        d(FROB[3]); d(s_copy); ld0(R_FZ); d(s_mul)               # r *= (f^l3)^(p^3)
    else:
        # BLS12-381 (Hayashida-Hayasaka-Teruya, cubed as in elp/pairing.h final_exp<C, false>): 3 (p^4-p^2+1)/r = (z-1)^2 (z+p) (z^2+p^2-1) + 3
        st(R_F)
        exp_z(R_F)                                           # f^z
        mul_by(R_F, conj=True); st(R_A)                      # a = f^(z-1)
        exp_z(R_A)
        mul_by(R_A, conj=True); st(R_A)                      # a = a^(z-1)
        exp_z(R_A); st(R_T)                                  # t = a^z
        ld0(R_A); d(FROB[1]); mul_by(R_T); st(R_B)           # b = a^p a^z = a^(z+p)
        exp_z(R_B); st(R_T)
        exp_z(R_T); st(R_T)                                  # t = b^(z^2)
        ld0(R_B); d(FROB[2]); mul_by(R_T); mul_by(R_B, conj=True); st(R_T)      # c = b^(p^2) b^(z^2) b^-1
        ld0(R_F); d(s_cyc); mul_by(R_F)                      # f^3
        mul_by(R_T)                                          # c f^3
    prog.append(("check",))


R_F, R_FZ, R_F2Z, R_F6Z, R_F6Z2, R_A, R_B = range(7)
R_T = 1                                                      # BLS12-381 uses four: f (0), t (1), a (2), b (3)
if not IS_BN:
    R_A, R_B = 2, 3
ZABS = abs(CV.z)
final_exp_entries(PROG)

# ---- the closing step of AGGREGATED verification: [ F f_gg(P2) ]^((p^12-1)/r) == 1 with F = the product of the batch's Miller values
# (loaded into W1 by the entry "ldf") and P2 = -sum d_i sig2_i: the fixed pair's Miller loop alone (no point arithmetic), one product, the same final exponentiation
s_scale = step("scale_lf")                                   # the fixed line evaluated at P2 where no other step carries it on its spare lanes
for j, (src, pt) in enumerate(((lfa[0], yP2), (lfa[1], yP2), (lfb[0], xP2), (lfb[1], xP2))):
    s_scale.out(12 + j, lmul(src, pt), LFS + j)
PROG_TAIL = []
ln = 0
for i, dgt in enumerate(DIG):
    PROG_TAIL.append(("line", ln)); ln += 1
    PROG_TAIL.append(("dot", s_sqr.id))
    PROG_TAIL.append(("dot", s_lf.id))
    if dgt:
        PROG_TAIL.append(("line", ln)); ln += 1
        PROG_TAIL.append(("dot", s_scale.id))
        PROG_TAIL.append(("dot", s_lf.id))
PROG_TAIL.append(("dot", s_conj.id))
for _ in range(2 if IS_BN else 0):
    PROG_TAIL.append(("line", ln)); ln += 1
    PROG_TAIL.append(("dot", s_scale.id))
    PROG_TAIL.append(("dot", s_lf.id))
assert ln == NLINES
PROG_TAIL.append(("ldf",))
PROG_TAIL.append(("dot", s_mul.id))
final_exp_entries(PROG_TAIL)

# ---------------------------------------------------------------------------------------------------------------- simulator


def simulate(sig1, sig2n, K, lines, trace=None, prog=None, Fval=None):
    """Runs PROG over big integers for one item: sig1 = (x, y), sig2n = -sig2 = (x, y) or None, K = ((xr, xi), (yr, yi)) or None; lines[n] = (a, b, c) Fp2 triples of gg.
    Returns True iff the result is 1."""
    slot = [0] * 128
    cst = {idx: val for idx, val in CONSTS.values()}

    def rd(s_):
        return cst[s_] if s_ >= CONST_BASE else slot[s_]

    live1 = sig1 is not None and K is not None
    live2 = sig2n is not None
    if K is not None:
        slot[T_:T_ + 6] = [K[0][0], K[0][1], K[1][0], K[1][1], 1, 0]
        slot[Q_:Q_ + 4] = [K[0][0], K[0][1], K[1][0], K[1][1]]
    if sig1 is not None:
        slot[P1], slot[P1 + 1] = sig1
    if sig2n is not None:
        slot[P2], slot[P2 + 1] = sig2n
    slot[W0] = 1
    slot[X0], slot[X0 + 1] = 1, 1
    slot[L1 + 4], slot[L1 + 5], slot[L1 + 6], slot[L1 + 7] = TB3[0] % P, TB3[1] % P, 3 * TB3[0] % P, 3 * TB3[1] % P      # E = 3 b', E3 = 9 b' for Z = 1
    regs = [[0] * 12 for _ in range(NREG)]
    for ent in (prog or PROG):
        if ent[0] == "dot":
            s_ = STEPS[ent[1]]
            dead = (s_.tag == 1 and not live1) or (s_.tag == 2 and not live2)
            res = {}
            for lane, l in enumerate(s_.lanes):
                if not l:
                    continue
                terms, dest, post = l
                v = sum(c * rd(a) * rd(b) for a, b, c in terms) % P
                if post:
                    v = (3 * v + post[2] * rd(post[1])) % P
                res[lane] = (dest, v)
            for lane, (dest, v) in res.items():
                if not (dead and lane < 12):
                    slot[dest] = v
            if s_.wx and not dead:
                for k in range(6):
                    a, b = res[2 * k][1], res[2 * k + 1][1]
                    slot[X0 + 2 * k], slot[X0 + 2 * k + 1] = (a - b) % P, (a + b) % P
        elif ent[0] == "line":
            a, b, c = lines[ent[1]]
            slot[LF:LF + 6] = [a[0], a[1], b[0], b[1], c[0], c[1]]
        elif ent[0] == "inv":
            slot[ent[2]] = pow(slot[ent[1]], -1, P) if slot[ent[1]] else 0
        elif ent[0] == "st":
            regs[ent[1]] = slot[W0:W0 + 12]
        elif ent[0] == "ld":
            if ent[2] == 0:
                slot[W0:W0 + 12] = regs[ent[1]]
                for k in range(6):
                    a, b = slot[W0 + 2 * k], slot[W0 + 2 * k + 1]
                    slot[X0 + 2 * k], slot[X0 + 2 * k + 1] = (a - b) % P, (a + b) % P
            else:
                slot[W1:W1 + 12] = regs[ent[1]]
        elif ent[0] == "ldf":
            slot[W1:W1 + 12] = list(Fval)
        elif ent[0] == "check":
            return slot[W0:W0 + 12] == [1] + [0] * 11
        if trace is not None:
            trace.append((ent, list(slot)))
    raise AssertionError("no check")


def model_lines(Q):
    """the lines of the fixed argument as elp/pairing.h ml_precompute stores them: per step (a, b, c) with line = a y_P + b x_P w + c w^3, projective formulas"""
    X_, Y_, Z_ = Q[0], Q[1], (1, 0)
    out = []
    inv2 = pow(2, -1, P)
    b3 = F.f2_muls(F.b2, 3)

    def dbl():
        nonlocal X_, Y_, Z_
        A = F.f2_muls(F.f2_mul(X_, Y_), inv2)
        B = F.f2_sqr(Y_)
        Cz = F.f2_sqr(Z_)
        E_ = F.f2_mul(Cz, b3)
        Fq = F.f2_muls(E_, 3)
        G_ = F.f2_muls(F.f2_add(B, Fq), inv2)
        H_ = F.f2_sub(F.f2_sqr(F.f2_add(Y_, Z_)), F.f2_add(B, Cz))
        J = F.f2_sqr(X_)
        line = (H_, F.f2_neg(F.f2_muls(J, 3)), F.f2_sub(B, E_))
        X_ = F.f2_mul(A, F.f2_sub(B, Fq))
        Y_ = F.f2_sub(F.f2_sqr(G_), F.f2_muls(F.f2_sqr(E_), 3))
        Z_ = F.f2_mul(B, H_)
        return line

    def add(xq_, yq_):
        nonlocal X_, Y_, Z_
        th = F.f2_sub(Y_, F.f2_mul(yq_, Z_))
        mu_ = F.f2_sub(X_, F.f2_mul(xq_, Z_))
        line = (mu_, F.f2_neg(th), F.f2_sub(F.f2_mul(th, xq_), F.f2_mul(mu_, yq_)))
        Cc = F.f2_sqr(th)
        D_ = F.f2_sqr(mu_)
        E_ = F.f2_mul(mu_, D_)
        Fq = F.f2_mul(Z_, Cc)
        G_ = F.f2_mul(X_, D_)
        H_ = F.f2_sub(F.f2_add(E_, Fq), F.f2_muls(G_, 2))
        Xn = F.f2_mul(mu_, H_)
        Yn = F.f2_sub(F.f2_mul(th, F.f2_sub(G_, H_)), F.f2_mul(E_, Y_))
        Zn = F.f2_mul(Z_, E_)
        X_, Y_, Z_ = Xn, Yn, Zn
        return line

    for d in DIG:
        out.append(dbl())
        if d:
            out.append(add(Q[0], Q[1] if d > 0 else F.f2_neg(Q[1])))
    if IS_BN:
        Y_ = F.f2_neg(Y_)
        q1 = G.g2_frob(Q)
        q2 = G.g2_frob(q1)
        out.append(add(q1[0], q1[1]))
        out.append(add(q2[0], F.f2_neg(q2[1])))
    return out


def g2_point():
    """a point of G2: the twist point from the first x = 1, 2, ... that works, cofactor cleared"""
    if IS_BN:
        h2 = 2 * CV.p - CV.r
    else:
        z = CV.z
        h2 = (z**8 - 4 * z**7 + 5 * z**6 - 4 * z**4 + 6 * z**3 - 4 * z**2 - 4 * z + 13) // 9
    x = (1, 0)
    while True:
        y = F.f2_sqrt(F.f2_add(F.f2_mul(F.f2_sqr(x), x), F.b2))
        if y is not None:
            R = None
            for bit in bin(h2)[2:]:
                R = G.g2_add(R, R)
                if bit == "1":
                    R = G.g2_add(R, (x, y))
            if R is not None:
                return R
        x = ((x[0] + 1) % P, 0)


def self_test():
    M = Mcl(CV)
    rnd = random.Random(2026)
    g1 = M.hash_to_g1(b"row16")
    gg = g2_point()
    lines = model_lines(gg)
    assert len(lines) == NLINES, (len(lines), NLINES)
    n_ok = 0
    for trial in range(4):
        a, k = rnd.randrange(1, CV.r), rnd.randrange(1, CV.r)
        sig1 = G.g1_mul(g1, a)
        K = G.g2_mul(gg, k)
        sig2 = G.g1_mul(sig1, k)                          # e(sig1, k gg) = e(k sig1, gg)
        good = simulate(sig1, G.g1_neg(sig2), K, lines)
        bad = simulate(sig1, G.g1_neg(G.g1_add(sig2, g1)), K, lines)
        assert good is True and bad is False, (trial, good, bad)
        n_ok += 1
    assert simulate(sig1, None, K, lines) is False        # e(sig1, K) != 1
    # the aggregated tail: F = the model's Miller value of (sig1, K) (w-basis coefficients), P2 = -sig2
    fm = G.miller_loop(sig1, K)
    Fv = [c_ for k_ in range(6) for c_ in fm[k_]]
    assert simulate(None, G.g1_neg(sig2), None, lines, prog=PROG_TAIL, Fval=Fv) is True
    assert simulate(None, G.g1_neg(G.g1_add(sig2, g1)), None, lines, prog=PROG_TAIL, Fval=Fv) is False
    assert simulate(None, None, None, lines, prog=PROG_TAIL, Fval=[1] + [0] * 11) is True
    assert simulate(None, None, K, lines) is True         # nothing live: the empty product
    # the Miller value itself against the model, after the easy part (which removes the subfield factors of the projective lines): checked through the verdicts above
    # and directly on one item
    return n_ok


# ---------------------------------------------------------------------------------------------------------------- emitter
def mont(v):
    return v * (1 << (LBITS * NLIMB)) % P


def balanced(x, nl=None, w=None):
    nl, w = nl or NLIMB, w or LBITS
    out = []
    for i in range(nl):
        if i == nl - 1:
            out.append(x)
            break
        d = x & ((1 << w) - 1)
        if d >= 1 << (w - 1):
            d -= 1 << w
        out.append(d)
        x = (x - d) >> w
    return out


def enc_term(a, b, c):
    assert 0 <= a < 256 and 0 <= b < 256 and -128 <= c < 128
    return a | (b << 8) | ((c & 0xFF) << 16)


def emit(path):
    L = []
    A = L.append
    A("// GENERATED by tools/gen_row16.py --curve %s -- do not edit.  Tables and program of the row-of-16 pairing check; see the generator for the formulas." % CURVE)
    A("#pragma once")
    A("#include <stdint.h>")
    A("#ifndef ROW16_DEV")
    A("#define ROW16_DEV static const      /* the device translation unit defines it as __constant__ */")
    A("#endif")
    A("namespace %s {" % ("row16" if IS_BN else "row16_bls"))
    A("constexpr int NSLOT = %d, NCONST = %d, CONST_BASE = %d, NSTEP = %d, NPROG = %d, NLINES = %d, NREG = %d;" % (NSLOT, len(CONSTS), CONST_BASE, len(STEPS), len(PROG), NLINES, NREG))
    A("constexpr int SLOT_T = %d, SLOT_Q = %d, SLOT_P1 = %d, SLOT_P2 = %d, SLOT_LF = %d, SLOT_W0 = %d, SLOT_X0 = %d, SLOT_W1 = %d, SLOT_ONE = %d, SLOT_E = %d;" % (T_, Q_, P1, P2, LF, W0, X0, W1, ONE, L1 + 4))
    A("constexpr int CONST_E_INIT = %d;      // four constants: E = 3 b' and E3 = 9 b' of a point with Z = 1" % E_INIT[0])
    A("constexpr int STEP_MUL_W0_W1 = %d;    // the step W0 <- W0 * W1 (also used on its own by the product tree of aggregated verification)" % s_mul.id)
    A("// constants in Montgomery form, balanced %d-bit limbs" % LBITS)
    A("ROW16_DEV int32_t CONSTS[NCONST][%d] = {" % NLIMB)
    for name, (idx, val) in sorted(CONSTS.items(), key=lambda kv: kv[1][0]):
        A("  {%s},  // %d %s" % (",".join(str(v) for v in balanced(mont(val))), idx, name))
    A("};")
    A("// per step type: number of terms (1, 2, 3, 4, 6, 8 or 12), flags (1 = also write xi * result to X0, tag << 1: 1 = variable pair, 2 = fixed pair)")
    A("ROW16_DEV uint8_t STEP_NT[NSTEP] = {%s};" % ",".join(str(s_.nt) for s_ in STEPS))
    A("ROW16_DEV uint8_t STEP_FLAGS[NSTEP] = {%s};" % ",".join(str((1 if s_.wx else 0) | (s_.tag << 1)) for s_ in STEPS))
    A("// terms: a-slot | b-slot << 8 | (signed 8-bit coefficient) << 16;  dest: slot | valid << 8 | gs << 9 | gs_plus << 10")
    A("ROW16_DEV uint32_t STEP_TERMS[NSTEP][16][12] = {")
    for s_ in STEPS:
        A("  {  // %d %s" % (s_.id, s_.name))
        for lane in range(NLANES):
            l = s_.lanes[lane]
            terms = [enc_term(a, b, c) for a, b, c in l[0]] if l else []
            terms += [enc_term(ZERO, ZERO, 1)] * (12 - len(terms))
            A("    {%s}," % ",".join("0x%x" % t for t in terms))
        A("  },")
    A("};")
    A("ROW16_DEV uint16_t STEP_DEST[NSTEP][16] = {")
    for s_ in STEPS:
        row = []
        for lane in range(NLANES):
            l = s_.lanes[lane]
            if not l:
                row.append(0)
            else:
                v = l[1] | (1 << 8)
                if l[2]:
                    assert l[2][0] == "gs" and l[2][1] == l[1]
                    v |= (1 << 9) | ((1 << 10) if l[2][2] > 0 else 0)
                row.append(v)
        A("  {%s}," % ",".join("0x%x" % v for v in row))
    A("};")
    A("// program: op | arg0 << 8 | arg1 << 16;  op 0 = dot(step), 1 = load fixed line(n), 2 = invert(src, dst), 3 = load register(reg, area), 4 = store register(reg), 5 = check")
    def enc_prog(prog):
        ops = []
        for ent in prog:
            if ent[0] == "dot":
                ops.append(0 | (ent[1] << 8))
            elif ent[0] == "line":
                ops.append(1 | (ent[1] << 8))
            elif ent[0] == "inv":
                ops.append(2 | (ent[1] << 8) | (ent[2] << 16))
            elif ent[0] == "ld":
                ops.append(3 | (ent[1] << 8) | (ent[2] << 16))
            elif ent[0] == "st":
                ops.append(4 | (ent[1] << 8))
            elif ent[0] == "ldf":
                ops.append(6)
            else:
                ops.append(5)
        return ops

    A("ROW16_DEV uint32_t PROG[NPROG] = {%s};" % ",".join("0x%x" % o for o in enc_prog(PROG)))
    A("// the closing step of aggregated verification: op 6 = load F (the product of the batch's Miller values) into W1")
    A("constexpr int NPROG_TAIL = %d;" % len(PROG_TAIL))
    A("ROW16_DEV uint32_t PROG_TAIL[NPROG_TAIL] = {%s};" % ",".join("0x%x" % o for o in enc_prog(PROG_TAIL)))
    A("}  // namespace")
    with open(path, "w") as fh:
        fh.write("\n".join(L) + "\n")
    print("wrote %s: %d step types, %d program entries, %d constants, %d slots per row" % (path, len(STEPS), len(PROG), len(CONSTS), NSLOT))


if __name__ == "__main__":
    n = self_test()
    print("simulator: %d valid + %d invalid signatures agree with the model's pairing semantics" % (n, n))
    target = os.path.join(ROOT, "ps-signature-and-el-passo_amd", "csrc", "elpasso_pair16_prog.h" if IS_BN else "elpasso_pair16_prog_bls12_381.h")
    if "--check" in sys.argv:      # the committed header must be what the (simulated) tables emit: tests/test_row16_gen.py
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            tmp = os.path.join(td, "prog.h")
            emit(tmp)
            same = open(tmp).read() == open(target).read()
        print("committed header %s the generator's output" % ("equals" if same else "DIFFERS from"))
        sys.exit(0 if same else 1)
    emit(target)


# ---------------------------------------------------------------------------------------------------------------- debugging aid: tools/pair16_check.hip
def debug_vectors(path):
    """Test vectors for tools/pair16_check.hip (gg, K, sig1, sig2 in std form) and, per chosen program counter, the simulator's slots: the tool dumps the device's slots
    at the same counters and tools/gen_row16.py --compare <dump> reports the first difference."""
    M = Mcl(CV)
    rnd = random.Random(7)
    g1 = M.hash_to_g1(b"row16")
    x = (1, 0)
    while True:
        y = F.f2_sqrt(F.f2_add(F.f2_mul(F.f2_sqr(x), x), F.b2))
        if y is not None:
            R = None
            for bit in bin(2 * CV.p - CV.r)[2:]:
                R = G.g2_add(R, R)
                if bit == "1":
                    R = G.g2_add(R, (x, y))
            if R is not None:
                gg = R
                break
        x = ((x[0] + 1) % P, 0)
    a, k = rnd.randrange(1, CV.r), rnd.randrange(1, CV.r)
    sig1 = G.g1_mul(g1, a)
    K = G.g2_mul(gg, k)
    sig2 = G.g1_mul(sig1, k)
    return gg, K, sig1, sig2


def words(v, n=8):
    return ",".join("0x%08xu" % ((v >> (32 * i)) & 0xFFFFFFFF) for i in range(n))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--vectors":
    gg, K, sig1, sig2 = debug_vectors(None)
    bad2 = G.g1_add(sig2, sig1)
    out = ["// GENERATED by tools/gen_row16.py --vectors: inputs of tools/pair16_check.hip (std form, 8 words per base-field value)",
           "static const unsigned V_GG[32] = {%s};" % ",".join(words(c) for c in (gg[0][0], gg[0][1], gg[1][0], gg[1][1])),
           "static const unsigned V_K[32] = {%s};" % ",".join(words(c) for c in (K[0][0], K[0][1], K[1][0], K[1][1])),
           "static const unsigned V_SIG1[16] = {%s};" % ",".join(words(c) for c in sig1),
           "static const unsigned V_SIG2[16] = {%s};" % ",".join(words(c) for c in sig2),
           "static const unsigned V_BAD2[16] = {%s};" % ",".join(words(c) for c in bad2)]
    open(os.path.join(ROOT, "tools", "pair16_vectors.h"), "w").write("\n".join(out) + "\n")
    print("wrote tools/pair16_vectors.h")

if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[1] == "--compare":
    # dump lines: "pc <pc> slot <s> <hex value>" for row 0 (the valid item); compare with the simulator's state before entry pc
    gg, K, sig1, sig2 = debug_vectors(None)
    lines = model_lines(gg)
    trace = []
    simulate(sig1, G.g1_neg(sig2), K, lines, trace=trace)
    # trace[j] = state AFTER entry j  ->  state before entry pc = trace[pc - 1]
    dump = {}
    for ln in open(sys.argv[2]):
        t = ln.split()
        if len(t) == 5 and t[0] == "pc":
            dump.setdefault(int(t[1]), {})[int(t[3])] = int(t[4], 16)
    inv_names = {v: k for k, v in NAMES.items()}
    for pc in sorted(dump):
        want = trace[pc - 1][1] if pc > 0 else None
        if want is None:
            continue
        nbad = 0
        for s_, got in sorted(dump[pc].items()):
            if s_ < NSLOT and got != want[s_]:
                base = max(b for b in inv_names if b <= s_)
                if nbad < 12:
                    print("pc %d (before %s): slot %d (%s+%d) differs: device %x, simulator %x" % (pc, PROG[pc], s_, inv_names[base], s_ - base, got, want[s_]))
                nbad += 1
        print("pc %d: %d differing slots of %d" % (pc, nbad, len(dump[pc])))
