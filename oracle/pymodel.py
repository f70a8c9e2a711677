"""Big-integer Python model of the EL PASSO / PS-signature hot path (TEST INFRASTRUCTURE ONLY).

This file is part of the *oracle*: it may be imported only by tests/, tools under oracle/,
`__graft_entry__.smoke()` and bench.py's cpu_baseline leg.  It is never on the product path.

What it restates (all citations into /root/reference):
  * protocol layer: src/ps-verifier.cc:13-229, src/ps-signer.cc:29-146, src/ps-requester.cc:19-432
  * wire codec:      src/ps-encoding.cc:5-121 (base64), :123-382 (TLV), :384-489 (messages)
  * arithmetic:      third-parties/mcl (herumi/mcl, un-vendored submodule, .gitmodules:1-4, pinned
                     version unknown, approx. v1.2x).  mcl's *published conventions* are restated here
                     (BN254 "mcl default" curve, little-endian serialisation with y-parity flag,
                     Fr::setHashOf masking, BN Shallue-van de Woestijne hashAndMapToG1).
Pinning: every convention below is checked against golden vectors captured from the reference's own
prebuilt wasm (tests/golden/bn254_*.json, produced by oracle/gen_fixtures.js) in tests/test_oracle_golden.py.

The model is deliberately simple (affine formulas, generic square-and-multiply, no windows); it is the
slow-but-obvious statement that the C oracle (oracle/elp_oracle.c) and the HIP path are checked against.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

# --------------------------------------------------------------------------------------------
# Curve parameters
# --------------------------------------------------------------------------------------------


class Curve:
    """Pairing-friendly curve y^2 = x^3 + b with sextic twist over Fp2 = Fp[i]/(i^2+1), xi = 1+i."""

    def __init__(self, name, z, p, r, b, twist, fbytes, ate_loop, is_bn):
        self.name, self.z, self.p, self.r, self.b = name, z, p, r, b
        self.twist = twist            # 'D' : b' = b/xi   'M' : b' = b*xi
        self.fbytes = fbytes          # serialised Fp size
        self.ate_loop = ate_loop      # |6z+2| (BN) or |z| (BLS12)
        self.is_bn = is_bn
        self.xi = (1, 1)


def _bn_params(z):
    p = 36 * z**4 + 36 * z**3 + 24 * z**2 + 6 * z + 1
    r = 36 * z**4 + 36 * z**3 + 18 * z**2 + 6 * z + 1
    return p, r


_z_bn = -0x4080000000000001
_p_bn, _r_bn = _bn_params(_z_bn)
assert _p_bn == 0x2523648240000001BA344D80000000086121000000000013A700000000000013
assert _r_bn == 0x2523648240000001BA344D8000000007FF9F800000000010A10000000000000D
BN254 = Curve("BN254", _z_bn, _p_bn, _r_bn, 2, "D", 32, abs(6 * _z_bn + 2), True)

_z_bls = -0xD201000000010000
_r_bls = _z_bls**4 - _z_bls**2 + 1
_p_bls = ((_z_bls - 1) ** 2 * _r_bls) // 3 + _z_bls
assert _p_bls == 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
BLS12_381 = Curve("BLS12_381", _z_bls, _p_bls, _r_bls, 4, "M", 48, abs(_z_bls), False)

# --------------------------------------------------------------------------------------------
# Field towers.  Fp2 elements are tuples (a, b) = a + b i.  Fp12 = Fp2[w]/(w^6 - xi): list of 6 Fp2.
# --------------------------------------------------------------------------------------------


class Field:
    def __init__(self, cv: Curve):
        self.cv = cv
        self.p = cv.p
        p = cv.p
        # Frobenius constants gamma_k = xi^(k (p-1)/6)
        self.gamma = [self.f2_pow(cv.xi, k * (p - 1) // 6) for k in range(6)]
        # twist coefficient
        if cv.twist == "D":
            self.b2 = self.f2_mul((cv.b, 0), self.f2_inv(cv.xi))
        else:
            self.b2 = self.f2_mul((cv.b, 0), cv.xi)

    # ---- Fp
    def inv(self, a):
        return pow(a, -1, self.p)

    def sqrt(self, a):
        """p = 3 mod 4 square root; returns None if a is not a square."""
        p = self.p
        a %= p
        y = pow(a, (p + 1) // 4, p)
        return y if y * y % p == a else None

    def legendre(self, a):
        a %= self.p
        if a == 0:
            return 0
        return 1 if pow(a, (self.p - 1) // 2, self.p) == 1 else -1

    # ---- Fp2
    def f2_add(self, x, y):
        return ((x[0] + y[0]) % self.p, (x[1] + y[1]) % self.p)

    def f2_sub(self, x, y):
        return ((x[0] - y[0]) % self.p, (x[1] - y[1]) % self.p)

    def f2_neg(self, x):
        return ((-x[0]) % self.p, (-x[1]) % self.p)

    def f2_mul(self, x, y):
        p = self.p
        return ((x[0] * y[0] - x[1] * y[1]) % p, (x[0] * y[1] + x[1] * y[0]) % p)

    def f2_sqr(self, x):
        return self.f2_mul(x, x)

    def f2_muls(self, x, s):
        return (x[0] * s % self.p, x[1] * s % self.p)

    def f2_conj(self, x):
        return (x[0], (-x[1]) % self.p)

    def f2_inv(self, x):
        p = self.p
        n = pow(x[0] * x[0] + x[1] * x[1], -1, p)
        return (x[0] * n % p, (-x[1]) * n % p)

    def f2_pow(self, x, e):
        r = (1, 0)
        while e:
            if e & 1:
                r = self.f2_mul(r, x)
            x = self.f2_mul(x, x)
            e >>= 1
        return r

    def f2_mul_xi(self, x):
        # (a + b i)(1 + i) = (a - b) + (a + b) i
        return ((x[0] - x[1]) % self.p, (x[0] + x[1]) % self.p)

    def f2_is_zero(self, x):
        return x[0] % self.p == 0 and x[1] % self.p == 0

    def f2_sqrt(self, a):
        """Square root in Fp2 (p = 3 mod 4), complex method. Returns None if not a square."""
        p = self.p
        a0, a1 = a[0] % p, a[1] % p
        if a1 == 0:
            s = self.sqrt(a0)
            if s is not None:
                return (s, 0)
            s = self.sqrt((-a0) % p)
            return (0, s)            # (s i)^2 = -s^2 = a0
        n = self.sqrt((a0 * a0 + a1 * a1) % p)
        if n is None:
            return None
        inv2 = self.inv(2)
        t = (a0 + n) * inv2 % p
        x = self.sqrt(t)
        if x is None:
            t = (a0 - n) * inv2 % p
            x = self.sqrt(t)
            if x is None:
                return None
        y = a1 * self.inv(2 * x % p) % p
        r = (x, y)
        assert self.f2_sqr(r) == (a0, a1)
        return r

    # ---- Fp12 = Fp2[w]/(w^6 - xi)
    def f12_one(self):
        return [(1, 0)] + [(0, 0)] * 5

    def f12_mul(self, x, y):
        acc = [(0, 0)] * 11
        for i in range(6):
            if x[i] == (0, 0):
                continue
            for j in range(6):
                if y[j] == (0, 0):
                    continue
                acc[i + j] = self.f2_add(acc[i + j], self.f2_mul(x[i], y[j]))
        out = list(acc[:6])
        for k in range(6, 11):
            out[k - 6] = self.f2_add(out[k - 6], self.f2_mul_xi(acc[k]))
        return out

    def f12_sqr(self, x):
        return self.f12_mul(x, x)

    def f12_conj(self, x):
        """x^(p^6): w -> -w."""
        return [x[k] if k % 2 == 0 else self.f2_neg(x[k]) for k in range(6)]

    def f12_frob(self, x, n=1):
        for _ in range(n):
            x = [self.f2_mul(self.f2_conj(x[k]), self.gamma[k]) for k in range(6)]
        return x

    def f12_inv(self, x):
        # x = a + b w with a,b in Fp6 = Fp2[v]/(v^3 - xi), v = w^2.  1/x = (a - b w)/(a^2 - v b^2)
        a = [x[0], x[2], x[4]]
        b = [x[1], x[3], x[5]]
        a2 = self.f6_mul(a, a)
        b2 = self.f6_mul(b, b)
        vb2 = [self.f2_mul_xi(b2[2]), b2[0], b2[1]]
        d = [self.f2_sub(a2[i], vb2[i]) for i in range(3)]
        di = self.f6_inv(d)
        ra = self.f6_mul(a, di)
        rb = self.f6_mul(b, di)
        return [ra[0], self.f2_neg(rb[0]), ra[1], self.f2_neg(rb[1]), ra[2], self.f2_neg(rb[2])]

    def f6_mul(self, x, y):
        acc = [(0, 0)] * 5
        for i in range(3):
            for j in range(3):
                acc[i + j] = self.f2_add(acc[i + j], self.f2_mul(x[i], y[j]))
        return [self.f2_add(acc[0], self.f2_mul_xi(acc[3])), self.f2_add(acc[1], self.f2_mul_xi(acc[4])), acc[2]]

    def f6_inv(self, x):
        a, b, c = x
        m = self.f2_mul
        A = self.f2_sub(m(a, a), self.f2_mul_xi(m(b, c)))
        B = self.f2_sub(self.f2_mul_xi(m(c, c)), m(a, b))
        C = self.f2_sub(m(b, b), m(a, c))
        F = self.f2_add(m(a, A), self.f2_mul_xi(self.f2_add(m(c, B), m(b, C))))
        Fi = self.f2_inv(F)
        return [m(A, Fi), m(B, Fi), m(C, Fi)]

    def f12_pow(self, x, e):
        r = self.f12_one()
        for bit in bin(e)[2:]:
            r = self.f12_sqr(r)
            if bit == "1":
                r = self.f12_mul(r, x)
        return r


# --------------------------------------------------------------------------------------------
# Groups (affine, None = infinity).  G1 coords are ints, G2 coords are Fp2 tuples.
# --------------------------------------------------------------------------------------------


class Groups:
    def __init__(self, cv: Curve):
        self.cv = cv
        self.F = Field(cv)
        self.p = cv.p
        if not cv.is_bn:
            # GLV data of G1 (see g1_mul): L = z^2 - 1 satisfies L^2 + L + 1 = 0 mod r; beta = the cube root of unity in Fp with (beta x, y) = [L](x, y) on G1
            self.glv_L = cv.z * cv.z - 1
            assert (self.glv_L * self.glv_L + self.glv_L + 1) % cv.r == 0
            p = cv.p
            g = 2
            while pow(g, (p - 1) // 3, p) == 1:
                g += 1
            w = pow(g, (p - 1) // 3, p)
            # a point of G1: clear the cofactor of the first curve point found from x = 1, 2, ...
            x = 1
            while True:
                y = self.F.sqrt((x * x * x + cv.b) % p)
                if y is not None:
                    Q = self.g1_mul_plain((x, y), (cv.z - 1) ** 2 // 3)
                    if Q is not None:
                        break
                x += 1
            want = self.g1_mul_plain(Q, self.glv_L)
            self.glv_beta = w if (w * Q[0] % p, Q[1]) == want else w * w % p
            assert (self.glv_beta * Q[0] % p, Q[1]) == want

    # ---- G1
    def g1_on_curve(self, P):
        if P is None:
            return True
        x, y = P
        return (y * y - x * x * x - self.cv.b) % self.p == 0

    def g1_neg(self, P):
        return None if P is None else (P[0], (-P[1]) % self.p)

    def g1_add(self, P, Q):
        p = self.p
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return (x3, (lam * (x1 - x3) - y1) % p)

    def g1_mul(self, P, k):
        """G1::mul as mcl evaluates it.  On BN254 E(Fp) = G1 and any correct method gives [k mod r]P.  On BLS12-381 mcl's G1::mul is the GLV
        method with the endomorphism psi(x, y) = (beta x, y) = [L] on G1, L = z^2 - 1: k mod r is split by plain division, b, a = divmod(k, L),
        and the result is [a]P + [b]psi(P).  Inside G1 that is [k]P; for a point of E(Fp) OUTSIDE G1 (which mcl's default accepts on
        deserialisation) it is not -- e.g. psi fixes the points (0, +-2) of order 3, so the order-3 component is multiplied by a + b instead of
        k.  The split is pinned by the reference wasm's verdicts on 20 crafted proofs / requests (tests/golden/bls12_381_oracle_edge.json
        "crafted_phi_c_mod_3", bls12_381_oracle_requests.json "crafted_A_c_mod_3": accepted iff 3 | a + b)."""
        k %= self.cv.r
        if self.cv.is_bn or P is None:
            return self.g1_mul_plain(P, k)
        b, a = divmod(k, self.glv_L)
        psiP = (self.glv_beta * P[0] % self.p, P[1])
        return self.g1_add(self.g1_mul_plain(P, a), self.g1_mul_plain(psiP, b))

    def g1_mul_plain(self, P, k):          # [k]P for any integer k >= 0, no reduction modulo r (points outside the order-r subgroup)
        R = None
        for bit in bin(k)[2:] if k else "":
            R = self.g1_add(R, R)
            if bit == "1":
                R = self.g1_add(R, P)
        return R

    # ---- G2 (on the twist E'(Fp2))
    def g2_on_curve(self, P):
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.f2_sub(F.f2_sqr(y), F.f2_add(F.f2_mul(F.f2_sqr(x), x), F.b2)) == (0, 0)

    def g2_neg(self, P):
        return None if P is None else (P[0], self.F.f2_neg(P[1]))

    def g2_add(self, P, Q):
        F = self.F
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if F.f2_add(y1, y2) == (0, 0):
                return None
            lam = F.f2_mul(F.f2_muls(F.f2_sqr(x1), 3), F.f2_inv(F.f2_muls(y1, 2)))
        else:
            lam = F.f2_mul(F.f2_sub(y2, y1), F.f2_inv(F.f2_sub(x2, x1)))
        x3 = F.f2_sub(F.f2_sub(F.f2_sqr(lam), x1), x2)
        return (x3, F.f2_sub(F.f2_mul(lam, F.f2_sub(x1, x3)), y1))

    def g2_mul(self, P, k):
        k %= self.cv.r
        R = None
        for bit in bin(k)[2:] if k else "":
            R = self.g2_add(R, R)
            if bit == "1":
                R = self.g2_add(R, P)
        return R

    def g2_frob(self, Q):
        """pi_p on the twist: (x, y) -> (conj(x) * gamma_2', conj(y) * gamma_3') (twist-dependent)."""
        F = self.F
        if Q is None:
            return None
        x, y = Q
        if self.cv.twist == "D":
            # untwist (x w^2, y w^3); Frobenius; twist back
            return (F.f2_mul(F.f2_conj(x), F.gamma[2]), F.f2_mul(F.f2_conj(y), F.gamma[3]))
        # M-type: untwist (x / w^2, y / w^3)
        return (F.f2_mul(F.f2_conj(x), F.f2_inv(F.gamma[2])), F.f2_mul(F.f2_conj(y), F.f2_inv(F.gamma[3])))

    # ---- pairing: optimal ate, affine Miller loop on the twist, sparse lines embedded in Fp12
    def _line(self, T, lam, P):
        """Line through T with twist-slope lam evaluated at P in G1 -> sparse Fp12 (list of 6 Fp2).

        D-type untwist psi(x', y') = (x' w^2, y' w^3):  l(P) = y_P - lam x_P w + (lam x_T - y_T) w^3
        M-type untwist psi(x', y') = (x'/w^2, y'/w^3); multiplying the line by w^3 (killed by the final
        exponentiation, w^3 lies in Fp4... we instead scale by xi = w^6 to stay exact):
            l(P) * w^3 = y_P w^3 - lam x_P w^2 + (lam x_T - y_T)
        """
        F = self.F
        xP, yP = P
        xT, yT = T
        c = F.f2_sub(F.f2_mul(lam, xT), yT)
        out = [(0, 0)] * 6
        if self.cv.twist == "D":
            out[0] = (yP % self.p, 0)
            out[1] = F.f2_neg(F.f2_muls(lam, xP))
            out[3] = c
        else:
            out[3] = (yP % self.p, 0)
            out[2] = F.f2_neg(F.f2_muls(lam, xP))
            out[0] = c
        return out

    def _dbl_step(self, T, P):
        F = self.F
        x, y = T
        lam = F.f2_mul(F.f2_muls(F.f2_sqr(x), 3), F.f2_inv(F.f2_muls(y, 2)))
        l = self._line(T, lam, P)
        return self.g2_add(T, T), l

    def _add_step(self, T, Q, P):
        F = self.F
        lam = F.f2_mul(F.f2_sub(Q[1], T[1]), F.f2_inv(F.f2_sub(Q[0], T[0])))
        l = self._line(T, lam, P)
        return self.g2_add(T, Q), l

    def miller_loop(self, P, Q):
        """f_{s,Q}(P) for the optimal ate pairing (s = 6z+2 for BN, z for BLS12)."""
        F = self.F
        if P is None or Q is None:
            return F.f12_one()
        s = self.cv.ate_loop
        f = F.f12_one()
        T = Q
        for bit in bin(s)[3:]:
            T, l = self._dbl_step(T, P)
            f = F.f12_mul(F.f12_sqr(f), l)
            if bit == "1":
                T, l = self._add_step(T, Q, P)
                f = F.f12_mul(f, l)
        if self.cv.z < 0:
            T = self.g2_neg(T)
            f = F.f12_conj(f)
        if self.cv.is_bn:
            Q1 = self.g2_frob(Q)
            Q2 = self.g2_neg(self.g2_frob(Q1))
            T, l = self._add_step(T, Q1, P)
            f = F.f12_mul(f, l)
            T, l = self._add_step(T, Q2, P)
            f = F.f12_mul(f, l)
        return f

    def final_exp(self, f):
        F = self.F
        p, r = self.cv.p, self.cv.r
        # easy part: f^((p^6-1)(p^2+1))
        t = F.f12_mul(F.f12_conj(f), F.f12_inv(f))
        t = F.f12_mul(F.f12_frob(t, 2), t)
        # hard part: plain exponentiation by (p^4 - p^2 + 1)/r
        return F.f12_pow(t, (p**4 - p**2 + 1) // r)

    def pairing(self, P, Q):
        return self.final_exp(self.miller_loop(P, Q))

    def pairing_check(self, pairs):
        """prod e(P_i, Q_i) == 1 with a single final exponentiation."""
        F = self.F
        f = F.f12_one()
        for P, Q in pairs:
            f = F.f12_mul(f, self.miller_loop(P, Q))
        return self.final_exp(f) == F.f12_one()


# --------------------------------------------------------------------------------------------
# mcl conventions (BN254): serialisation, hashing
# --------------------------------------------------------------------------------------------


class Mcl:
    """mcl-compatible encodings and hashes on top of Groups (BN254 pinned by golden vectors)."""

    def __init__(self, cv: Curve = BN254):
        self.cv = cv
        self.G = Groups(cv)
        self.F = self.G.F
        self.p, self.r = cv.p, cv.r
        self.fb = cv.fbytes
        F = self.F
        # SvdW constants for y^2 = x^3 + b (mcl MapTo: the same Shallue-van de Woestijne routine serves BN254 and, in mcl's default
        # (non-ETH) mode, BLS12-381): c1 = sqrt(-3) = (-3)^((p+1)/4), c2 = (c1 - 1)/2
        c1 = F.sqrt((-3) % self.p)
        self.c1 = c1
        self.c2 = (c1 - 1) * F.inv(2) % self.p
        if cv.is_bn:
            # pinned by tests/golden/bn254_*.json
            assert self.c1 == 0x252364824000000126CD890000000003CF0F0000000000060C00000000000004
            assert self.c2 == 0x25236482400000017080EB4000000006181800000000000CD98000000000000B
            self.g1_cofactor = 1
        else:
            # pinned by tests/golden/bls12_381_*.json (reference wasm with mcl's BLS12-381 CurveParam, oracle/wasm_curve.js)
            assert self.c1 == 0xBE32CE5FBEED9CA374D38C0ED41EEFD5BB675277CDF12D11BC2FB026C41400045C03FFFFFFFDFFFD
            self.g1_cofactor = (cv.z - 1) ** 2 // 3

    # ---- scalars
    def fr_ser(self, x):
        return int(x % self.r).to_bytes(32, "little")

    def fr_de(self, b):
        return int.from_bytes(b, "little")

    def _set_hash_of(self, msg: bytes, mod: int) -> int:
        """Fr/Fp::setHashOf: SHA-256 (modulus of at most 256 bits) or SHA-512 (wider: the 381-bit Fp of BLS12-381) -> the first
        ceil(bitlen/8) bytes as an LE integer -> mask to bitlen(mod) bits -> if >= mod clear the top bit."""
        nb = mod.bit_length()
        d = hashlib.sha256(msg).digest() if nb <= 256 else hashlib.sha512(msg).digest()
        h = int.from_bytes(d[:(nb + 7) // 8], "little")
        h &= (1 << nb) - 1
        if h >= mod:
            h &= (1 << (nb - 1)) - 1
        return h

    def fr_hash(self, msg) -> int:
        if isinstance(msg, str):
            msg = msg.encode()
        return self._set_hash_of(msg, self.r)

    def fp_hash(self, msg) -> int:
        if isinstance(msg, str):
            msg = msg.encode()
        return self._set_hash_of(msg, self.p)

    # ---- points
    def g1_ser(self, P) -> bytes:
        n = self.fb
        if P is None:
            return bytes(n)
        x, y = P
        b = bytearray(int(x).to_bytes(n, "little"))
        if y & 1:
            b[n - 1] |= 0x80
        return bytes(b)

    def g1_de(self, bs: bytes):
        n = self.fb
        bs = bytes(bs)
        if bs == bytes(n):
            return None
        odd = bool(bs[n - 1] & 0x80)
        x = int.from_bytes(bs, "little") & ((1 << (8 * n - 1)) - 1)
        if x >= self.p:
            raise ValueError("G1 x out of range")
        y = self.F.sqrt((x * x * x + self.cv.b) % self.p)
        if y is None:
            raise ValueError("G1 x not on curve")
        if bool(y & 1) != odd:
            y = self.p - y
        return (x, y)

    def g2_ser(self, P) -> bytes:
        n = self.fb
        if P is None:
            return bytes(2 * n)
        (xa, xb), (ya, yb) = P
        b = bytearray(int(xa).to_bytes(n, "little") + int(xb).to_bytes(n, "little"))
        if ya & 1:
            b[2 * n - 1] |= 0x80
        return bytes(b)

    def g2_de(self, bs: bytes):
        n = self.fb
        bs = bytes(bs)
        if bs == bytes(2 * n):
            return None
        odd = bool(bs[2 * n - 1] & 0x80)
        xa = int.from_bytes(bs[:n], "little")
        xb = int.from_bytes(bs[n:], "little") & ((1 << (8 * n - 1)) - 1)
        F = self.F
        x = (xa, xb)
        y = F.f2_sqrt(F.f2_add(F.f2_mul(F.f2_sqr(x), x), F.b2))
        if y is None:
            raise ValueError("G2 x not on curve")
        if bool(y[0] & 1) != odd:
            y = F.f2_neg(y)
        return (x, y)

    def g1_hex(self, P) -> str:
        return self.g1_ser(P).hex()

    def g2_hex(self, P) -> str:
        return self.g2_ser(P).hex()

    def g1_getstr(self, P) -> str:
        """G1::getStr() (decimal): "1 <x> <y>" ; "0" for infinity (src/ps-verifier.cc:231-235)."""
        if P is None:
            return "0"
        return "1 %d %d" % (P[0], P[1])

    # ---- hash to G1: mcl's hashAndMapToG1 = Fp::setHashOf -> Shallue-van de Woestijne map (MapTo::calcBN) -> cofactor clearing
    #      (cofactor 1 on BN254; (z-1)^2/3 on BLS12-381).  Call sites: src/ps-verifier.cc:94,186, src/ps-requester.cc:185,336.
    def hash_to_g1(self, msg):
        if isinstance(msg, str):
            msg = msg.encode()
        P = self.map_to_g1(self.fp_hash(msg))
        if self.g1_cofactor != 1:
            P = self.G.g1_mul_plain(P, self.g1_cofactor)
        return P

    def map_to_g1(self, t):
        p, b = self.p, self.cv.b
        F = self.F
        t %= p
        if t == 0:
            raise ValueError("t = 0")
        neg = F.legendre(t) < 0
        w = (t * t + b + 1) % p
        w = self.c1 * t % p * F.inv(w) % p
        for i in range(3):
            if i == 0:
                x = (self.c2 - t * w) % p
            elif i == 1:
                x = (-x - 1) % p
            else:
                x = (1 + F.inv(w * w % p)) % p
            y = F.sqrt((x * x * x + b) % p)
            if y is not None:
                if neg:
                    y = (-y) % p
                self._last_branch = (i, neg)
                return (x, y)
        raise AssertionError("unreachable")

    # ---- SHA-256 helpers (cybozu::Sha256 update/digest = plain SHA-256)
    @staticmethod
    def sha256(data: bytes) -> bytes:
        return hashlib.sha256(data).digest()

    def challenge(self, hex_parts: Sequence[str], ad: str) -> int:
        """Fr::setHashOf(SHA256(hex_1 || ... || hex_n || ad)) i.e. SHA-256 applied twice
        (src/ps-verifier.cc:111-122, src/ps-signer.cc:96-101)."""
        m = "".join(hex_parts).encode() + ad.encode()
        return self.fr_hash(self.sha256(m))


# --------------------------------------------------------------------------------------------
# Wire codec (src/ps-encoding.cc)
# --------------------------------------------------------------------------------------------

T_G1, T_G2, T_FR, T_G1L, T_G2L, T_FRL, T_STRL = 1, 2, 3, 4, 5, 6, 7


class Buf:
    def __init__(self, data: bytes = b""):
        self.b = bytearray(data)

    # append
    def var(self, n):
        if n < 253:
            self.b.append(n)
        elif n <= 0xFFFF:
            self.b += bytes([253, (n >> 8) & 0xFF, n & 0xFF])

    def elem(self, t, raw, with_type=True):
        if with_type:
            self.b.append(t)
        self.var(len(raw))
        self.b += raw

    def lst(self, t, raws):
        self.b.append(t)
        self.var(len(raws))
        for r in raws:
            self.var(len(r))
            self.b += r

    # parse
    def pvar(self, off):
        f = self.b[off]
        if f < 253:
            return f, 1
        if f == 253:
            return (self.b[off + 1] << 8) | self.b[off + 2], 3
        return 0, 0

    def pelem(self, off, t):
        if self.b[off] != t:
            raise ValueError("type mismatch at %d: %d != %d" % (off, self.b[off], t))
        n, s = self.pvar(off + 1)
        st = off + 1 + s
        return bytes(self.b[st:st + n]), st + n

    def plist(self, off, t):
        if self.b[off] != t:
            raise ValueError("type mismatch at %d" % off)
        cnt, s = self.pvar(off + 1)
        off = off + 1 + s
        out = []
        for _ in range(cnt):
            n, s = self.pvar(off)
            off += s
            out.append(bytes(self.b[off:off + n]))
            off += n
        return out, off


@dataclass
class PubKey:
    g: object
    gg: object
    XX: object
    Yi: list
    YYi: list


@dataclass
class Credential:
    sig1: object
    sig2: object


@dataclass
class CredRequest:
    A: object
    c: int
    rs: List[int]
    attributes: List[bytes]


@dataclass
class IdProof:
    sig1: object
    sig2: object
    k: object
    phi: object
    c: int
    rs: List[int]
    attributes: List[bytes]
    E1: object = None
    E2: object = None
    has_E: bool = False


class Codec:
    def __init__(self, m: Mcl):
        self.m = m

    def pk_decode(self, data: bytes) -> PubKey:
        m, b = self.m, Buf(data)
        g, o = b.pelem(0, T_G1)
        gg, o = b.pelem(o, T_G2)
        XX, o = b.pelem(o, T_G2)
        Yi, o = b.plist(o, T_G1L)
        YYi, o = b.plist(o, T_G2L)
        return PubKey(m.g1_de(g), m.g2_de(gg), m.g2_de(XX), [m.g1_de(x) for x in Yi], [m.g2_de(x) for x in YYi])

    def pk_encode(self, pk: PubKey) -> bytes:
        m, b = self.m, Buf()
        b.elem(T_G1, m.g1_ser(pk.g))
        b.elem(T_G2, m.g2_ser(pk.gg))
        b.elem(T_G2, m.g2_ser(pk.XX))
        b.lst(T_G1L, [m.g1_ser(x) for x in pk.Yi])
        b.lst(T_G2L, [m.g2_ser(x) for x in pk.YYi])
        return bytes(b.b)

    def cred_decode(self, data) -> Credential:
        m, b = self.m, Buf(data)
        s1, o = b.pelem(0, T_G1)
        s2, o = b.pelem(o, T_G1)
        return Credential(m.g1_de(s1), m.g1_de(s2))

    def cred_encode(self, c: Credential) -> bytes:
        m, b = self.m, Buf()
        b.elem(T_G1, m.g1_ser(c.sig1))
        b.elem(T_G1, m.g1_ser(c.sig2))
        return bytes(b.b)

    def req_decode(self, data) -> CredRequest:
        m, b = self.m, Buf(data)
        A, o = b.pelem(0, T_G1)
        c, o = b.pelem(o, T_FR)
        rs, o = b.plist(o, T_FRL)
        at, o = b.plist(o, T_STRL)
        return CredRequest(m.g1_de(A), m.fr_de(c), [m.fr_de(x) for x in rs], at)

    def req_encode(self, q: CredRequest) -> bytes:
        m, b = self.m, Buf()
        b.elem(T_G1, m.g1_ser(q.A))
        b.elem(T_FR, m.fr_ser(q.c))
        b.lst(T_FRL, [m.fr_ser(x) for x in q.rs])
        b.lst(T_STRL, list(q.attributes))
        return bytes(b.b)

    def proof_decode(self, data) -> IdProof:
        m, b = self.m, Buf(data)
        s1, o = b.pelem(0, T_G1)
        s2, o = b.pelem(o, T_G1)
        k, o = b.pelem(o, T_G2)
        phi, o = b.pelem(o, T_G1)
        c, o = b.pelem(o, T_FR)
        rs, o = b.plist(o, T_FRL)
        at, o = b.plist(o, T_STRL)
        pr = IdProof(m.g1_de(s1), m.g1_de(s2), m.g2_de(k), m.g1_de(phi), m.fr_de(c), [m.fr_de(x) for x in rs], at)
        if o < len(b.b):
            e1, o = b.pelem(o, T_G1)
            e2, o = b.pelem(o, T_G1)
            pr.E1, pr.E2, pr.has_E = m.g1_de(e1), m.g1_de(e2), True
        return pr

    def proof_encode(self, pr: IdProof) -> bytes:
        m, b = self.m, Buf()
        b.elem(T_G1, m.g1_ser(pr.sig1))
        b.elem(T_G1, m.g1_ser(pr.sig2))
        b.elem(T_G2, m.g2_ser(pr.k))
        b.elem(T_G1, m.g1_ser(pr.phi))
        b.elem(T_FR, m.fr_ser(pr.c))
        b.lst(T_FRL, [m.fr_ser(x) for x in pr.rs])
        b.lst(T_STRL, list(pr.attributes))
        if pr.has_E:
            b.elem(T_G1, m.g1_ser(pr.E1))
            b.elem(T_G1, m.g1_ser(pr.E2))
        return bytes(b.b)


# --------------------------------------------------------------------------------------------
# Protocol layer, reference structure (one scalar-mult per term, two pairings)
# --------------------------------------------------------------------------------------------


def _b(s):
    return s.encode() if isinstance(s, str) else bytes(s)


class Protocol:
    def __init__(self, m: Mcl):
        self.m = m
        self.G = m.G
        # the library's ELP_OPT_STRICT_SIGNATURE for el_passo_verify_id (include/elpasso.h): False = the reference's behaviour (sig1 = sig2 = O passes,
        # golden case "sig_both_zero"), True = sig1 must be admissible as in PSVerifier::verify
        self.strict = False
        # the library's ELP_OPT_SUBGROUP_CHECK (BLS12-381 only; default 1 in the library): True = prover-supplied G1 points must lie in the order-r
        # subgroup (project policy, a DELIBERATE divergence from the reference: mcl's default does not test the order of a deserialised G1 point --
        # what the reference's wasm answers on such inputs is recorded in tests/golden/bls12_381_oracle_edge.json), False = no test.
        self.subgroup_check = True

    # sigma_1 must not be the identity of G1 (src/ps-verifier.cc:16-18).  On a curve with a G1 cofactor (BLS12-381) that is asked of the order-r component:
    # a point of E(Fp) whose order divides the cofactor pairs to 1 with everything, so "sig1 is not the point at infinity" alone admits the forgery
    # sig1 = T (order 3), sig2 = O for any K.  Project policy (csrc/elp/pipeline.h sig1_admissible): sig1 != O and sig1 in G1.  A no-op beyond the
    # infinity test on BN254 (E(Fp) = G1), hence invisible to the golden vectors.
    def sig1_admissible(self, sig1) -> bool:
        return sig1 is not None and self.in_g1(sig1)

    # -- PSVerifier::verify (src/ps-verifier.cc:13-35)
    def ps_verify(self, pk: PubKey, cred: Credential, all_attributes) -> bool:
        m, G = self.m, self.G
        if not self.sig1_admissible(cred.sig1):
            return False
        K = pk.XX
        for i, a in enumerate(all_attributes):
            K = G.g2_add(K, G.g2_mul(pk.YYi[i], m.fr_hash(_b(a))))
        return G.pairing(cred.sig1, K) == G.pairing(cred.sig2, pk.gg)

    # -- prepare_hybrid_verification (src/ps-verifier.cc:214-229)
    def hybrid_k(self, pk, k, attributes):
        m, G = self.m, self.G
        K = k
        for i, a in enumerate(attributes):
            if len(a) == 0:
                continue
            K = G.g2_add(K, G.g2_mul(pk.YYi[i], m.fr_hash(_b(a))))
        return K

    def _vk(self, pk, pr: IdProof, rt):
        G = self.G
        V = G.g2_mul(pr.k, pr.c)
        j = 0
        for i, a in enumerate(pr.attributes):
            if len(a) == 0:
                V = G.g2_add(V, G.g2_mul(pk.YYi[i], pr.rs[j]))
                j += 1
        V = G.g2_add(V, G.g2_mul(pk.gg, rt))
        V = G.g2_add(V, G.g2_mul(pk.XX, (1 - pr.c) % self.m.r))
        return V

    # Project policy on curves whose G1 cofactor is not 1 (BLS12-381; a no-op on BN254 where E(Fp) = G1, hence invisible to the golden vectors):
    # prover-supplied G1 points must lie in the order-r subgroup (csrc/elp/pipeline.h g1_in_subgroup, include/elpasso.h ELP_OPT_SUBGROUP_CHECK).
    # Here by the definition, [r]P == O.
    def in_g1(self, P) -> bool:
        if not self.subgroup_check:
            return True
        return P is None or self.G.g1_mul_plain(P, self.m.r) is None

    # -- el_passo_verify_id (src/ps-verifier.cc:37-138)
    def verify_id(self, pk, pr: IdProof, ad, svc, authority_pk, g, h, pairing=True) -> bool:
        m, G = self.m, self.G
        if not pr.has_E:
            return False
        if self.strict and not self.sig1_admissible(pr.sig1):
            return False
        if not (self.in_g1(pr.phi) and self.in_g1(pr.E1) and self.in_g1(pr.E2)):
            return False
        Vk = self._vk(pk, pr, pr.rs[len(pr.rs) - 2])
        Hs = m.hash_to_g1(_b(svc))
        Vphi = G.g1_add(G.g1_mul(pr.phi, pr.c), G.g1_mul(Hs, pr.rs[0]))
        reps = pr.rs[len(pr.rs) - 1]
        VE1 = G.g1_add(G.g1_mul(pr.E1, pr.c), G.g1_mul(g, reps))
        VE2 = G.g1_add(G.g1_add(G.g1_mul(pr.E2, pr.c), G.g1_mul(authority_pk, reps)), G.g1_mul(h, pr.rs[1]))
        c2 = m.challenge([m.g2_hex(pr.k), m.g1_hex(pr.phi), m.g1_hex(pr.E1), m.g1_hex(pr.E2),
                          m.g2_hex(Vk), m.g1_hex(Vphi), m.g1_hex(VE1), m.g1_hex(VE2)], _b(ad).decode("latin1"))
        if c2 != pr.c:
            return False
        if not pairing:
            return True
        K = self.hybrid_k(pk, pr.k, pr.attributes)
        return G.pairing(pr.sig1, K) == G.pairing(pr.sig2, pk.gg)

    # -- el_passo_verify_id_without_id_retrieval (src/ps-verifier.cc:140-212)
    def verify_id_noretr(self, pk, pr: IdProof, ad, svc, pairing=True) -> bool:
        m, G = self.m, self.G
        if self.strict and not self.sig1_admissible(pr.sig1):
            return False
        if not self.in_g1(pr.phi):
            return False
        Vk = self._vk(pk, pr, pr.rs[len(pr.rs) - 1])
        Hs = m.hash_to_g1(_b(svc))
        Vphi = G.g1_add(G.g1_mul(pr.phi, pr.c), G.g1_mul(Hs, pr.rs[0]))
        c2 = m.challenge([m.g2_hex(pr.k), m.g1_hex(pr.phi), m.g2_hex(Vk), m.g1_hex(Vphi)], _b(ad).decode("latin1"))
        if c2 != pr.c:
            return False
        if not pairing:
            return True
        K = self.hybrid_k(pk, pr.k, pr.attributes)
        return G.pairing(pr.sig1, K) == G.pairing(pr.sig2, pk.gg)

    # -- PSSigner::el_passo_nizk_verify_request (src/ps-signer.cc:74-110)
    def nizk_verify_request(self, pk, rq: CredRequest, ad) -> bool:
        m, G = self.m, self.G
        if not self.in_g1(rq.A):
            return False
        V = G.g1_mul(rq.A, rq.c)
        V = G.g1_add(V, G.g1_mul(pk.g, rq.rs[0]))
        j = 1
        for i, a in enumerate(rq.attributes):
            if len(a) == 0:
                V = G.g1_add(V, G.g1_mul(pk.Yi[i], rq.rs[j]))
                j += 1
        c2 = m.challenge([m.g1_hex(rq.A), m.g1_hex(V)], _b(ad).decode("latin1"))
        return c2 == rq.c

    # -- PSSigner::sign_hybrid / sign_commitment (src/ps-signer.cc:112-146); nonce u injected
    def sign_hybrid(self, pk, sk_X, A, attributes, u):
        m, G = self.m, self.G
        fa = A
        if len(attributes) != 1:               # quirk src/ps-signer.cc:115-117
            for i, a in enumerate(attributes):
                if len(a) == 0:
                    continue
                fa = G.g1_add(fa, G.g1_mul(pk.Yi[i], m.fr_hash(_b(a))))
        return Credential(G.g1_mul(pk.g, u), G.g1_mul(G.g1_add(sk_X, fa), u))

    def provide_id(self, pk, sk_X, rq, ad, u):
        if not self.nizk_verify_request(pk, rq, ad):
            return None
        return self.sign_hybrid(pk, sk_X, rq.A, rq.attributes, u)

    # -- PSSigner::key_gen (src/ps-signer.cc:29-55); secrets injected
    def key_gen(self, g, gg, x, ys):
        G = self.G
        return PubKey(g, gg, G.g2_mul(gg, x), [G.g1_mul(g, y) for y in ys], [G.g2_mul(gg, y) for y in ys]), G.g1_mul(g, x)

    # -- PSRequester::el_passo_request_id (src/ps-requester.cc:19-99); rnd = [t1, rho_0, rho_hidden...]
    def request_id(self, pk, attrs: Sequence[Tuple[bytes, bool]], ad, rnd):
        m, G = self.m, self.G
        if len(attrs) != len(pk.Yi):
            raise RuntimeError("attribute size does not match")
        it = iter(rnd)
        t1 = next(it)
        A = G.g1_mul(pk.g, t1)
        rho0 = next(it)
        rhos = [rho0]
        V = G.g1_mul(pk.g, rho0)
        hs = []
        for i, (a, hide) in enumerate(attrs):
            if hide:
                ah = m.fr_hash(_b(a))
                hs.append(ah)
                A = G.g1_add(A, G.g1_mul(pk.Yi[i], ah))
                rho = next(it)
                rhos.append(rho)
                V = G.g1_add(V, G.g1_mul(pk.Yi[i], rho))
        c = m.challenge([m.g1_hex(A), m.g1_hex(V)], _b(ad).decode("latin1"))
        rs = [(rhos[0] - t1 * c) % m.r] + [(rhos[i + 1] - hs[i] * c) % m.r for i in range(len(hs))]
        return CredRequest(A, c, rs, [b"" if hide else _b(a) for a, hide in attrs]), t1

    def unblind(self, cred, t1):
        G = self.G
        return Credential(cred.sig1, G.g1_add(cred.sig2, G.g1_neg(G.g1_mul(cred.sig1, t1))))

    def randomize(self, cred, t):
        G = self.G
        return Credential(G.g1_mul(cred.sig1, t), G.g1_mul(cred.sig2, t))

    # -- PSRequester::el_passo_prove_id (src/ps-requester.cc:150-310)
    #    rnd = [t, r, epsilon, rho_hidden..., rho_t, rho_eps]
    def prove_id(self, pk, cred, attrs, ad, svc, authority_pk, g, h, rnd, with_retrieval=True):
        m, G = self.m, self.G
        if len(attrs) != len(pk.Yi):
            raise RuntimeError("attribute size does not match")
        it = iter(rnd)
        t, r = next(it), next(it)
        sig1 = G.g1_mul(cred.sig1, r)
        sig2 = G.g1_mul(G.g1_add(G.g1_mul(cred.sig1, t), cred.sig2), r)
        if with_retrieval:
            eps = next(it)
            gamma = m.fr_hash(_b(attrs[1][0]))
            E1 = G.g1_mul(g, eps)
            E2 = G.g1_add(G.g1_mul(authority_pk, eps), G.g1_mul(h, gamma))
        Hs = m.hash_to_g1(_b(svc))
        s = m.fr_hash(_b(attrs[0][0]))
        phi = G.g1_mul(Hs, s)
        k = pk.XX
        hs = []
        for i, (a, hide) in enumerate(attrs):
            if hide:
                ah = m.fr_hash(_b(a))
                hs.append(ah)
                k = G.g2_add(k, G.g2_mul(pk.YYi[i], ah))
        k = G.g2_add(k, G.g2_mul(pk.gg, t))
        Vk = pk.XX
        rhos = []
        for i, (a, hide) in enumerate(attrs):
            if hide:
                rho = next(it)
                rhos.append(rho)
                Vk = G.g2_add(Vk, G.g2_mul(pk.YYi[i], rho))
        rho_t = next(it)
        rhos.append(rho_t)
        Vk = G.g2_add(Vk, G.g2_mul(pk.gg, rho_t))
        Vphi = G.g1_mul(Hs, rhos[0])
        if with_retrieval:
            rho_e = next(it)
            rhos.append(rho_e)
            VE1 = G.g1_mul(g, rho_e)
            VE2 = G.g1_add(G.g1_mul(authority_pk, rho_e), G.g1_mul(h, rhos[1]))
            c = m.challenge([m.g2_hex(k), m.g1_hex(phi), m.g1_hex(E1), m.g1_hex(E2),
                             m.g2_hex(Vk), m.g1_hex(Vphi), m.g1_hex(VE1), m.g1_hex(VE2)], _b(ad).decode("latin1"))
        else:
            c = m.challenge([m.g2_hex(k), m.g1_hex(phi), m.g2_hex(Vk), m.g1_hex(Vphi)], _b(ad).decode("latin1"))
        rs = [(rhos[i] - hs[i] * c) % m.r for i in range(len(hs))]
        rs.append((rho_t - t * c) % m.r)
        if with_retrieval:
            rs.append((rho_e - eps * c) % m.r)
        pr = IdProof(sig1, sig2, k, phi, c, rs, [b"" if hide else _b(a) for a, hide in attrs])
        if with_retrieval:
            pr.E1, pr.E2, pr.has_E = E1, E2, True
        return pr


def scalar_stream(seed: int, i: int, r: int) -> int:
    """Deterministic synthetic scalar S(seed, i) = LE-int(SHA-256(le64(seed) || le64(i))) mod r."""
    d = hashlib.sha256(seed.to_bytes(8, "little") + i.to_bytes(8, "little")).digest()
    return int.from_bytes(d, "little") % r
