"""Derivation of the GLV (G1) / GLS (G2) scalar-decomposition constants emitted into csrc/elp/params_<curve>.h.
Build tooling (uses the big-int model only to derive and self-check numbers)."""
from fractions import Fraction


def sqrt_mod(a, p):
    a %= p
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    q, s = p - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, t, r = s, pow(z, q, p), pow(a, q, p), pow(a, (q + 1) // 2, p)
    while t != 1:
        i, tt = 0, t
        while tt != 1:
            tt = tt * tt % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c = i, b * b % p
        t, r = t * c % p, r * b % p
    return r


def lll(B):
    """Textbook LLL (delta = 3/4) over exact rationals; B = list of integer rows."""
    B = [list(r) for r in B]
    n = len(B)

    def gs():
        Bs, mu = [], [[Fraction(0)] * n for _ in range(n)]
        for i in range(n):
            v = [Fraction(x) for x in B[i]]
            for j in range(i):
                d = sum(a * b for a, b in zip(Bs[j], Bs[j]))
                mu[i][j] = sum(Fraction(a) * b for a, b in zip(B[i], Bs[j])) / d
                v = [a - mu[i][j] * b for a, b in zip(v, Bs[j])]
            Bs.append(v)
        return Bs, mu

    k = 1
    while k < n:
        Bs, mu = gs()
        for j in range(k - 1, -1, -1):
            q = round(mu[k][j])
            if q:
                B[k] = [a - q * b for a, b in zip(B[k], B[j])]
                Bs, mu = gs()
        if sum(a * a for a in Bs[k]) >= (Fraction(3, 4) - mu[k][k - 1] ** 2) * sum(a * a for a in Bs[k - 1]):
            k += 1
        else:
            B[k], B[k - 1] = B[k - 1], B[k]
            k = max(k - 1, 1)
    return B


def inv_first_row(M):
    n = len(M)
    A = [[Fraction(x) for x in r] + [Fraction(int(i == j)) for j in range(n)] for i, r in enumerate(M)]
    for i in range(n):
        p = next(r for r in range(i, n) if A[r][i] != 0)
        A[i], A[p] = A[p], A[i]
        f = A[i][i]
        A[i] = [x / f for x in A[i]]
        for r in range(n):
            if r != i and A[r][i] != 0:
                g = A[r][i]
                A[r] = [a - g * b for a, b in zip(A[r], A[i])]
    return [A[0][n + j] for j in range(n)]


class Decomp:
    """k -> (k_0..k_{d-1}) with sum k_i lam^i == k (mod r), by Babai rounding against a reduced basis."""

    def __init__(self, r, lam, d):
        self.r, self.lam, self.d = r, lam, d
        rows = [[r] + [0] * (d - 1)] + [[(-pow(lam, i, r)) % r] + [int(j == i - 1) for j in range(d - 1)] for i in range(1, d)]
        self.B = lll(rows)
        for row in self.B:
            assert sum(c * pow(lam, i, r) for i, c in enumerate(row)) % r == 0
        inv = inv_first_row(self.B)            # (k,0,..,0) * B^-1 = k * inv
        self.G = [int(round(x * (1 << 256))) for x in inv]      # signed

    def split(self, k):
        """Mirrors the device arithmetic: c_j = (k*|G_j| + 2^255) >> 256 with the sign of G_j."""
        c = []
        for g in self.G:
            v = (k * abs(g) + (1 << 255)) >> 256
            c.append(-v if g < 0 else v)
        out = []
        for i in range(self.d):
            out.append((k if i == 0 else 0) - sum(c[j] * self.B[j][i] for j in range(self.d)))
        assert sum(v * pow(self.lam, i, self.r) for i, v in enumerate(out)) % self.r == k % self.r
        return out


def derive(cv):
    """Returns dict with beta, lam1, Decomp for G1 (d=2) and lam2, Decomp for G2 (d=4), self-checked on the model."""
    import random
    from oracle.pymodel import Groups, Mcl
    G = Groups(cv)
    p, r = cv.p, cv.r
    m = Mcl(cv)
    P = m.hash_to_g1(b"glv-check")
    s3 = sqrt_mod(p - 3, p)
    beta = (-1 + s3) * pow(2, -1, p) % p
    assert pow(beta, 3, p) == 1 and beta != 1
    t3 = sqrt_mod(r - 3, r)
    lam1 = (-1 + t3) * pow(2, -1, r) % r
    if G.g1_mul(P, lam1) != (beta * P[0] % p, P[1]):
        lam1 = (-1 - t3) * pow(2, -1, r) % r
    assert G.g1_mul(P, lam1) == (beta * P[0] % p, P[1])
    d1 = Decomp(r, lam1, 2)
    lam2 = p % r
    d2 = Decomp(r, lam2, 4)
    rnd = random.Random(5)
    mx1 = mx2 = 0
    for k in [0, 1, 2, r - 1, r - 2, (r - 1) // 2, lam1, lam2, r - lam2] + [rnd.randrange(r) for _ in range(3000)]:
        mx1 = max(mx1, max(abs(v) for v in d1.split(k)))
        mx2 = max(mx2, max(abs(v) for v in d2.split(k)))
    assert mx1.bit_length() <= 130 and mx2.bit_length() <= 67, (mx1.bit_length(), mx2.bit_length())
    # worst case of the round-off (Babai) step: |k_i| <= (1/2 + eps) * sum_j |B_ji| (+ one basis vector for the truncated c_j); the
    # signed 4-bit recoding on the device (limbs_add_eights) needs k_i + 0x88..8 to fit 33 / 17 nibbles
    for dec, nib in ((d1, 33), (d2, 17)):
        for i in range(dec.d):
            worst = sum(abs(row[i]) for row in dec.B) * 3 // 2 + 1
            assert worst + (8 * (16 ** nib - 1)) // 15 < 16 ** nib, (nib, worst.bit_length())
    return {"beta": beta, "lam1": lam1, "d1": d1, "lam2": lam2, "d2": d2, "bits1": mx1.bit_length(), "bits2": mx2.bit_length()}


if __name__ == "__main__":
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle.pymodel import BLS12_381, BN254, Groups
    for cv in (BN254, BLS12_381):
        d = derive(cv)
        print(cv.name, "G1 sub-scalar bits", d["bits1"], "G2 sub-scalar bits", d["bits2"])
        print("  G1 basis", d["d1"].B, [g.bit_length() for g in d["d1"].G])
        print("  G2 basis", d["d2"].B, [g.bit_length() for g in d["d2"].G])
