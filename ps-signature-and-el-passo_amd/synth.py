"""Synthetic workload generator for bench.py and the full-size GPU tests.

Deterministic inputs: scalar stream S(seed, i) = LE(SHA-256(le64(seed) || le64(i))) mod r, seed 20211; generators
g = hashAndMapToG1("abc"), gg = a fixed order-r G2 point; RP parameters as test/ps-tests.cc:106-121 of the reference
(authority_pk = H1("ghi"), h = H1("jkl"), g = H1("abc"), service "service", associated data "hello"); item n has the
attributes "a{i}-{n}", the first H hidden; every 97th item (n % 97 == 13) is corrupted by flipping bit 0 of c.

Because the generator plays IdP and user at once it knows every discrete logarithm, so each group element of a proof is
ONE fixed-base multi-scalar multiplication on the GPU (elp_g1_msm_fixed / elp_g2_msm_fixed); the proofs are distributed
exactly like honestly generated ones.  Host-side work is only Fr arithmetic on Python integers and SHA-256.
Nothing here imports the oracle.
"""
import hashlib

import numpy as np

R_BN254 = 0x2523648240000001BA344D8000000007FF9F800000000010A10000000000000D
# order-r point on the BN254 twist (affine x.a | x.b | y.a | y.b, little-endian), used as the G2 generator gg
GG_BN254 = bytes.fromhex(
    "53e90ba632bda83099d860ce92347de44dca398c07f2271a5a8f3f54897a390fb1011c784fa0b538b8eb217d224924c74f2ffbc74eb8c23285de12b"
    "fde717623b861e3414fe19b5bb6e3bc8b95ca355bbe1b14e987ce5a06d1aaaa621de185218785e57ab4a43cd84d4db2517e9f21fffb93978e27f729"
    "bd866ad84d3fd1320a")


R_BLS12_381 = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
# standard BLS12-381 G2 generator (affine x.a | x.b | y.a | y.b, 48-byte little-endian coordinates)
GG_BLS12_381 = b"".join(int(v, 16).to_bytes(48, "little") for v in (
    "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8",
    "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e",
    "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801",
    "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be"))
CURVES = {0: (R_BN254, GG_BN254), 1: (R_BLS12_381, GG_BLS12_381)}


def scalar_stream(seed, i, r=R_BN254):
    d = hashlib.sha256(int(seed).to_bytes(8, "little") + int(i).to_bytes(8, "little")).digest()
    return int.from_bytes(d, "little") % r


def fr_set_hash_of(msg, r=R_BN254):
    """mcl Fr::setHashOf (src/ps-verifier.cc:25,122,224): LE(SHA-256) masked to bitlen(r), top bit cleared if >= r."""
    h = int.from_bytes(hashlib.sha256(msg).digest(), "little")
    nb = r.bit_length()
    h &= (1 << nb) - 1
    if h >= r:
        h &= (1 << (nb - 1)) - 1
    return h


def _fr_bytes(vals):
    return b"".join(int(v).to_bytes(32, "little") for v in vals)


def g1_wire(aff, F=32):
    """affine std (n x 2F) -> mcl wire form (n x F): x with the parity of y in the top bit; zeros stay zeros."""
    a = np.frombuffer(aff, dtype=np.uint8).reshape(-1, 2 * F)
    out = a[:, :F].copy()
    out[:, F - 1] |= (a[:, F] & 1) << 7
    return out


def g2_wire(aff, F=32):
    a = np.frombuffer(aff, dtype=np.uint8).reshape(-1, 4 * F)
    out = a[:, :2 * F].copy()
    out[:, 2 * F - 1] |= (a[:, 2 * F] & 1) << 7
    return out


class Workload:
    """Key material + RP parameters installed into an elpasso Context, and batch generators on top of it."""

    def __init__(self, ctx, nattr, seed=20211, window_bits=0, service=b"service", ad=b"hello"):
        self.ctx, self.A, self.seed = ctx, nattr, seed
        self.r, gg_const = CURVES[getattr(ctx, "curve", 0)]
        self.F = ctx.F
        self.service, self.ad = service, ad
        r = self.r
        self.x = scalar_stream(seed, 0, r)
        self.ys = [scalar_stream(seed, 1 + i, r) for i in range(nattr)]
        pts = ctx.hash_to_g1([b"abc", b"ghi", b"jkl"])
        G1 = ctx.G1
        self.g, self.apk, self.h = pts[:G1], pts[G1:2 * G1], pts[2 * G1:3 * G1]
        self.gg = gg_const
        # key_gen (src/ps-signer.cc:29-55): X = g^x, XX = gg^x, Y_i = g^y_i, YY_i = gg^y_i
        ks = _fr_bytes([self.x] + self.ys)
        g1s = ctx.g1_mul(self.g * (nattr + 1), ks)
        g2s = ctx.g2_mul(self.gg * (nattr + 1), ks)
        self.X, self.Yi = g1s[:G1], g1s[G1:]
        self.XX, self.YYi = g2s[:ctx.G2], g2s[ctx.G2:]
        import time
        t0 = time.perf_counter()
        ctx.set_pubkey(self.g, self.gg, self.XX, self.Yi, self.YYi, window_bits)
        t1 = time.perf_counter()
        ctx.set_rp(service, self.apk, self.g, self.h)
        ctx.set_signer_secret(self.X)
        t2 = time.perf_counter()
        # key set-up as a relying party pays it: all tables once (set_pubkey), then the G1 tables again for the RP parameters / signer secret
        self.t_set_pubkey_ms, self.t_set_params_ms = (t1 - t0) * 1e3, (t2 - t1) * 1e3
        self._ctr = 1000

    def _fresh(self, count):
        base = self._ctr
        self._ctr += count
        return [scalar_stream(self.seed, base + j, self.r) for j in range(count)]

    def attributes(self, n):
        return [b"a%d-%d" % (i, n) for i in range(self.A)]

    # ---------------------------------------------------------------------------------------------------------
    def verify_id_batch(self, n_items, nhidden, first_item=0, with_retrieval=True, corrupt_every=97, corrupt_at=13,
                        degenerate_items=(), window_bits=8):
        """Returns (records bytes, hidden_mask, expected flags np.uint8[n]).

        Items listed in `degenerate_items` are VALID proofs whose blinding t is chosen so that k equals the first table
        entry added while accumulating K = k * prod YY_i^{m_i} (k == d0 * YY_H with d0 the lowest window digit of m_H):
        the verifier's group law meets the exceptional case P + P and must still accept."""
        ctx, A, H, r = self.ctx, self.A, nhidden, self.r
        x, ys = self.x, self.ys
        N = n_items
        e_sig, e_k, e_phi, e_E1, e_E2, e_Vk, e_Vphi, e_VE1, e_VE2 = ([] for _ in range(9))
        keep = []
        for n in range(first_item, first_item + N):
            m = [fr_set_hash_of(a, r) for a in self.attributes(n)]
            nrnd = 4 + H + 2
            u, t, rr, eps, *rest = self._fresh(nrnd)
            if n in degenerate_items and H < A:
                d0 = (m[H] % r) & ((1 << window_bits) - 1)
                if d0 > (1 << (window_bits - 1)):
                    d0 -= 1 << window_bits             # signed digits (csrc/elp/curve.h fixed_base_digit): the entry added first is the NEGATED |d0| YY_H
                if d0:
                    t = (d0 * ys[H] - x - sum(ys[j] * m[j] for j in range(H))) % r
            rho = rest[:H]
            rho_t, rho_e = rest[H], rest[H + 1]
            full = (x + sum(y * mi for y, mi in zip(ys, m))) % r
            # randomised signature: sig1' = g^(u r), sig2' = g^(r u (x + sum y_i m_i + t))   (src/ps-requester.cc:163-170)
            e_sig += [u * rr % r, rr * u % r * ((full + t) % r) % r]
            hid = (x + sum(ys[j] * m[j] for j in range(H)) + t) % r
            e_k.append(hid)                                         # k = XX prod YY_j^m_j gg^t        (:189-204)
            e_phi.append(m[0])                                      # phi = H1(svc)^s                  (:182-187)
            e_E1.append(eps)                                        # E1 = g^eps                       (:172-180)
            e_E2 += [eps, m[1] if A > 1 else 0]                     # E2 = y^eps h^gamma
            e_Vk.append((x + sum(ys[j] * rho[j] for j in range(H)) + rho_t) % r)      # (:227-246)
            e_Vphi.append(rho[0])
            e_VE1.append(rho_e)
            e_VE2 += [rho_e, rho[1] if H > 1 else 0]
            keep.append((m, t, eps, rho, rho_t, rho_e))
        G1, G2 = ctx.G1, ctx.G2
        hs, geg, apk, hh = A + 1, A + 2, A + 3, A + 4
        sig = ctx.g1_msm_fixed([0], _fr_bytes(e_sig))               # 2N points: sig1', sig2'
        kk = ctx.g2_msm_fixed([0], _fr_bytes(e_k))
        phi = ctx.g1_msm_fixed([hs], _fr_bytes(e_phi))
        Vk = ctx.g2_msm_fixed([0], _fr_bytes(e_Vk))
        Vphi = ctx.g1_msm_fixed([hs], _fr_bytes(e_Vphi))
        F = self.F
        parts = [g2_wire(kk, F), g1_wire(phi, F)]
        if with_retrieval:
            E1 = ctx.g1_msm_fixed([geg], _fr_bytes(e_E1))
            E2 = ctx.g1_msm_fixed([apk, hh], _fr_bytes(e_E2))
            VE1 = ctx.g1_msm_fixed([geg], _fr_bytes(e_VE1))
            VE2 = ctx.g1_msm_fixed([apk, hh], _fr_bytes(e_VE2))
            parts += [g1_wire(E1, F), g1_wire(E2, F)]
        parts += [g2_wire(Vk, F), g1_wire(Vphi, F)]
        if with_retrieval:
            parts += [g1_wire(VE1, F), g1_wire(VE2, F)]
        recs = bytearray()
        expect = np.ones(N, dtype=np.uint8)
        for i in range(N):
            m, t, eps, rho, rho_t, rho_e = keep[i]
            tr = b"".join(bytes(p[i]).hex().encode() for p in parts) + self.ad
            c = fr_set_hash_of(hashlib.sha256(tr).digest(), r)      # SHA-256 twice (src/ps-requester.cc:263-274)
            rs = [(rho[j] - m[j] * c) % r for j in range(H)] + [(rho_t - t * c) % r]
            if with_retrieval:
                rs.append((rho_e - eps * c) % r)
            if corrupt_every and (first_item + i) % corrupt_every == corrupt_at:
                c ^= 1
                expect[i] = 0
            recs += sig[2 * i * G1:(2 * i + 2) * G1] + phi[i * G1:(i + 1) * G1]
            if with_retrieval:
                recs += E1[i * G1:(i + 1) * G1] + E2[i * G1:(i + 1) * G1]
            recs += kk[i * G2:(i + 1) * G2] + _fr_bytes([c] + rs + m[H:])
        return bytes(recs), (1 << H) - 1, expect

    # ---------------------------------------------------------------------------------------------------------
    def wire_messages(self, recs, n_items, nhidden, first_item=0, with_retrieval=True):
        """The same proofs as IdProof::toBufferString() messages (src/ps-encoding.cc:451-467: T-L-V, compressed points, attribute
        strings with "" for hidden ones).  Returns (bytes, uint32 offsets[n+1]) for elp_verify_id_wire_batch."""
        A, H, F = self.A, nhidden, self.F
        G1, G2 = 2 * F, 4 * F
        rsz = len(recs) // n_items
        a = np.frombuffer(recs, dtype=np.uint8).reshape(n_items, rsz)
        o = 0
        sig1 = g1_wire(a[:, o:o + G1].tobytes(), F); o += G1
        sig2 = g1_wire(a[:, o:o + G1].tobytes(), F); o += G1
        phi = g1_wire(a[:, o:o + G1].tobytes(), F); o += G1
        if with_retrieval:
            E1 = g1_wire(a[:, o:o + G1].tobytes(), F); o += G1
            E2 = g1_wire(a[:, o:o + G1].tobytes(), F); o += G1
        kk = g2_wire(a[:, o:o + G2].tobytes(), F); o += G2
        c = a[:, o:o + 32]; o += 32
        nrs = H + (2 if with_retrieval else 1)
        rs = a[:, o:o + 32 * nrs]
        hdr1, hdr2 = bytes([1, F]), bytes([2, 2 * F])
        out, off = bytearray(), np.zeros(n_items + 1, dtype=np.uint32)
        for i in range(n_items):
            out += hdr1 + sig1[i].tobytes() + hdr1 + sig2[i].tobytes() + hdr2 + kk[i].tobytes() + hdr1 + phi[i].tobytes()
            out += bytes([3, 32]) + c[i].tobytes() + bytes([6, nrs])
            r = rs[i].tobytes()
            for j in range(nrs):
                out += bytes([32]) + r[32 * j:32 * j + 32]
            out += bytes([7, A])
            for j, s in enumerate(self.attributes(first_item + i)):
                if j < H:
                    out.append(0)
                else:
                    out.append(len(s))
                    out += s
            if with_retrieval:
                out += hdr1 + E1[i].tobytes() + hdr1 + E2[i].tobytes()
            off[i + 1] = len(out)
        return bytes(out), off

    # ---------------------------------------------------------------------------------------------------------
    def prove_id_batch(self, n_items, nhidden, first_item=0, with_retrieval=True):
        """Input records of elp_prove_id_batch: valid credentials (g^u, g^(u (x + sum y_i m_i))) + the prover's randomness
        in the reference's draw order (src/ps-requester.cc:150-310).  Returns (records, hidden_mask)."""
        ctx, A, H, r = self.ctx, self.A, nhidden, self.r
        e, tails = [], []
        for n in range(first_item, first_item + n_items):
            m = [fr_set_hash_of(a, r) for a in self.attributes(n)]
            u, *rnd = self._fresh(1 + 2 + (1 if with_retrieval else 0) + H + 1 + (1 if with_retrieval else 0))
            full = (self.x + sum(y * mi for y, mi in zip(self.ys, m))) % r
            e += [u, u * full % r]
            tails.append(_fr_bytes(m + rnd))
        sig = ctx.g1_msm_fixed([0], _fr_bytes(e))
        G1 = ctx.G1
        return b"".join(sig[2 * i * G1:(2 * i + 2) * G1] + tails[i] for i in range(n_items)), (1 << H) - 1

    def distinct_proofs_dev(self, n_items, nhidden, users, dev, stream, corrupt_every=97, corrupt_at=13, seed=7):
        """n_items DISTINCT el_passo_verify_id records on the device, synthesised by the batch prover (elp_prove_id_batch_dev): `users` credentials (the
        deterministic prove_id_batch inputs) presented n_items / users times each with fresh prover randomness per presentation -- every record differs in
        sig1', sig2', k, E1, E2, c and the responses (a user's pseudonym phi and revealed attributes repeat, as they do when a user signs on again).  Every
        corrupt_every-th item gets bit 0 of c flipped.  Returns (torch uint8 tensor of records, hidden_mask, expected flags np.uint8[n_items]).
        Builds 2^20 proofs in seconds where the host-side generator of verify_id_batch needs a minute."""
        import torch
        ctx, A, H = self.ctx, self.A, nhidden
        assert n_items % users == 0
        base, mask = self.prove_id_batch(users, H, with_retrieval=True)
        psz = len(base) // users
        nrand = 2 + 1 + H + 1 + 1                                   # t, r, eps, rho[H], rho_t, rho_e
        a = np.frombuffer(base, dtype=np.uint8).reshape(users, psz)
        recs = np.tile(a, (n_items // users, 1))
        rng = np.random.default_rng(seed)
        fresh = rng.integers(0, 256, size=(n_items - users, nrand * 32), dtype=np.uint8)
        fresh[:, 31::32] &= 0x1F                                    # below 2^253 < r on both curves
        recs[users:, psz - nrand * 32:] = fresh
        d_in = torch.from_numpy(recs.reshape(-1)).to(dev)
        osz = ctx.lib.elp_verify_id_record_size(ctx.curve, A, H, 1)
        d_out = torch.zeros(n_items * osz, dtype=torch.uint8, device=dev)
        d_fl = torch.zeros(n_items, dtype=torch.uint8, device=dev)
        d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        d_ad = torch.from_numpy(np.frombuffer(self.ad, dtype=np.uint8).copy()).to(dev)
        ctx._chk(ctx.lib.elp_prove_id_batch_dev(ctx.h, stream, n_items, d_in.data_ptr(), mask, 1, d_ad.data_ptr(), None, len(self.ad), d_out.data_ptr(),
                                                d_fl.data_ptr(), d_cnt.data_ptr()))
        torch.cuda.synchronize()
        assert int(d_fl.sum().item()) == n_items, "the batch prover rejected an input"
        expect = np.ones(n_items, dtype=np.uint8)
        if corrupt_every:
            bad = np.arange(corrupt_at, n_items, corrupt_every)
            expect[bad] = 0
            off_c = 5 * ctx.G1 + ctx.G2                             # sig1 | sig2 | phi | E1 | E2 | k | c ...
            idx = torch.from_numpy(bad.astype(np.int64) * osz + off_c).to(dev)
            d_out[idx] ^= 1
        return d_out, mask, expect

    def request_id_batch(self, n_items, nhidden, first_item=0):
        """Input records of elp_request_id_batch: m[A] | t | rho_0 | rho[H].  Returns (records, hidden_mask)."""
        r = self.r
        out = []
        for n in range(first_item, first_item + n_items):
            m = [fr_set_hash_of(a, r) for a in self.attributes(n)]
            out.append(_fr_bytes(m + self._fresh(2 + nhidden)))
        return b"".join(out), (1 << nhidden) - 1

    # ---------------------------------------------------------------------------------------------------------
    def ps_verify_batch(self, n_items, first_item=0, corrupt_every=97, corrupt_at=13):
        """PS signatures on all-plaintext attributes: sigma = (g^u, g^(u (x + sum y_i m_i)))."""
        ctx, r = self.ctx, self.r
        e, ms = [], []
        expect = np.ones(n_items, dtype=np.uint8)
        for n in range(first_item, first_item + n_items):
            m = [fr_set_hash_of(a, r) for a in self.attributes(n)]
            (u,) = self._fresh(1)
            full = (self.x + sum(y * mi for y, mi in zip(self.ys, m))) % r
            if corrupt_every and n % corrupt_every == corrupt_at:
                full = (full + 1) % r
                expect[n - first_item] = 0
            e += [u, u * full % r]
            ms.append(m)
        sig = ctx.g1_msm_fixed([0], _fr_bytes(e))
        G1 = ctx.G1
        recs = b"".join(sig[2 * i * G1:(2 * i + 2) * G1] + _fr_bytes(ms[i]) for i in range(n_items))
        return recs, expect

    # ---------------------------------------------------------------------------------------------------------
    def provide_id_batch(self, n_items, nhidden, first_item=0, corrupt_every=97, corrupt_at=13):
        """Credential requests (src/ps-requester.cc:19-99) + signing nonces.  Returns (records, mask, expect)."""
        ctx, A, H, r = self.ctx, self.A, nhidden, self.r
        eA, eV, keep = [], [], []
        for n in range(first_item, first_item + n_items):
            m = [fr_set_hash_of(a, r) for a in self.attributes(n)]
            t1, rho0, u, *rho = self._fresh(3 + H)
            eA.append((t1 + sum(self.ys[j] * m[j] for j in range(H))) % r)       # A = g^t prod Y_j^m_j
            eV.append((rho0 + sum(self.ys[j] * rho[j] for j in range(H))) % r)   # V = g^rho0 prod Y_j^rho_j
            keep.append((m, t1, rho0, rho, u))
        Apts = ctx.g1_msm_fixed([0], _fr_bytes(eA))
        Vpts = ctx.g1_msm_fixed([0], _fr_bytes(eV))
        wa, wv = g1_wire(Apts, self.F), g1_wire(Vpts, self.F)
        G1 = ctx.G1
        recs = bytearray()
        expect = np.ones(n_items, dtype=np.uint8)
        for i in range(n_items):
            m, t1, rho0, rho, u = keep[i]
            c = fr_set_hash_of(hashlib.sha256(bytes(wa[i]).hex().encode() + bytes(wv[i]).hex().encode() + self.ad).digest(), r)
            rs = [(rho0 - t1 * c) % r] + [(rho[j] - m[j] * c) % r for j in range(H)]
            if corrupt_every and (first_item + i) % corrupt_every == corrupt_at:
                c ^= 1
                expect[i] = 0
            recs += Apts[i * G1:(i + 1) * G1] + _fr_bytes([c] + rs + m[H:] + [u])
        return bytes(recs), (1 << H) - 1, expect
