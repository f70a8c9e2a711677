"""ctypes binding of include/elpasso.h (the HIP hot path).  There is NO CPU fallback: if the shared library is
missing, or no GPU is present, every entry point raises."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ELP_LIB") or os.path.join(HERE, "csrc", "libelpasso_hip.so")   # ELP_LIB: A/B builds of the same HIP library

CURVE_BN254 = 0
CURVE_BLS12_381 = 1
OPT_STRICT_SIGNATURE = 1
OPT_PAIRED_LAYOUT = 2
OPT_TABLE_WORKSPACE = 3
OPT_SPLIT_PHASES = 4
OPT_SUBGROUP_CHECK = 5
OPT_COOP_PAIRING = 6
OPT_COALESCED_RECORDS = 7
OPT_STREAM_OVERLAP = 8
OPT_FAULT_INJECT = 9
OPT_PAIR4 = 10
OPT_WIRE_DECODE = 11
OPT_AGG_TWO_PER_LANE = 12
OPT_PAIR16 = 13

_c = ctypes
_u8p = _c.c_void_p
_SIGS = {
    "elp_init": (_c.c_int, [_c.c_int, _c.c_int, _c.POINTER(_c.c_void_p)]),
    "elp_device_count": (_c.c_int, []),
    "elp_verify_id_batch_submit": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_size_t, _c.c_void_p, _c.c_uint64, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "elp_verify_id_batch_wait": (_c.c_int, [_c.c_void_p, _c.c_int, _c.POINTER(_c.c_uint64)]),
    "elp_verify_id_batch_stage": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_size_t, _c.c_size_t, _c.c_size_t, _c.c_size_t, _c.c_void_p]),
    "elp_destroy": (None, [_c.c_void_p]),
    "elp_last_error": (_c.c_char_p, [_c.c_void_p]),
    "elp_field_bytes": (_c.c_int, [_c.c_int]),
    "elp_set_option": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int]),
    "elp_key_table_bytes": (_c.c_size_t, [_c.c_void_p]),
    "elp_host_alloc": (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.POINTER(_c.c_void_p)]),
    "elp_host_free": (None, [_c.c_void_p, _c.c_void_p]),
    "elp_version": (_c.c_char_p, []),
    "elp_set_pubkey": (_c.c_int, [_c.c_void_p, _c.c_int, _u8p, _u8p, _u8p, _u8p, _u8p, _c.c_int]),
    "elp_set_rp": (_c.c_int, [_c.c_void_p, _u8p, _c.c_size_t, _u8p, _u8p, _u8p]),
    "elp_set_signer_secret": (_c.c_int, [_c.c_void_p, _u8p]),
    "elp_g1_decompress": (_c.c_int, [_c.c_void_p, _c.c_size_t, _u8p, _u8p, _u8p]),
    "elp_g2_decompress": (_c.c_int, [_c.c_void_p, _c.c_size_t, _u8p, _u8p, _u8p]),
    "elp_g1_mul": (_c.c_int, [_c.c_void_p, _c.c_size_t, _u8p, _u8p, _u8p]),
    "elp_g2_mul": (_c.c_int, [_c.c_void_p, _c.c_size_t, _u8p, _u8p, _u8p]),
    "elp_g1_add": (_c.c_int, [_c.c_void_p, _c.c_size_t, _u8p, _u8p, _u8p]),
    "elp_g2_add": (_c.c_int, [_c.c_void_p, _c.c_size_t, _u8p, _u8p, _u8p]),
    "elp_g1_msm_fixed": (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.c_int, _u8p, _u8p, _u8p]),
    "elp_g2_msm_fixed": (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.c_int, _u8p, _u8p, _u8p]),
    "elp_g1_msm": (_c.c_int, [_c.c_void_p, _c.c_size_t, _u8p, _u8p, _u8p]),
    "elp_g2_msm": (_c.c_int, [_c.c_void_p, _c.c_size_t, _u8p, _u8p, _u8p]),
    "elp_msm_workspace_bytes": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_size_t]),
    "elp_g1_msm_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "elp_g2_msm_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "elp_hash_to_g1": (_c.c_int, [_c.c_void_p, _c.c_size_t, _u8p, _u8p, _u8p]),
    "elp_pairing": (_c.c_int, [_c.c_void_p, _c.c_size_t, _u8p, _u8p, _u8p]),
    "elp_pairing_check": (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.c_int, _u8p, _u8p, _u8p]),
    "elp_verify_id_record_size": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "elp_ps_verify_record_size": (_c.c_size_t, [_c.c_int, _c.c_int]),
    "elp_provide_id_record_size": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int]),
    "elp_verify_id_batch": (_c.c_int, [_c.c_void_p, _c.c_size_t, _u8p, _c.c_uint64, _c.c_int, _u8p, _u8p, _c.c_size_t, _u8p,
                                       _c.POINTER(_c.c_uint64)]),
    "elp_verify_id_wire_batch": (_c.c_int, [_c.c_void_p, _c.c_size_t, _u8p, _u8p, _c.c_int, _u8p, _u8p, _c.c_size_t, _u8p,
                                            _c.POINTER(_c.c_uint64)]),
    "elp_verify_id_wire_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p,
                                                _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p]),
    "elp_verify_id_batch_aggregated": (_c.c_int, [_c.c_void_p, _c.c_size_t, _u8p, _c.c_uint64, _c.c_int, _u8p, _u8p, _c.c_size_t, _u8p, _u8p,
                                                  _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_int)]),
    "elp_verify_id_batch_aggregated_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_uint64, _c.c_int, _c.c_void_p,
                                                      _c.c_void_p, _c.c_size_t, _u8p, _c.c_void_p, _c.c_void_p]),
    "elp_ps_verify_batch": (_c.c_int, [_c.c_void_p, _c.c_size_t, _u8p, _c.c_int, _u8p, _c.POINTER(_c.c_uint64)]),
    "elp_provide_id_batch": (_c.c_int, [_c.c_void_p, _c.c_size_t, _u8p, _c.c_uint64, _u8p, _u8p, _c.c_size_t, _u8p, _u8p,
                                        _c.POINTER(_c.c_uint64)]),
    "elp_request_id_record_size": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int]),
    "elp_request_id_out_size": (_c.c_size_t, [_c.c_int, _c.c_int]),
    "elp_prove_id_record_size": (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "elp_request_id_batch": (_c.c_int, [_c.c_void_p, _c.c_size_t, _u8p, _c.c_uint64, _u8p, _u8p, _c.c_size_t, _u8p]),
    "elp_prove_id_batch": (_c.c_int, [_c.c_void_p, _c.c_size_t, _u8p, _c.c_uint64, _c.c_int, _u8p, _u8p, _c.c_size_t, _u8p, _u8p,
                                      _c.POINTER(_c.c_uint64)]),
    "elp_request_id_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_uint64, _c.c_void_p, _c.c_void_p,
                                            _c.c_size_t, _c.c_void_p]),
    "elp_prove_id_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_uint64, _c.c_int, _c.c_void_p,
                                          _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "elp_verify_id_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_uint64, _c.c_int, _c.c_void_p,
                                           _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p]),
    "elp_ps_verify_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p]),
    "elp_provide_id_batch_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_uint64, _c.c_void_p,
                                            _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "elp_time_verify_id_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_size_t, _c.c_void_p, _c.c_uint64, _c.c_int,
                                          _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p,
                                          _c.POINTER(_c.c_float)]),
    "elp_bench_fp_mul": (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.c_int, _c.POINTER(_c.c_float)]),
    "elp_bench_op": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_size_t, _c.c_int, _c.POINTER(_c.c_float)]),
}
EXPORTED_SYMBOLS = sorted(_SIGS)

_lib = None


class ElpassoError(RuntimeError):
    pass


def load_library():
    """Load libelpasso_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ElpassoError("HIP library %s is missing: run `python -m __graft_entry__` / build.py first; "
                               "there is no CPU fallback" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _buf(x):
    """bytes / bytearray / numpy array -> (keepalive, void*)"""
    if x is None:
        return None, None
    if isinstance(x, np.ndarray):
        a = np.ascontiguousarray(x)
        return a, a.ctypes.data
    a = np.frombuffer(bytes(x), dtype=np.uint8)
    return a, a.ctypes.data


class Context:
    """One elp_ctx (one GPU, one stream)."""

    def __init__(self, curve=CURVE_BN254, device=0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.elp_init(curve, device, ctypes.byref(h))
        if rc != 0:
            raise ElpassoError("elp_init failed (%d)%s" % (rc, ": no GPU, and there is no CPU fallback" if rc == -4 else ""))
        self.h = h
        self.curve = curve
        self.F = self.lib.elp_field_bytes(curve)
        self.G1, self.G2, self.GT = 2 * self.F, 4 * self.F, 12 * self.F
        self.A = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.elp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise ElpassoError("elpasso error %d: %s" % (rc, self.lib.elp_last_error(self.h).decode()))

    # ---- setup
    def set_strict_signature(self, on):
        """ELP_OPT_STRICT_SIGNATURE (default on): reject sig1 == infinity in verify_id; off = the reference's behaviour."""
        self._chk(self.lib.elp_set_option(self.h, OPT_STRICT_SIGNATURE, int(bool(on))))

    def set_paired_layout(self, on):
        """ELP_OPT_PAIRED_LAYOUT: False / 0 = one lane per item, True / 1 = two lanes per item, 2 = chosen by batch size (default)."""
        self._chk(self.lib.elp_set_option(self.h, OPT_PAIRED_LAYOUT, int(on)))

    def set_table_workspace(self, on):
        """ELP_OPT_TABLE_WORKSPACE: per-item tables of the variable-base multiplications in a launch workspace (default) or in private memory."""
        self._chk(self.lib.elp_set_option(self.h, OPT_TABLE_WORKSPACE, int(bool(on))))

    def key_table_bytes(self):
        """Device bytes of the installed key's fixed-base tables."""
        return int(self.lib.elp_key_table_bytes(self.h))

    def set_subgroup_check(self, on):
        """ELP_OPT_SUBGROUP_CHECK (default on; BLS12-381): prover-supplied G1 points outside the order-r subgroup reject the item."""
        self._chk(self.lib.elp_set_option(self.h, OPT_SUBGROUP_CHECK, int(bool(on))))

    def set_coop_pairing(self, on):
        """ELP_OPT_COOP_PAIRING: 0 = off, 1 = cooperative pairing check for batches of <= 4096 items and the aggregated tail (default), > 1 = that batch limit."""
        self._chk(self.lib.elp_set_option(self.h, OPT_COOP_PAIRING, int(on)))

    def set_coalesced_records(self, on):
        """ELP_OPT_COALESCED_RECORDS: verify_id records fetched per workgroup with coalesced 16-byte loads through LDS into a private copy (k_verify_id_staged)."""
        self._chk(self.lib.elp_set_option(self.h, OPT_COALESCED_RECORDS, int(bool(on))))

    def set_stream_overlap(self, on):
        """ELP_OPT_STREAM_OVERLAP (default off): independent kernels of one call on the context's second stream (small-batch verify_id, aggregated tail)."""
        self._chk(self.lib.elp_set_option(self.h, OPT_STREAM_OVERLAP, int(bool(on))))

    def set_pair4(self, mode):
        """ELP_OPT_PAIR4: 0 = off, 1 = the four-lanes-per-item pairing check for mid-size batches (default), 2 = wherever the path exists."""
        self._chk(self.lib.elp_set_option(self.h, OPT_PAIR4, int(mode)))

    def set_wire_decode(self, on):
        """ELP_OPT_WIRE_DECODE: wire batches of up to 16 384 messages through a decode kernel + the record paths (default) or always through the fused wire kernels."""
        self._chk(self.lib.elp_set_option(self.h, OPT_WIRE_DECODE, int(bool(on))))

    def set_pair16(self, value):
        """ELP_OPT_PAIR16: small PS-verification batches on one 16-lane row per item (0 off, 1 on up to 4 096 items, > 1: up to that many)."""
        self._chk(self.lib.elp_set_option(self.h, OPT_PAIR16, int(value)))

    def set_agg_two_per_lane(self, mode):
        """ELP_OPT_AGG_TWO_PER_LANE: aggregated verification with two proofs per lane (0 never -- the default --, 1 where it saves rounds of lanes, 2 always)."""
        self._chk(self.lib.elp_set_option(self.h, OPT_AGG_TWO_PER_LANE, int(mode)))

    def set_split_phases(self, on):
        """ELP_OPT_SPLIT_PHASES (default 0): 1 / 2 = one-lane-per-item verify_id as phase-split kernels (NIZK jobs, then the pairing); identical results."""
        self._chk(self.lib.elp_set_option(self.h, OPT_SPLIT_PHASES, int(on)))

    def set_pubkey(self, g, gg, XX, Yi, YYi, window_bits=0):
        A = len(Yi) // self.G1
        assert len(YYi) == A * self.G2
        ka = [_buf(x) for x in (g, gg, XX, Yi, YYi)]
        self._chk(self.lib.elp_set_pubkey(self.h, A, ka[0][1], ka[1][1], ka[2][1], ka[3][1], ka[4][1], window_bits))
        self.A = A

    def set_rp(self, service_name, authority_pk=None, g=None, h=None):
        sn = bytes(service_name) if service_name is not None else None
        ka = [_buf(x) for x in (sn, authority_pk, g, h)]
        self._chk(self.lib.elp_set_rp(self.h, ka[0][1], len(sn) if sn is not None else 0, ka[1][1], ka[2][1], ka[3][1]))

    def set_signer_secret(self, X):
        k = _buf(X)
        self._chk(self.lib.elp_set_signer_secret(self.h, k[1]))

    # ---- primitives (bytes in, bytes out)
    def _dec(self, fn, n, wire, osz):
        out = np.zeros(n * osz, dtype=np.uint8)
        ok = np.zeros(n, dtype=np.uint8)
        k = _buf(wire)
        self._chk(fn(self.h, n, k[1], out.ctypes.data, ok.ctypes.data))
        return out.tobytes(), ok

    def g1_decompress(self, wire):
        return self._dec(self.lib.elp_g1_decompress, len(wire) // self.F, wire, self.G1)

    def g2_decompress(self, wire):
        return self._dec(self.lib.elp_g2_decompress, len(wire) // (2 * self.F), wire, self.G2)

    def _bin(self, fn, n, a, b, osz):
        out = np.zeros(n * osz, dtype=np.uint8)
        ka, kb = _buf(a), _buf(b)
        self._chk(fn(self.h, n, ka[1], kb[1], out.ctypes.data))
        return out.tobytes()

    def g1_mul(self, pts, ks):
        return self._bin(self.lib.elp_g1_mul, len(pts) // self.G1, pts, ks, self.G1)

    def g2_mul(self, pts, ks):
        return self._bin(self.lib.elp_g2_mul, len(pts) // self.G2, pts, ks, self.G2)

    def g1_add(self, a, b):
        return self._bin(self.lib.elp_g1_add, len(a) // self.G1, a, b, self.G1)

    def g2_add(self, a, b):
        return self._bin(self.lib.elp_g2_add, len(a) // self.G2, a, b, self.G2)

    def _msm(self, fn, base_ids, scalars, osz):
        ids = np.asarray(base_ids, dtype=np.int32)
        nt = len(ids)
        n = len(scalars) // (32 * nt)
        out = np.zeros(n * osz, dtype=np.uint8)
        ks = _buf(scalars)
        self._chk(fn(self.h, n, nt, ids.ctypes.data, ks[1], out.ctypes.data))
        return out.tobytes()

    def g1_msm_fixed(self, base_ids, scalars):
        return self._msm(self.lib.elp_g1_msm_fixed, base_ids, scalars, self.G1)

    def g2_msm_fixed(self, base_ids, scalars):
        return self._msm(self.lib.elp_g2_msm_fixed, base_ids, scalars, self.G2)

    def g1_msm(self, pts, ks):
        """sum_i ks[i] * pts[i] (single output, Pippenger)."""
        out = np.zeros(self.G1, dtype=np.uint8)
        ka, kb = _buf(pts), _buf(ks)
        self._chk(self.lib.elp_g1_msm(self.h, len(pts) // self.G1, ka[1], kb[1], out.ctypes.data))
        return out.tobytes()

    def g2_msm(self, pts, ks):
        out = np.zeros(self.G2, dtype=np.uint8)
        ka, kb = _buf(pts), _buf(ks)
        self._chk(self.lib.elp_g2_msm(self.h, len(pts) // self.G2, ka[1], kb[1], out.ctypes.data))
        return out.tobytes()

    def hash_to_g1(self, msgs):
        off = np.zeros(len(msgs) + 1, dtype=np.uint32)
        for i, m in enumerate(msgs):
            off[i + 1] = off[i] + len(m)
        data = _buf(b"".join(bytes(m) for m in msgs) or b"\0")
        out = np.zeros(len(msgs) * self.G1, dtype=np.uint8)
        self._chk(self.lib.elp_hash_to_g1(self.h, len(msgs), data[1], off.ctypes.data, out.ctypes.data))
        return out.tobytes()

    def pairing(self, g1, g2):
        return self._bin(self.lib.elp_pairing, len(g1) // self.G1, g1, g2, self.GT)

    def pairing_check(self, npairs, g1, g2):
        n = len(g1) // (self.G1 * npairs)
        ok = np.zeros(n, dtype=np.uint8)
        ka, kb = _buf(g1), _buf(g2)
        self._chk(self.lib.elp_pairing_check(self.h, n, npairs, ka[1], kb[1], ok.ctypes.data))
        return ok

    # ---- fused batches over host buffers
    @staticmethod
    def _ad(ad):
        """ad: bytes (shared) or list of bytes (per item) -> (data, offsets or None, ad_len)."""
        if isinstance(ad, (bytes, bytearray)):
            return bytes(ad), None, len(ad)
        off = np.zeros(len(ad) + 1, dtype=np.uint32)
        for i, a in enumerate(ad):
            off[i + 1] = off[i] + len(a)
        return b"".join(bytes(a) for a in ad), off, 0

    def verify_id_batch(self, records, hidden_mask, with_retrieval, ad):
        H = bin(hidden_mask).count("1")
        rsz = self.lib.elp_verify_id_record_size(self.curve, self.A, H, int(with_retrieval))
        n = len(records) // rsz
        assert n * rsz == len(records), "record size mismatch"
        data, off, adl = self._ad(ad)
        flags = np.zeros(n, dtype=np.uint8)
        cnt = ctypes.c_uint64(0)
        kr, kd = _buf(records), _buf(data or b"\0")
        self._chk(self.lib.elp_verify_id_batch(self.h, n, kr[1], hidden_mask, int(with_retrieval), kd[1],
                                               off.ctypes.data if off is not None else None, adl, flags.ctypes.data,
                                               ctypes.byref(cnt)))
        return flags, cnt.value

    def verify_id_batch_aggregated(self, records, hidden_mask, with_retrieval, ad, seed=None):
        """Returns (flags, accepted, batch_equation_held).  seed=None: the library draws it from the OS CSPRNG (recommended)."""
        if seed is not None and len(seed) != 32:
            raise ValueError("seed must be exactly 32 bytes")
        H = bin(hidden_mask).count("1")
        rsz = self.lib.elp_verify_id_record_size(self.curve, self.A, H, int(with_retrieval))
        n = len(records) // rsz
        data, off, adl = self._ad(ad)
        flags = np.zeros(n, dtype=np.uint8)
        cnt, held = ctypes.c_uint64(0), ctypes.c_int(0)
        kr, kd, ks = _buf(records), _buf(data or b"\0"), _buf(seed)
        self._chk(self.lib.elp_verify_id_batch_aggregated(self.h, n, kr[1], hidden_mask, int(with_retrieval), kd[1],
                                                          off.ctypes.data if off is not None else None, adl, ks[1], flags.ctypes.data,
                                                          ctypes.byref(cnt), ctypes.byref(held)))
        return flags, cnt.value, bool(held.value)

    def verify_id_wire_batch(self, messages, with_retrieval, ad):
        """messages: list of raw IdProof wire messages (bytes, base64 already decoded)."""
        n = len(messages)
        moff = np.zeros(n + 1, dtype=np.uint32)
        for i, m in enumerate(messages):
            moff[i + 1] = moff[i] + len(m)
        data, off, adl = self._ad(ad)
        flags = np.zeros(n, dtype=np.uint8)
        cnt = ctypes.c_uint64(0)
        km, kd = _buf(b"".join(bytes(m) for m in messages) or b"\0"), _buf(data or b"\0")
        self._chk(self.lib.elp_verify_id_wire_batch(self.h, n, km[1], moff.ctypes.data, int(with_retrieval), kd[1],
                                                    off.ctypes.data if off is not None else None, adl, flags.ctypes.data,
                                                    ctypes.byref(cnt)))
        return flags, cnt.value

    def ps_verify_batch(self, records, nattr):
        rsz = self.lib.elp_ps_verify_record_size(self.curve, nattr)
        n = len(records) // rsz
        flags = np.zeros(n, dtype=np.uint8)
        cnt = ctypes.c_uint64(0)
        kr = _buf(records)
        self._chk(self.lib.elp_ps_verify_batch(self.h, n, kr[1], nattr, flags.ctypes.data, ctypes.byref(cnt)))
        return flags, cnt.value

    def request_id_batch(self, records, hidden_mask, ad):
        """PSRequester::el_passo_request_id in batch; records: m[A] | t | rho_0 | rho[H] -> A | c | rs[H+1] per item."""
        H = bin(hidden_mask).count("1")
        rsz = self.lib.elp_request_id_record_size(self.curve, self.A, H)
        osz = self.lib.elp_request_id_out_size(self.curve, H)
        n = len(records) // rsz
        assert n * rsz == len(records), "record size mismatch"
        data, off, adl = self._ad(ad)
        out = np.zeros(n * osz, dtype=np.uint8)
        kr, kd = _buf(records), _buf(data or b"\0")
        self._chk(self.lib.elp_request_id_batch(self.h, n, kr[1], hidden_mask, kd[1], off.ctypes.data if off is not None else None,
                                                adl, out.ctypes.data))
        return out.tobytes()

    def prove_id_batch(self, records, hidden_mask, with_retrieval, ad):
        """PSRequester::el_passo_prove_id in batch; returns (verify_id records, flags, produced)."""
        H = bin(hidden_mask).count("1")
        rsz = self.lib.elp_prove_id_record_size(self.curve, self.A, H, int(with_retrieval))
        osz = self.lib.elp_verify_id_record_size(self.curve, self.A, H, int(with_retrieval))
        n = len(records) // rsz
        assert n * rsz == len(records), "record size mismatch"
        data, off, adl = self._ad(ad)
        flags = np.zeros(n, dtype=np.uint8)
        out = np.zeros(n * osz, dtype=np.uint8)
        cnt = ctypes.c_uint64(0)
        kr, kd = _buf(records), _buf(data or b"\0")
        self._chk(self.lib.elp_prove_id_batch(self.h, n, kr[1], hidden_mask, int(with_retrieval), kd[1],
                                              off.ctypes.data if off is not None else None, adl, out.ctypes.data, flags.ctypes.data,
                                              ctypes.byref(cnt)))
        return out.tobytes(), flags, cnt.value

    def provide_id_batch(self, records, hidden_mask, ad):
        H = bin(hidden_mask).count("1")
        rsz = self.lib.elp_provide_id_record_size(self.curve, self.A, H)
        n = len(records) // rsz
        assert n * rsz == len(records), "record size mismatch"
        data, off, adl = self._ad(ad)
        flags = np.zeros(n, dtype=np.uint8)
        sigs = np.zeros(n * 2 * self.G1, dtype=np.uint8)
        cnt = ctypes.c_uint64(0)
        kr, kd = _buf(records), _buf(data or b"\0")
        self._chk(self.lib.elp_provide_id_batch(self.h, n, kr[1], hidden_mask, kd[1], off.ctypes.data if off is not None else None,
                                                adl, sigs.ctypes.data, flags.ctypes.data, ctypes.byref(cnt)))
        return sigs.tobytes(), flags, cnt.value
