"""The C-ABI library loads (no GPU needed to dlopen it) and exports every function include/elpasso.h declares; the Python
binding declares a signature for each of them; without a GPU every entry fails loudly (no CPU fallback)."""
import ctypes
import importlib
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared():
    txt = open(os.path.join(ROOT, "include", "elpasso.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(elp_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported_and_bound():
    elp = importlib.import_module("ps-signature-and-el-passo_amd")
    b = importlib.import_module("ps-signature-and-el-passo_amd.build")
    b.build_hip()
    lib = elp.load_library()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "library does not export %s" % n
        assert n in elp.elpasso.EXPORTED_SYMBOLS, "python binding lacks %s" % n
    assert lib.elp_field_bytes(0) == 32 and lib.elp_field_bytes(1) == 48
    assert lib.elp_verify_id_record_size(0, 8, 4, 1) == 800          # 804 B with the 4-byte verdict
    assert lib.elp_verify_id_record_size(1, 8, 4, 1) == 1024         # BLS12-381: 1028 B
    assert lib.elp_ps_verify_record_size(0, 3) == 224 and lib.elp_provide_id_record_size(0, 8, 4) == 416


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    elp = importlib.import_module("ps-signature-and-el-passo_amd")
    with pytest.raises(elp.ElpassoError):
        elp.Context()
    h = ctypes.c_void_p()
    assert elp.load_library().elp_init(0, 0, ctypes.byref(h)) == -4   # ELP_ERR_NODEVICE


def test_host_layer_builds():
    b = importlib.import_module("ps-signature-and-el-passo_amd.build")
    lib = ctypes.CDLL(b.build_host())
    for n in ("elph_init", "elph_verify_id_b64", "elph_prove_id_b64", "elph_request_id_b64", "elph_ps_verify_b64", "elph_user_name_b64"):
        assert hasattr(lib, n)


def test_host_layer_sha256_one_shot_both_code_paths():
    """csrc/host: Fr::setHashOf goes through a one-shot SHA-256 that uses the x86 SHA extensions where the CPU has them; both code paths against hashlib, every
    length across the one- / two-block padding boundary and a few multi-block ones."""
    import ctypes
    import hashlib
    import importlib
    import os
    b = importlib.import_module("ps-signature-and-el-passo_amd.build")
    if not os.path.exists(b.HOST_LIB):
        import pytest
        pytest.skip("host library not built yet")
    L = ctypes.CDLL(b.HOST_LIB)
    L.elph_sha256.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_int]
    out = ctypes.create_string_buffer(32)
    for n in list(range(0, 200)) + [255, 256, 1000, 4097]:
        msg = bytes((7 * i + n) & 0xFF for i in range(n))
        for force in (1, 0):
            L.elph_sha256(msg, n, out, force)
            assert out.raw == hashlib.sha256(msg).digest(), (n, force)
    assert L.elph_sha256_has_hardware() in (0, 1)
