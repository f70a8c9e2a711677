import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def elp():
    """The product package (directory name contains hyphens, so it is imported by string)."""
    import importlib
    return importlib.import_module("ps-signature-and-el-passo_amd")


@pytest.fixture(scope="session")
def gpu_ctx(elp):
    ctx = elp.Context(elp.CURVE_BN254, 0)   # raises without a GPU or without the built .so: no fallback
    # parity tests compare with the reference's verdicts bit for bit, including its acceptance of sig1 = sig2 = infinity (golden
    # case "sig_both_zero"); the library's default rejects that forgery (ELP_OPT_STRICT_SIGNATURE, tested in test_gpu_round2.py::test_strict_signature_default_and_state_checks and test_gpu_host_layer.py::test_default_strict_signature_through_reference_api)
    ctx.set_strict_signature(False)
    yield ctx
    ctx.close()
