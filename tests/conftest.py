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
    yield ctx
    ctx.close()
