import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _ensure_built():
    """The test session needs the in-tree binaries (product .so files, the lab build the forced-variant tests use, and
    the oracle).  They are git-ignored, so on a fresh checkout build them once here (hipcc cross-compiles gfx950 without
    a GPU); an existing build is left alone."""
    need = [os.path.join(ROOT, "qutlass_amd", "libqutlass_amd.so"), os.path.join(ROOT, "qutlass", "_CUDA.abi3.so"),
            os.path.join(ROOT, "qutlass_amd", "libqutlass_amd_bench.so"), os.path.join(ROOT, "oracle", "libqutlass_oracle.so")]
    if all(os.path.exists(f) for f in need):
        return
    import importlib.util

    spec = importlib.util.spec_from_file_location("_qutlass_amd_build", os.path.join(ROOT, "qutlass_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build(verbose=True)
    b.build_bench_lib(verbose=True)
    import oracle

    oracle.build()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
