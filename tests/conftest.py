import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the package has no fallback path: make sure the HIP library exists (hipcc cross-compiles without a GPU)
    so = os.path.join(ROOT, "scmomer_amd", "libscmomer_hip.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "scmomer_amd", "csrc"), "-j8"], check=True)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
