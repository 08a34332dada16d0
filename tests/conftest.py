"""pytest configuration: markers, import path, and on-demand build of the CPU checkers."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the oracle's bit-exactness contract is the single-thread one (SURVEY.md finding 2/3)
os.environ.setdefault("OMP_NUM_THREADS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The C restatement is test infrastructure: build it if the .so is missing (gcc only, seconds)."""
    odir = os.path.join(ROOT, "oracle")
    if not (os.path.exists(os.path.join(odir, "liboracle_f32.so")) and os.path.exists(os.path.join(odir, "liboracle_f64.so"))):
        subprocess.check_call(["make", "-C", odir, "oracle"], stdout=subprocess.DEVNULL)
    yield


GOLDEN = os.path.join(ROOT, "tests", "golden")
