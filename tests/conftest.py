import os
import sys

import pytest

# the tests drive alternative launch plans / kernels through the library's development switches (CDC_PF, CDC_NO_*, ...),
# which a process honours only when it was started with CDC_DEV=1 (cdc_internal.h: dev_env)
os.environ.setdefault("CDC_DEV", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
