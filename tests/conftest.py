import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _f32_compute_after_each_test():
    """--mixed-precision is a process-wide switch (like the Keras policy): a CLI test that turns it on must not
    leak bf16 operands into the parity tests that follow."""
    yield
    mod = sys.modules.get("speech_recognition_amd.ops")
    if mod is not None and hasattr(mod, "set_mixed_precision"):
        mod.set_mixed_precision(False)
