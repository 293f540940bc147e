import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a test that stops making progress must fail with a traceback, not hang the suite: every wait on the device is bounded
    # (the one-launch sweeps give up after ~0.3 s), so ten minutes of silence is a defect worth a stack dump
    if config.pluginmanager.hasplugin("timeout") and not config.getoption("timeout", None):
        config.option.timeout = 900
    if getattr(config.option, "faulthandler_timeout", None) in (None, 0.0):
        try:
            config._inicache["faulthandler_timeout"] = 600.0
        except Exception:
            pass


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
