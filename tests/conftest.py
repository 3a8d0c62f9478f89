import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "qwen3-asr-swift_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


import pytest  # noqa: E402


@pytest.fixture(scope="session")
def sd_small_stress():
    """The full-size Qwen3-ASR-0.6B state dict with seeded 'stress' weights, generated ONCE per test session (780 M random values: several
    seconds) and shared by the full-geometry GPU test modules; tests must not modify it."""
    from oracle import config as C
    from qasr import synth
    return synth.synth_state_dict(C.AUDIO_SMALL, C.TEXT_SMALL, seed=0, init="stress")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
