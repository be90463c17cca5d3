import os
import sys

import pytest

os.environ.setdefault("FFNO_ALLOW_TEST_BACKEND", "1")   # the emulator hook of fourierflow_amd._lib is inert without it

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "emu: runs the HIP kernel sources through the CPU wave emulator")
