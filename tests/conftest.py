import os
import sys

import pytest

# the tests flip experiment knobs of the library (iteration-path switches, kernel variants): csrc/mln_options.h
# (MELLON_AMD_TEST_SHIPPED_DEFAULTS=1: leave it unset -- tests/test_gpu_defaults.py re-runs the golden tests that way)
if os.environ.get("MELLON_AMD_TEST_SHIPPED_DEFAULTS") != "1":
    os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
