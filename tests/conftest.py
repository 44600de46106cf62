import os
import sys

import pytest

# the tests flip experiment knobs of the library (iteration-path switches, kernel variants): csrc/mln_options.h
os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
