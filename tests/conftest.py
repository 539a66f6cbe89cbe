import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names(prefix=""):
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR)
                  if f.endswith(".npz") and f.startswith(prefix))


def load_golden(name):
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


DENSE_GOLDEN = [n for n in golden_names() if n.startswith(("d1_", "d2_", "d3_", "darcy_"))]
FACT_GOLDEN = [n for n in golden_names() if n.startswith(("tucker_", "cp_"))]


@pytest.fixture
def golden():
    return load_golden
