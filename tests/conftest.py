import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(tryfirst=True)            # conftest hooks register after xdist's own tryfirst hook, so this one runs first
def pytest_cmdline_main(config):
    """The CPU tier (`-m "not gpu"`) is dominated by the kernels in host emulation (one OS thread per lane): ~20 min in
    one process, ~7 min over four pytest-xdist workers on the 8 cores of the build container.  When that tier is selected
    and the caller did not choose a worker count, spread it over min(4, cores / 2) workers (xdist's own hook, which runs
    after this one, turns `numprocesses` into the worker set-up; the emulation library build is under a file lock:
    engine_runner.emu_lib).  SC_TEST_WORKERS=0 keeps everything in one process; the GPU tier is never spread (one
    device)."""
    if getattr(config, "workerinput", None) is not None:
        return None
    opt = config.option
    if not hasattr(opt, "numprocesses") or opt.numprocesses is not None:
        return None
    if "not gpu" not in (getattr(opt, "markexpr", "") or ""):
        return None
    if getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False):
        return None
    try:
        want = int(os.environ.get("SC_TEST_WORKERS", "-1"))
    except ValueError:
        want = -1
    if want < 0:
        want = min(4, (os.cpu_count() or 1) // 2)
    if want >= 2:
        opt.numprocesses = want
    return None


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
