"""CPU tier (needs the built library): the product library reads only its documented environment switches.

VERDICT r5 item 7: 34 `getenv` A-B / diagnostic switches lived in libsc_engine.so, one of them on a launch path and one
routing production plans to a known-bad kernel.  They are compiled in only with -DSC_DIAG now (scripts/build_diag.py, the
emulation tier); this test is `strings libsc_engine.so | grep '^SC_'` against the list INTEGRATION.md documents."""
import os
import re

PUBLIC = {"SC_NO_SIDE_STREAM", "SC_PLAN_NO_MX_FFT", "SC_TKC"}


def test_only_documented_switch_names_in_the_product_library():
    from neuraloperator_amd.csrc import build as b
    so = b.build(verbose=False)
    data = open(so, "rb").read()
    names = {m.group(0).decode() for m in re.finditer(rb"(?<![\x20-\x7e])SC_[A-Z0-9_]{3,}(?![\x20-\x7e])", data)}
    assert names <= PUBLIC, "undocumented environment switch names in the product library: " + ", ".join(sorted(names - PUBLIC))
    assert PUBLIC <= names, "a documented switch is no longer read: " + ", ".join(sorted(PUBLIC - names))
    doc = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    for n in PUBLIC:
        assert n in doc, n + " is not documented in INTEGRATION.md"


def test_no_getenv_in_a_kernel_header():
    """launch helpers live in the kernel headers: nothing there may read the environment (the side-stream switch of
    sc_device.h is read once per process)"""
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neuraloperator_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.startswith("sc_kernels_") and f.endswith(".h"):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f
