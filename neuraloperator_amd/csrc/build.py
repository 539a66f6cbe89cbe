"""Build libsc_engine.so for gfx950 with hipcc (in-tree, next to the package)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libsc_engine.so")
SOURCES = ["sc_engine.cpp"]
# every kernel header next to the source (globbed: a new sc_kernels_*.h can not be forgotten here)
HEADERS = sorted(f for f in os.listdir(HERE) if f.endswith(".h")) + [os.path.join("..", "..", "include", "sc_engine.h")]


def find_hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.isfile(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or put /opt/rocm/bin on PATH)")


def is_stale():
    if not os.path.isfile(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, out=None, defines=()):
    """out/defines: measurement variants (scripts/) -- the product library takes neither."""
    if out is None and not force and not is_stale():
        if verbose:
            print(f"[build] {OUT} up to date")
        return OUT
    out = out or OUT
    cmd = [find_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + \
          [f"-D{d}" for d in defines] + ["-x", "hip"] + [os.path.join(HERE, s) for s in SOURCES] + ["-o", out]
    if verbose:
        print("[build]", " ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
