"""CPU tier (needs the built library and llvm-readelf, no GPU): NO kernel the library ships may use scratch memory.

A kernel with a non-zero private segment spills registers or keeps a local array in memory -- on gfx950 that is HBM
traffic and latency inside loops that are budgeted to the byte (VERDICT r4 weak 7 listed twelve such instantiations, among
them kernels on BASELINE configs[4]'s path).  The test reads the code object hipcc embedded in libsc_engine.so (the
clang offload bundle in .hip_fatbin), parses the AMDGPU metadata note of every kernel and fails on any
.private_segment_fixed_size or .vgpr_spill_count above zero (spilled SGPRs live in lanes of a VGPR, not in memory: they
are reported, not failed on)."""
import os
import re
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = shutil.which("llvm-readelf") or "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def gfx950_code_object(so_path):
    data = open(so_path, "rb").read()
    at = data.find(MAGIC)
    assert at >= 0, "no offload bundle in " + so_path
    n, = struct.unpack_from("<Q", data, at + len(MAGIC))
    off = at + len(MAGIC) + 8
    for _ in range(n):
        o, size, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        triple = data[off:off + tl].decode()
        off += tl
        if "gfx950" in triple:
            return data[at + o:at + o + size]
    raise AssertionError("no gfx950 entry in the bundle")


def kernel_resources(code_object, tmp_path):
    f = os.path.join(str(tmp_path), "sc_engine_gfx950.co")
    open(f, "wb").write(code_object)
    notes = subprocess.run([READELF, "--notes", f], capture_output=True, text=True, check=True).stdout
    out = {}
    for blk in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out[name] = {k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
                     for k in ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "vgpr_count")}
    return out


@pytest.mark.skipif(not os.path.isfile(READELF), reason="llvm-readelf not found")
def test_no_shipped_kernel_uses_scratch(tmp_path):
    from neuraloperator_amd.csrc import build as b
    so = b.build(verbose=False)                         # up to date after __graft_entry__.build(); rebuilt if stale
    res = kernel_resources(gfx950_code_object(so), tmp_path)
    assert len(res) > 400, f"only {len(res)} kernels found: the note parser no longer matches"
    bad = {n: r for n, r in res.items() if r["private_segment_fixed_size"] or r["vgpr_spill_count"]}
    if bad and shutil.which("c++filt"):
        bad = {subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()[:140]: r for n, r in bad.items()}
    assert not bad, "kernels with scratch / spills:\n" + "\n".join(f"  {n}: {r}" for n, r in bad.items())
