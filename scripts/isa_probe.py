"""Compile explicit instantiations of engine kernels to gfx950 ISA and print their resource use (registers, spills,
scratch, LDS, occupancy) -- seconds per kernel instead of the four minutes of the whole library.

    python scripts/isa_probe.py sc_kernels_fft2p.h 'template __global__ void k_f2p_col_inv_w1024<true>(const cf32*, cf32*, const cf32*, int, int, int, int, int);' [-DX ...] [--keep out.s]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neuraloperator_amd", "csrc")


def probe(header, insts, defines=(), keep=None):
    src = '#include "%s"\n%s\nint main() { return 0; }\n' % (os.path.join(CSRC, header), "\n".join(insts))
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "p.hip")
        open(f, "w").write(src)
        out = keep or os.path.join(d, "p.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only",
                               "-S", "-o", out, f] + list(defines))
        asm = open(out).read()
    res = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S):
        body = m.group(2)
        g = lambda k: int(re.search(r"\.amdhsa_" + k + r"\s+(\d+)", body).group(1))
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        meta = re.search(re.escape(m.group(1)) + r"\n.*?\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", asm, re.S)
        agpr = re.search(r"; AccumOffset: (\d+)", asm)
        res.append(dict(name=name, scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"),
                        vgpr_next=g("next_free_vgpr"), sgpr=g("next_free_sgpr")))
    # spills / occupancy from the comment block hipcc writes behind each function
    for r, m in zip(res, re.finditer(r"; NumVgprs: (\d+)\n; NumAgprs: (\d+)\n; TotalNumVgprs: (\d+)\n; ScratchSize: (\d+)\n; "
                                     r"MemoryBound: \d+\n; FloatMode: \d+\n; IeeeMode: \d+\n; LDSByteSize: (\d+) bytes/workgroup.*?\n"
                                     r"; SGPRBlocks: \d+\n; VGPRBlocks: \d+\n; NumSGPRsForWavesPerEU: \d+\n; "
                                     r"NumVGPRsForWavesPerEU: \d+\n; AccumOffset: \d+\n; Occupancy: (\d+)", asm)):
        r.update(vgpr=int(m.group(1)), agpr=int(m.group(2)), occupancy=int(m.group(6)))
    return res, asm


if __name__ == "__main__":
    args = sys.argv[1:]
    keep = None
    if "--keep" in args:
        i = args.index("--keep")
        keep = args[i + 1]
        del args[i:i + 2]
    defs = [a for a in args if a.startswith("-D")]
    rest = [a for a in args if not a.startswith("-D")]
    res, asm = probe(rest[0], rest[1:], defs, keep)
    for r in res:
        spills = len(re.findall(r"scratch_store", asm)) if len(res) == 1 else "?"
        print(f"scratch {r['scratch']:4d}  vgpr {r.get('vgpr', '?'):>3} agpr {r.get('agpr', '?'):>3} occ {r.get('occupancy', '?')}  "
              f"lds {r['lds']:6d}  scratch_stores {spills}  {r['name'][:110]}")
