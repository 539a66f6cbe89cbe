#!/bin/bash
# scripts/pmlp_variants.sh name "[-DPK_NW=8 -DPK_LIN=true ...]"  ->  scripts/session/plinx/name.hsaco + name.sym (mangled kernel name)
set -e
HERE=$(cd "$(dirname "$0")" && pwd); OUT=$HERE/session/plinx; L=/opt/rocm/lib/llvm/bin
mkdir -p "$OUT"; name=$1; flags=$2
hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S $flags -x hip "$HERE/plinx_kern.hip" -o "$OUT/$name.s" 2>/dev/null
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$OUT/$name.s" -o "$OUT/$name.o"
$L/ld.lld -shared "$OUT/$name.o" -o "$OUT/$name.hsaco"; rm -f "$OUT/$name.o"
$L/llvm-readelf -s "$OUT/$name.hsaco" | grep "FUNC.*k_plinx_bwd" | awk '{print $8}' > "$OUT/$name.sym"
python3 - "$OUT/$name.s" "$name" <<PY
import re,sys
s=open(sys.argv[1]).read()
for m in re.finditer(r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.vgpr_count:\s*(\d+)\s*\n\s*\.vgpr_spill_count:\s*(\d+)", s, re.S):
    if "plinx_bwd" in m.group(2): print(sys.argv[2], m.group(2)[:40], "lds", m.group(1), "scratch", m.group(3), "vgpr", m.group(4), "spill", m.group(5))
PY
