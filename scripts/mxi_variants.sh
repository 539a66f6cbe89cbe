#!/bin/bash
# Build variants of k_fft2d_inv_mx as stand-alone code objects for scripts/ubench_mxi.cpp (round 6):
#   scripts/mxi_variants.sh name "[-D... / -mllvm ...]" [sed-script applied to the generated assembly]
# Output: scripts/session/mxi/name.hsaco (+ name.s)
set -e
HERE=$(cd "$(dirname "$0")" && pwd); OUT=$HERE/session/mxi; L=/opt/rocm/lib/llvm/bin
mkdir -p "$OUT"
name=$1; flags=$2; edit=$3
hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S $flags -x hip "$HERE/mxi_kern.hip" -o "$OUT/$name.s" 2>/dev/null
if [ -n "$edit" ]; then python3 "$HERE/mxi_edit_isa.py" "$OUT/$name.s" $edit; fi
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$OUT/$name.s" -o "$OUT/$name.o"
$L/ld.lld -shared "$OUT/$name.o" -o "$OUT/$name.hsaco"
rm -f "$OUT/$name.o"
echo "built $OUT/$name.hsaco"
