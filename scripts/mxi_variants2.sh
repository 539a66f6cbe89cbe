#!/bin/bash
# as mxi_variants.sh, the assembly edited by ${EDITOR_PY:-mxi_edit2.py} MODE
set -e
HERE=$(cd "$(dirname "$0")" && pwd); OUT=$HERE/session/mxi; L=/opt/rocm/lib/llvm/bin
name=$1; mode=$2
cp "$OUT/base.s" "$OUT/$name.s"; python3 "$HERE/${EDITOR_PY:-mxi_edit2.py}" "$OUT/$name.s" $mode
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$OUT/$name.s" -o "$OUT/$name.o"
$L/ld.lld -shared "$OUT/$name.o" -o "$OUT/$name.hsaco"; rm -f "$OUT/$name.o"
