"""CPU tier (needs the built library and llvm-objdump, no GPU): no kernel that executes v_mfma_f32_16x16x32_bf16 may hold
a packed-fp32 instruction with operand selects op_sel:[0,1,.].

Round 6 (sc_kernels_fft3mx.h F3_NOTE_PK_MUL_LX): on MI355X `v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32` with
op_sel:[0,1,.] -- the low lane takes the LOW half of src0 and the HIGH half of src1 -- return 0 in the low result of lanes
48-63 while another wave of the same SIMD executes v_mfma_f32_16x16x32_bf16 (scripts/ubench_pk_forms.hip: 18 % of such
executions; no other matrix instruction of the library -- 16x16x32_f16 and, rarely, i32_16x16x64_i8 do the same and are
watched too -- and no other op_sel combination).  That was round 5's
"non-repeatable k_fft2d_inv_mx<64>": hipcc had emitted the group-twiddle product of its column task in that encoding.
The kernels that run this matrix instruction are persistent with two workgroups per compute unit, i.e. the other wave of
the SIMD is the SAME kernel: their own code must be free of the encoding.  (Nothing else of the engine runs beside them:
the side stream only carries two-pass transform chunks and Tucker-chain launches, sc_engine.cpp.)

The test disassembles the code object hipcc embedded in libsc_engine.so and reads the instructions as shipped."""
import os
import re
import shutil
import subprocess

import pytest

from test_isa_scratch import gfx950_code_object

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
# the matrix instructions that trigger it (scripts/ubench_pk_forms.hip): the three 16 x 16 shapes gfx950 added -- the engine uses
# the bf16 one only; 32x32x16_bf16, 16x16x16_bf16 / _f16 and the fp32 shapes do not trigger it
TRIGGER = re.compile(r"v_mfma_(?:f32_16x16x32_bf16|f32_16x16x32_f16|i32_16x16x64_i8)")
PK = re.compile(r"^\s*(v_pk_(?:mul|fma|add)_f32)\s+(.*)$")


def op_sel_low(rest):
    """(op_sel of src0, op_sel of src1): which half the LOW lane reads; absent = 0"""
    m = re.search(r"op_sel:\[([0-9,]+)\]", rest)
    sel = [int(v) for v in m.group(1).split(",")] if m else [0, 0]
    return sel[0], sel[1]


def vulnerable(line):
    m = PK.match(line)
    return bool(m) and op_sel_low(m.group(2)) == (0, 1)


def test_parser_knows_the_failing_encoding():
    assert vulnerable("\tv_pk_mul_f32 v[22:23], v[4:5], v[6:7] op_sel:[0,1] op_sel_hi:[0,0]")        # round 5's instruction
    assert vulnerable("  v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[0,1,0] op_sel_hi:[1,1,1] // 0000: D3B0")
    assert vulnerable("v_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1]")
    assert not vulnerable("\tv_pk_mul_f32 v[22:23], v[6:7], v[4:5] op_sel:[1,0] op_sel_hi:[0,0]")   # the commuted product: clean
    assert not vulnerable("\tv_pk_fma_f32 v[6:7], v[6:7], v[2:3], v[22:23] op_sel:[0,0,1] op_sel_hi:[1,1,0]")
    assert not vulnerable("\tv_pk_mul_f32 v[4:5], v[2:3], v[36:37] op_sel_hi:[0,1]")
    assert not vulnerable("\tv_pk_fma_f32 v[14:15], v[14:15], v[20:21], v[22:23] op_sel:[1,1,0] op_sel_hi:[0,1,1]")


@pytest.mark.skipif(not os.path.isfile(OBJDUMP), reason="llvm-objdump not found")
def test_kernels_beside_the_bf16_matrix_instruction_hold_no_op_sel_01_packed_fp32(tmp_path):
    from neuraloperator_amd.csrc import build as b
    so = b.build(verbose=False)
    co = os.path.join(str(tmp_path), "sc_engine_gfx950.co")
    open(co, "wb").write(gfx950_code_object(so))
    asm = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout
    kernels, cur = {}, None
    for line in asm.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            kernels[cur] = {"trigger": 0, "bad": [], "pk": 0}
            continue
        if cur is None:
            continue
        if TRIGGER.search(line):
            kernels[cur]["trigger"] += 1
        if PK.match(line):
            kernels[cur]["pk"] += 1
            if vulnerable(line):
                kernels[cur]["bad"].append(line.split("//")[0].strip())
    assert len(kernels) > 400, f"only {len(kernels)} kernels found: the disassembly parser no longer matches"
    assert sum(k["pk"] for k in kernels.values()) > 30000, "packed-fp32 instructions not recognised"
    beside = {n: k for n, k in kernels.items() if k["trigger"]}
    # the two bfloat16 transforms, three heights each (H = 64 of the inverse-type kernel since round 6); the forward-type one
    # with two and with three bf16 terms per twiddle (SC_PLAN_MX_FFT_3TERM)
    assert sum("k_fft2d_inv_mx" in n for n in beside) == 3 and sum("k_fft2d_fwd_mx" in n for n in beside) == 6, sorted(beside)
    assert all("k_fft2d_inv_mx" in n or "k_fft2d_fwd_mx" in n for n in beside), \
        "a new kernel executes one of the triggering matrix instructions: " + ", ".join(sorted(beside))
    bad = {n: k["bad"] for n, k in beside.items() if k["bad"]}
    assert not bad, "packed-fp32 with op_sel:[0,1,.] beside v_mfma_f32_16x16x32_bf16 (F3_NOTE_PK_MUL_LX):\n" + "\n".join(
        f"  {n}: {len(v)} x e.g. {v[0]}" for n, v in bad.items())
