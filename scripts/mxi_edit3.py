"""Data-flow edits of the first twiddle product of the column task in k_fft2d_inv_mx<64> (round 6):
  mxi_edit3.py file.s MODE
  mul_scalar : each v_pk_mul_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[0,0]  ->  v_mul_f32 D.lo, A.lo, B.hi ; v_mul_f32 D.hi, A.lo, B.lo
  mul_twice  : the same instruction issued twice
  fma_scalar : the v_pk_fma_f32 that consumes its result (plain form, no op_sel) -> two v_fma_f32
  mul_nop    : s_nop 15 behind each of those v_pk_mul_f32"""
import re, sys
path, mode = sys.argv[1], sys.argv[2]
L = open(path).read().split("\n")
out, inside, n = [], False, 0
rx_mul = re.compile(r"v_pk_mul_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1\] op_sel_hi:\[0,0\]$")
rx_fma = re.compile(r"v_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\]$")
pending = set()
for ln in L:
    if ln.startswith("_Z14k_fft2d_inv_mxILi64E") and ":" in ln:
        inside = True
    st = ln.strip()
    if inside and st.startswith("s_endpgm"):
        inside = False
    m = rx_mul.match(st) if inside else None
    f = rx_fma.match(st) if inside else None
    if m:
        d0, d1, a0, a1, b0, b1 = map(int, m.groups())
        pending.add((d0, d1))
        if mode == "mul_scalar":
            # the low product must not clobber a source of the high one
            if d0 in (a0, b0):
                out.append(f"\tv_mul_f32_e32 v{d1}, v{a0}, v{b0}")
                out.append(f"\tv_mul_f32_e32 v{d0}, v{a0}, v{b1}")
            else:
                out.append(f"\tv_mul_f32_e32 v{d0}, v{a0}, v{b1}")
                out.append(f"\tv_mul_f32_e32 v{d1}, v{a0}, v{b0}")
            n += 1
            continue
        if mode == "commute":                # same products, the swapped operand as src0
            out.append(f"\tv_pk_mul_f32 v[{d0}:{d1}], v[{b0}:{b1}], v[{a0}:{a1}] op_sel:[1,0] op_sel_hi:[0,0]"); n += 1
            continue
        if mode == "fma_swap":               # natural halves here (result = (c im, c re)), the consumer swaps its src2
            out.append(f"\tv_pk_mul_f32 v[{d0}:{d1}], v[{a0}:{a1}], v[{b0}:{b1}] op_sel_hi:[0,1]"); n += 1
            continue
        out.append(ln)
        if mode == "mul_twice" and d0 not in (a0, a1, b0, b1) and d1 not in (a0, a1, b0, b1):
            out.append(ln); n += 1
        if mode == "mul_nop":
            out.append("\ts_nop 15"); n += 1
        continue
    if f and mode == "fma_swap" and (int(f.group(7)), int(f.group(8))) in pending:
        out.append("\t" + st + " op_sel:[0,0,1] op_sel_hi:[1,1,0]")
        continue
    if f and mode == "fma_scalar":
        d0, d1, a0, a1, b0, b1, c0, c1 = map(int, f.groups())
        if (c0, c1) in [(22, 23), (4, 5)] or True:
            # v_pk_fma d, a, b, c : lo = a.lo*b.lo+c.lo ; hi = a.hi*b.hi+c.hi  (in place on a is fine lane-wise: lo first then hi)
            if d0 in (a1, b1, c1):
                out.append(f"\tv_fma_f32 v{d1}, v{a1}, v{b1}, v{c1}")
                out.append(f"\tv_fma_f32 v{d0}, v{a0}, v{b0}, v{c0}")
            else:
                out.append(f"\tv_fma_f32 v{d0}, v{a0}, v{b0}, v{c0}")
                out.append(f"\tv_fma_f32 v{d1}, v{a1}, v{b1}, v{c1}")
            n += 1
            continue
    out.append(ln)
open(path, "w").write("\n".join(out))
print(f"[mxi_edit3] {mode}: {n} sites")
