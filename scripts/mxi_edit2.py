"""Data-flow edits of the scaling sequence of k_fft2d_inv_mx<64> (round 6):  mxi_edit2.py file.s MODE
  swap   : exchange the two v_cndmask that follow each packed scaling multiply (imaginary part selected first)
  hi     : v_mov_b32 v3, v2 in front of each packed scaling multiply (the unselected upper half of src0 holds the scale too)
  dup    : the first v_cndmask issued twice"""
import re, sys
path, mode = sys.argv[1], sys.argv[2]
L = open(path).read().split("\n")
out, inside, n, i = [], False, 0, 0
while i < len(L):
    ln = L[i]
    if ln.startswith("_Z14k_fft2d_inv_mxILi64E") and ":" in ln:
        inside = True
    if inside and ln.strip().startswith("s_endpgm"):
        inside = False
    st = ln.strip()
    if inside and re.match(r"v_pk_mul_f32 v\[\d+:\d+\], v\[2:3\], v\[\d+:\d+\] op_sel_hi:\[0,1\]", st):
        if mode == "hi":
            out.append("\tv_mov_b32_e32 v3, v2")
            n += 1
        out.append(ln)
        # find the two cndmasks behind it
        j = i + 1
        idx = []
        while j < len(L) and len(idx) < 2 and j < i + 6:
            if L[j].strip().startswith("v_cndmask_b32"):
                idx.append(j)
            j += 1
        if len(idx) == 2 and mode in ("swap", "dup"):
            for k in range(i + 1, idx[1] + 1):
                if k == idx[0]:
                    if mode == "swap":
                        out.append(L[idx[1]])
                    else:
                        out.append(L[idx[0]]); out.append(L[idx[0]])
                elif k == idx[1]:
                    out.append(L[idx[0]] if mode == "swap" else L[idx[1]])
                else:
                    out.append(L[k])
            n += 1
            i = idx[1] + 1
            continue
        i += 1
        continue
    out.append(ln)
    i += 1
open(path, "w").write("\n".join(out))
print(f"[mxi_edit2] {mode}: {n} sites")
