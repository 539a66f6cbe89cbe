"""Operand-select forms of the packed-fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) in a gfx950 assembly
file, per kernel.  Per source: N = natural halves (op_sel 0, op_sel_hi 1), L = low half in both lanes (0, 0),
H = high half in both lanes (1, 1), X = halves exchanged (1, 0).  Round 6: on MI355X
`v_pk_mul_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[0,0]` (signature mul:LX) returned 0 in the low result of lanes 48-63,
now and then, while another wave of the SIMD executed MFMAs.
  isa_pk_forms.py file.s [--bad-only]"""
import re, sys, collections

RX = re.compile(r"^\s*(v_pk_(?:mul|fma|add)_f32)\s+(.*)$")


def forms(line):
    m = RX.match(line)
    if not m:
        return None
    op, rest = m.group(1), m.group(2)
    nsrc = 3 if "fma" in op else 2
    sel, hi = [0] * 3, [1] * 3
    ms = re.search(r"op_sel:\[([0-9,]+)\]", rest)
    mh = re.search(r"op_sel_hi:\[([0-9,]+)\]", rest)
    if ms:
        for i, v in enumerate(ms.group(1).split(",")):
            sel[i] = int(v)
    if mh:
        for i, v in enumerate(mh.group(1).split(",")):
            hi[i] = int(v)
    code = {(0, 1): "N", (0, 0): "L", (1, 1): "H", (1, 0): "X"}
    return op[5:8] + ":" + "".join(code[(sel[i], hi[i])] for i in range(nsrc))


def scan(path):
    per = collections.OrderedDict()
    cur, has_mfma = None, {}
    for ln in open(path):
        if ln.startswith("_Z") and ln.rstrip().split(";")[0].strip().endswith(":"):
            cur = ln.split(":")[0]
            per[cur] = collections.Counter()
            has_mfma[cur] = False
            continue
        if cur is None:
            continue
        if "v_mfma" in ln:
            has_mfma[cur] = True
        f = forms(ln)
        if f:
            per[cur][f] += 1
    return per, has_mfma


def is_bad(sig):
    """the form seen to fail, and -- not proven either way -- every other form that exchanges the halves of src1"""
    return len(sig) > 5 and sig[5] == "X"


if __name__ == "__main__":
    per, has_mfma = scan(sys.argv[1])
    bad_only = "--bad-only" in sys.argv
    total = collections.Counter()
    for k, c in per.items():
        total.update(c)
        bad = {s: n for s, n in c.items() if is_bad(s)}
        if bad_only and not bad:
            continue
        if c:
            print(f"{k[:90]}  mfma={has_mfma[k]}  " + " ".join(f"{s}={n}" for s, n in sorted(c.items())) + ("   <-- src1 exchanged" if bad else ""))
    print("ALL:", " ".join(f"{s}={n}" for s, n in sorted(total.items())))
