"""Edit the generated assembly of k_fft2d_inv_mx<64> in place (scripts/mxi_variants.sh): insert an instruction before or
after every line that matches a regular expression, inside the H = 64 kernel only.
  mxi_edit_isa.py file.s REGEX INSERT [where]     REGEX: searched in the stripped line ('@' = blank);
                                                  INSERT: e.g. 's_nop@7' or 's_waitcnt@lgkmcnt(0)' ('@' = blank);
                                                  where: after (default) | before"""
import re, sys
path, rx, ins = sys.argv[1], sys.argv[2].replace("@", " "), sys.argv[3].replace("@", " ")
where = sys.argv[4] if len(sys.argv) > 4 else "after"
lines = open(path).read().split("\n")
out, inside, n = [], False, 0
for ln in lines:
    if ln.startswith("_Z14k_fft2d_inv_mxILi64E") and ":" in ln:
        inside = True
        out.append(ln)
        continue
    hit = False
    if inside:
        st = ln.strip()
        if st.startswith("s_endpgm"):
            inside = False
        elif st and not st.startswith((";", ".")) and re.search(rx, st):
            hit = True
    if hit and where == "before":
        out.append("\t" + ins)
    out.append(ln)
    if hit and where != "before":
        out.append("\t" + ins)
    n += hit
open(path, "w").write("\n".join(out))
print(f"[mxi_edit_isa] {n} x '{ins}' {where} /{rx}/")
