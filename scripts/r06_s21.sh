#!/bin/bash
cd "$(dirname "$0")/session/mxi"; O=/root/repo/gpurun_out/r06_s21; mkdir -p $O
{ for m in 6 7 8 1; do ./pk_forms 2000 $m 1 | grep -v "op_sel_hi:\[1,0\]\|op_sel_hi:\[0,1\]\|op_sel_hi:\[1,1\]"; done; } > $O/pk_forms_more.txt 2>&1
cat $O/pk_forms_more.txt
