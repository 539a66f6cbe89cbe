#!/bin/bash
cd "$(dirname "$0")/session/mxi"; O=/root/repo/gpurun_out/r06_s8; mkdir -p $O
{ for m in 1 2 3 4 5 0; do ./pk_forms 2000 $m 1; done; } > $O/pk_forms.txt 2>&1
cat $O/pk_forms.txt
