#!/bin/bash
cd "$(dirname "$0")/session/mxi"; O=/root/repo/gpurun_out/r06_s7; mkdir -p $O
{ ./pk_forms 3000 1 1; ./pk_forms 3000 1 2; ./pk_forms 3000 0 2; for h in 64 128 256; do ./mxi fix1.hsaco $h 300; done; } > $O/pk_forms.txt 2>&1
cat $O/pk_forms.txt
