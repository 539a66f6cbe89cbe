#!/bin/bash
# round 6, GPU call 16: two-term twiddle operand of k_fft2d_fwd_mx (SC_MX_TERMS=2 build) against the shipped three-term one
cd /root/repo; O=gpurun_out/r06_s16; mkdir -p $O
{ for r in 1 2; do for L in neuraloperator_amd/libsc_engine.so neuraloperator_amd/libsc_engine_mx2.so; do echo "== $L"; SC_ENGINE_LIB=$PWD/$L python scripts/mx_fft_ab.py 2>&1 | grep -v amdgpu.ids; done; done
  for L in neuraloperator_amd/libsc_engine.so neuraloperator_amd/libsc_engine_mx2.so; do echo "== bench --io bf16, $L"; SC_ENGINE_LIB=$PWD/$L python bench.py --io bf16 --no-extras --no-cpu-baseline --no-gpu-reference --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['cold_start']['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()})"; done
} > $O/mx_terms_ab.txt 2>&1
cat $O/mx_terms_ab.txt
