#!/bin/bash
cd /root/repo; O=gpurun_out/r06_s17; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_config.py -m gpu -x -q -k "bf16" 2>&1 | tail -4) > $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 >> $O/pytest.log
python bench.py --io bf16 --no-extras --no-cpu-baseline --no-gpu-reference --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['cold_start']['ms_per_step'], d['step_roofline']['frac_of_8TBs'], {k:v['ms'] for k,v in d['stages'].items()})" >> $O/pytest.log
python scripts/mx_fft_ab.py 2>&1 | grep -v amdgpu >> $O/pytest.log
cat $O/pytest.log
