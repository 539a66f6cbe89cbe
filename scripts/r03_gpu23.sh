#!/bin/bash
# round 3, GPU call 23: the Tucker chain as one autograd node (host time per step)
O=gpurun_out/r3y; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_at_config.py -x -q -k "tucker or tfno or factor" 2>&1 | tail -3) > $O/pytest.log
cat $O/pytest.log
python scripts/tfno_cpu_bound.py 2>&1 | tail -3 > $O/tfno_host.txt; cat $O/tfno_host.txt
python bench.py --no-cpu-baseline --no-gpu-reference --no-pmc > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print(d["ms_per_step"], {k: (v.get("ms_per_step"), v.get("frac_of_8TBs")) for k, v in d["extra"].items() if isinstance(v, dict)})
PY
