#!/bin/bash
# round 2, GPU call 12: whole GPU tier (incl. half / mixed precision goldens) + smoke + default bench line
O=gpurun_out/r2l; mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -8) > $O/pytest_gpu.log
cat $O/pytest_gpu.log
(timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > $O/smoke.log
cat $O/smoke.log
(timeout 400 python bench.py 2>&1 | tail -1) > $O/bench_default.json
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print(d["ms_per_step"], d["value"], d["step_roofline"]["frac_of_8TBs"], d["roofline"]["kernel"], d["roofline"]["frac"], d["cpu_baseline"], d.get("gpu_reference_baseline"))
PY
