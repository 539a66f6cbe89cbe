#!/bin/bash
# round 3, GPU call 14: roofline.traffic read from the PMC counters inside the bench run (two rocprofv3 sub-passes)
O=gpurun_out/r3m; mkdir -p $O
( time python bench.py --no-extras --no-cpu-baseline --no-gpu-reference > $O/bench_live_pmc.json 2> $O/bench_live_pmc.err ) 2> $O/time.txt
python - <<PY
import json
d = json.load(open("$O/bench_live_pmc.json"))
print(d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["traffic"], d["roofline"]["alg_bytes_per_launch"])
print(d["roofline"]["traffic_source"])
PY
tail -3 $O/time.txt
( time python bench.py --workload fno3d_128_m32_c32_b8 --no-extras --no-cpu-baseline --no-gpu-reference > $O/bench_live_pmc_3d.json 2> $O/bench_live_pmc_3d.err ) 2> $O/time3d.txt
python - <<PY
import json
d = json.load(open("$O/bench_live_pmc_3d.json"))
print(d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["traffic"], d["roofline"]["alg_bytes_per_launch"])
print(d["roofline"]["traffic_source"][:200])
PY
tail -3 $O/time3d.txt
