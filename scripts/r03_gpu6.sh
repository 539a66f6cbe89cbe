#!/bin/bash
# round 3, GPU call 6: (a) step-level A-B of the forward kernel's prefetch depth 2 / three workgroups per CU against
# the product (depth 1 / four) and the round-2 row stage, many rounds; (b) the two-pass inverse row kernel with all
# panel loads requested up front; (c) kernel stats of the TFNO step
O=gpurun_out/r3f; mkdir -p $O
P=neuraloperator_amd
KINDS=tf,ti,step ROUNDS=21 REPS=40 timeout 400 python scripts/pair_ab.py $P/libsc_engine.so $P/libsc_engine_pf2occ3.so $P/libsc_engine_r2base.so 2>&1 | grep -v amdgpu.ids | tail -5 > $O/step_ab.txt
cat $O/step_ab.txt
B="python bench.py --no-cpu-baseline --no-gpu-reference --no-extras --steps 10 --warmup 3"
for wl in fno2d_192_m64_c64_b32 fno2d_1024_m256_c128_b4; do
  $B --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err
  python - <<PY
import json
d = json.load(open("$O/bench_$wl.json"))
print("$wl", d["config"]["engine_path"], "ms/step", d["ms_per_step"], "step frac", d["step_roofline"]["frac_of_8TBs"], {k: v["ms"] for k, v in d["stages"].items()})
PY
done 2>&1 | tee $O/f2p_after_unroll.txt
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_tfno -o run -- python $GRAFT_REPO_ROOT/scripts/tfno_time.py factorized > $GRAFT_REPO_ROOT/$O/tfno_time.txt 2>&1)
python scripts/rocprof_summary.py /tmp/prof_tfno > $O/tfno_kernel_stats.txt 2>&1; head -30 $O/tfno_kernel_stats.txt | cut -c1-200
