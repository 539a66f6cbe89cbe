#!/bin/bash
# round 3, session 2, GPU call 11: kernel breakdown of the FNO3d 128^3 step and of the 1024^2 step
O=gpurun_out/s2k; mkdir -p $O
export TMPDIR=/tmp
for wl in fno3d_128_m32_c32_b8 fno2d_1024_m256_c128_b4; do
  (cd /tmp && rm -rf /tmp/prof_$wl && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$wl -o run -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 10 --warmup 3 --workload $wl --no-cpu-baseline --no-gpu-reference --no-extras --no-pmc > $GRAFT_REPO_ROOT/$O/bench_$wl.json 2> $GRAFT_REPO_ROOT/$O/bench_$wl.err)
  python scripts/rocprof_summary.py /tmp/prof_$wl > $O/kernel_stats_$wl.txt 2>&1
  head -c 300 $O/bench_$wl.json; echo; head -16 $O/kernel_stats_$wl.txt | cut -c1-175
done
