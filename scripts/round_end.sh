#!/bin/bash
# Round-end GPU pass (one gpurun call): the whole GPU tier, smoke, the default bench line (the driver's command), rocprofv3
# kernel stats over the same command, the PMC passes (HBM traffic of every BASELINE workload -> profiles/pmc_traffic.json;
# matrix-pipe busy cycles of the contraction kernels).  Writes gpurun_out/$1 (default rz); copy the summaries to profiles/.
O=gpurun_out/${1:-rz}; mkdir -p $O
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -60) > $O/pytest.log
(python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12) > $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-reference --no-extras --no-pmc > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err)
python scripts/rocprof_summary.py /tmp/prof > $O/kernel_stats.txt 2>&1
for wl in tucker:32,64,256,256,64,64 dense:8,32,128,128,128,32,32,32 dense:4,128,1024,1024,256,256; do
  k=${wl%%:*}; s=${wl#*:}; n=$(echo $s | tr ',' '_')
  (cd /tmp && rm -rf /tmp/prof_$n && LAYER_KIND=$k LAYER_SHAPE=$s LAYER_REPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1)
  python scripts/rocprof_summary.py /tmp/prof_$n > $O/kernel_stats_${k}_$n.txt 2>&1
done
# bf16 real-tensor I/O at the metric shape (BASELINE configs[1]): kernel stats and the matrix-pipe counters of its step
# (round 5: the forward-type transforms run their row pass on the matrix cores, k_fft2d_fwd_mx)
(cd /tmp && rm -rf /tmp/prof_bf16 && LAYER_IO=bf16 LAYER_REPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bf16 -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1)
python scripts/rocprof_summary.py /tmp/prof_bf16 > $O/kernel_stats_bf16.txt 2>&1
(cd /tmp && rm -rf /tmp/pmc_b && LAYER_IO=bf16 LAYER_REPS=3 timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/pmc_b -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1)
python scripts/pmc_summary.py /tmp/pmc_b > $O/pmc_sq_raw_bf16.txt 2>&1
# sizes off the factorised routes (Darcy grids, resolution changes) beside the reference chain, and their kernel stats
timeout 600 python scripts/odd_sizes_time.py 2>&1 | grep -v amdgpu.ids > $O/odd_sizes.txt
for s in 16,32,421,421,32,32 32,32,141,141,32,32 16,32,421,421,64,64 32,32,85,85,32,32; do
  n=$(echo $s | tr ',' '_')
  (cd /tmp && rm -rf /tmp/prof_$n && LAYER_KIND=dense LAYER_SHAPE=$s LAYER_REPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1)
  echo "== $s" >> $O/odd_stats.txt; python scripts/rocprof_summary.py /tmp/prof_$n | head -10 | cut -c1-150 >> $O/odd_stats.txt
done
timeout 1200 python scripts/pmc_traffic_regen.py $O/pmc_traffic.json > $O/pmc_regen.log 2>&1
(cd /tmp && LAYER_REPS=3 timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/pmc_s -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1)
python scripts/pmc_summary.py /tmp/pmc_s > $O/pmc_sq_raw.txt 2>&1
cat $O/pytest.log $O/smoke.log; head -c 900 $O/bench_default.json; echo; head -10 $O/kernel_stats.txt | cut -c1-170; tail -3 $O/pmc_regen.log | cut -c1-300
