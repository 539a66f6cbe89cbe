#!/bin/bash
# Round 3, round-end GPU pass (one gpurun call): the whole GPU tier, smoke, the default bench line (driver's command),
# rocprofv3 kernel stats over the same command, PMC passes over the layer step (HBM traffic; matrix-pipe busy cycles
# of the contraction kernels).  Writes gpurun_out/r3z; the summaries are copied to profiles/r03_*.
O=gpurun_out/r3z; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/pytest.log
(python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12) > $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-reference --no-extras --no-pmc > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err)
python scripts/rocprof_summary.py /tmp/prof > $O/kernel_stats.txt 2>&1
(cd /tmp && LAYER_REPS=3 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1)
(cd /tmp && LAYER_REPS=3 timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1)
(cd /tmp && LAYER_REPS=3 timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/pmc_s -o run -- python $GRAFT_REPO_ROOT/scripts/layer_one.py > /dev/null 2>&1)
python scripts/pmc_summary.py /tmp/pmc_f /tmp/pmc_w > $O/pmc_traffic_raw.txt 2>&1
python scripts/pmc_summary.py /tmp/pmc_s > $O/pmc_sq_raw.txt 2>&1
cat $O/pytest.log $O/smoke.log; head -c 700 $O/bench_default.json; echo; head -10 $O/kernel_stats.txt | cut -c1-170; grep -A3 "k_fft2d\|k_modegemm_dma" $O/pmc_traffic_raw.txt | head -40; grep -A8 "k_modegemm_dma" $O/pmc_sq_raw.txt | head -40
