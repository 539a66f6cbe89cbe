#!/bin/bash
# round 6, GPU call 15: configs[4] -- panel chunk size of the two-pass transforms: time + counter traffic
cd /root/repo; O=gpurun_out/r06_s15; mkdir -p $O; export TMPDIR=/tmp
D=/root/repo/neuraloperator_amd/libsc_engine_diag.so
for MB in 48 96 144 192 256; do
  echo "== SC_F2P_CHUNK_MB=$MB" >> $O/f2p_chunk_time.txt
  SC_ENGINE_LIB=$D SC_F2P_CHUNK_MB=$MB python scripts/f2p_time.py $D 2>&1 | grep -v "amdgpu.ids\|agreement\|direct-DFT" >> $O/f2p_chunk_time.txt
done
for MB in 48 96 192; do
  for C in FETCH_SIZE WRITE_SIZE; do
    SC_ENGINE_LIB=$D SC_F2P_CHUNK_MB=$MB rocprofv3 --kernel-trace --pmc $C -d $O/pmc_${MB}_$C -o run -- python scripts/f2p_time.py $D > /dev/null 2>&1
  done
  echo "== SC_F2P_CHUNK_MB=$MB (FETCH_SIZE: 32-byte units x 2 on gfx950 -> KB; WRITE_SIZE KB)" >> $O/f2p_chunk_pmc.txt
  python scripts/pmc_summary.py $O/pmc_${MB}_FETCH_SIZE $O/pmc_${MB}_WRITE_SIZE 2>&1 | grep -A3 "k_f2p" >> $O/f2p_chunk_pmc.txt
  rm -rf $O/pmc_${MB}_FETCH_SIZE $O/pmc_${MB}_WRITE_SIZE
done
for MB in 96 192; do
  echo "== bench step, SC_F2P_CHUNK_MB=$MB" >> $O/f2p_chunk_time.txt
  SC_ENGINE_LIB=$D SC_F2P_CHUNK_MB=$MB python bench.py --workload fno2d_1024_m256_c128_b4 --no-extras --no-cpu-baseline --no-gpu-reference --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['cold_start']['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()})" >> $O/f2p_chunk_time.txt
done
cat $O/f2p_chunk_time.txt $O/f2p_chunk_pmc.txt
bash scripts/graph_probe_repeat.sh 16 0 > $O/graph_probe_repeat.txt 2>&1; cat $O/graph_probe_repeat.txt
(timeout 600 python -m pytest tests/test_gpu_graph.py -m gpu -x -q 2>&1 | tail -3)
