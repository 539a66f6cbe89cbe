#!/bin/bash
# probe flakiness: N runs of the graph-probe child (one rank), with and without the quiesce pause
O=gpurun_out/r05_s5; mkdir -p $O
run() { # $1 = tag, rest = env
  ok=0; bad=0
  for i in 1 2 3 4 5 6 7 8; do
    env "${@:2}" RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600 + i)) python bench.py --graph-probe > $O/probe_$1_$i.log 2>&1
    rc=$?
    if [ $rc -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); echo "--- $1 run $i rc=$rc"; grep -v amdgpu.ids $O/probe_$1_$i.log | tail -12; fi
  done
  echo "$1: ok=$ok bad=$bad"
}
run noquiesce SC_GRAPH_QUIESCE_MS=0
run quiesce SC_GRAPH_QUIESCE_MS=500
