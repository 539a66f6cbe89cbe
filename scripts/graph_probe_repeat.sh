#!/bin/bash
# The graph probe child (bench.py --graph-probe, one rank: eager steps through torch.distributed, the native RCCL
# communicator, capture, replays) N times: how many abort?  (round 5: 2 of 8 without the 0.8 s pause; round 6: thread_local
# capture mode, no pause.)   scripts/graph_probe_repeat.sh N [SC_GRAPH_QUIESCE_MS]
N=${1:-12}; export SC_GRAPH_QUIESCE_MS=${2:-0}
ok=0; bad=0
for i in $(seq 1 $N); do
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600 + i)) timeout 120 python bench.py --graph-probe > /dev/null 2>/tmp/probe_$i.err
  rc=$?
  if [ $rc -eq 0 ]; then ok=$((ok + 1)); else bad=$((bad + 1)); echo "  run $i: exit $rc: $(tail -2 /tmp/probe_$i.err | tr '\n' ' ' | cut -c1-200)"; fi
done
echo "graph probe x $N (SC_GRAPH_QUIESCE_MS=$SC_GRAPH_QUIESCE_MS): ok=$ok bad=$bad"
