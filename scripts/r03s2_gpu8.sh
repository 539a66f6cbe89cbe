#!/bin/bash
O=gpurun_out/s2h; mkdir -p $O
timeout 300 python scripts/tfno_graph_time.py > $O/tfno_graph.txt 2>&1; tail -8 $O/tfno_graph.txt
timeout 300 python scripts/tfno_graph_time.py dense > $O/dense_graph.txt 2>&1; tail -5 $O/dense_graph.txt
