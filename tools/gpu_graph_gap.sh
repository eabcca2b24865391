#!/bin/bash
# One short gpurun call: the graph-to-graph boundary microbenchmark (tools/ubench/graph_gap.hip).  usage: gpurun -- 'bash tools/gpu_graph_gap.sh <tag>'
TAG=${1:-gg}; OUT=gpurun_out/$TAG; mkdir -p $OUT; B=tools/ubench/bin
timeout 120 $B/graph_gap 400 > $OUT/graph_gap.jsonl 2> $OUT/graph_gap.err
cat $OUT/graph_gap.jsonl; head $OUT/graph_gap.err
