#!/bin/bash
# One short gpurun call: the in-kernel hand-over microbenchmark (tools/ubench/ll_handover.hip), whole device and the actor's
# 32-CU partition.  usage: gpurun -- 'bash tools/gpu_ll_handover.sh <tag>'
TAG=${1:-llh}; OUT=gpurun_out/$TAG; mkdir -p $OUT; B=tools/ubench/bin
{ for m in 0; do timeout 90 $B/ll_handover $m; done; } > $OUT/ll_handover.jsonl 2> $OUT/ll_handover.err
cat $OUT/ll_handover.jsonl; cat $OUT/ll_handover.err | head
