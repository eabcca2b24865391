#!/bin/bash
# A/B of a kernel change at the rollout batch sizes: conv_big_bwd (HIP events) for two builds, interleaved.
TAG=${1:-abd}; A=$2; B=$3; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do for b in 256 1024; do for L in $A $B; do
  echo -n "{\"lib\": \"$(basename $L .so)\", \"batch\": $b, \"rep\": $rep, \"result\": "; DEEPRL_AMD_LIB=$L timeout 300 python tools/conv_big_bwd.py $b 40 2>/dev/null | tail -1 | tr -d '\n'; echo "}"
done; done; done | tee $OUT/conv_big_ab.jsonl | cut -c1-600
