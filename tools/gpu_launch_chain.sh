#!/bin/bash
# One short gpurun call: the dependent-launch microbenchmark (tools/ubench/launch_chain.hip), both builds, update-partition- and
# actor-partition-sized grids, and the runtime's fence-scope knob.  usage: gpurun -- 'bash tools/gpu_launch_chain.sh <tag>'
TAG=${1:-lc}; OUT=gpurun_out/$TAG; mkdir -p $OUT; B=tools/ubench/bin
{
for wgs in 224 32 832; do
  for bin in lc_plain lc_preload; do
    echo -n "{\"build\": \"$bin\", \"env\": \"\", \"result\": "; timeout 120 $B/$bin $wgs | tr -d '\n'; echo "}"
  done
done
echo -n "{\"build\": \"lc_plain\", \"env\": \"AMD_OPT_FLUSH=0\", \"result\": "; AMD_OPT_FLUSH=0 timeout 120 $B/lc_plain 224 | tr -d '\n'; echo "}"
echo -n "{\"build\": \"lc_preload\", \"env\": \"HIP_FORCE_DEV_KERNARG=0\", \"result\": "; HIP_FORCE_DEV_KERNARG=0 timeout 120 $B/lc_preload 224 | tr -d '\n'; echo "}"
} > $OUT/launch_chain.jsonl 2> $OUT/launch_chain.err
cat $OUT/launch_chain.jsonl
