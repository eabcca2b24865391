#!/bin/bash
# One gpurun call for the measurement items that are not part of tools/gpu_round.sh: the whole GPU suite, NatureConv backward at
# batch 256 / 512 / 1024 with counters (tools/conv_big_counters.sh), the gather microbench, the agent-level lines.
# usage: gpurun -- 'bash tools/gpu_items.sh <tag> <pytest -k expr | all | none>'
TAG=$1; KEXPR=${2:-all}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_quick.sh $TAG "$KEXPR"
bash tools/conv_big_counters.sh $TAG > $OUT/conv_big_stdout.txt 2>&1
tail -4 $OUT/conv_big_stdout.txt | cut -c1-600
python tools/bench_kernels.py > $OUT/bench_kernels.json 2> $OUT/bench_kernels.err
python - $OUT/bench_kernels.json <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
for k in sorted(d):
    if k.startswith("gather"):
        print(k, json.dumps(d[k])[:400])
P
timeout 300 python tools/bench_agents.py --cases ${AGENT_CASES:-dqn_pixel_uniform,dqn_pixel_uniform_host_async} --seconds 4 > $OUT/bench_agents.jsonl 2> $OUT/bench_agents.err
cut -c1-400 $OUT/bench_agents.jsonl
