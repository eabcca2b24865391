#!/bin/bash
# One gpurun call: full GPU suite on the new defaults (LDS actor fc4, 14-way fc4 K split, index prefetch), config-4 agent
# throughput after the QR / C51 loss-kernel and distributional-actor-head changes, runtime environment knobs.
TAG=${1:-r02zv}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
echo "== full GPU suite"
timeout 600 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3 | cut -c1-300; grep -E "^(FAILED|ERROR)|Error" $OUT/pytest_gpu.log | head -10 | cut -c1-300
echo "== agents (device environments)"
timeout 300 python tools/bench_agents.py --seconds 3 --cases c51_pixel_uniform_device,qr_dqn_pixel_uniform_device,c51_pixel_per_device,dqn_pixel_per_device,dqn_pixel_uniform_device > $OUT/bench_agents.jsonl 2> $OUT/bench_agents.err
cut -c1-200 $OUT/bench_agents.jsonl
DRA_ACTOR_DIST_GEMV=0 timeout 200 python tools/bench_agents.py --seconds 3 --cases c51_pixel_uniform_device,qr_dqn_pixel_uniform_device > $OUT/bench_agents_nogemv.jsonl 2> $OUT/bench_agents_nogemv.err
echo "(DRA_ACTOR_DIST_GEMV=0)"; cut -c1-200 $OUT/bench_agents_nogemv.jsonl
echo "== A/B: runtime knobs (updates/s)"
for rep in 1 2; do
  for cfg in "default|X=1" "devkernarg0|HIP_FORCE_DEV_KERNARG=0" "devkernarg1|HIP_FORCE_DEV_KERNARG=1" "pktcap0|DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "pktcap1|DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"; do
    name=${cfg%%|*}; kv=${cfg#*|}
    env $kv timeout 90 python bench.py --no-cpu-baseline --no-long-run --no-parity-check > $OUT/bench_${name}_$rep.json 2> $OUT/bench_${name}_$rep.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_${name}_$rep.json")); print("%-12s rep $rep: %8.1f updates/s" % ("$name", d["value"]))
except Exception as e:
    print("$name rep $rep: unreadable", e)
PY
  done
done
echo "== rocprofv3 kernel stats: qr / c51 after"
for c in c51_pixel_uniform_device qr_dqn_pixel_uniform_device; do
  (cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$c -- python $R/tools/bench_agents.py --seconds 2 --cases $c > $R/$OUT/prof_$c.log 2>&1)
  echo "== $c"; grep '"case"' $OUT/prof_$c.log | cut -c1-200
  python tools/prof_summary.py $OUT/prof_$c > $OUT/kernel_stats_$c.txt 2>&1; head -26 $OUT/kernel_stats_$c.txt | cut -c1-170
  rm -rf $OUT/prof_$c
done
echo "== done"
