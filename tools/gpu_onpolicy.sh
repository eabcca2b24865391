#!/bin/bash
# on-policy pixel agents: parity subset + agent lines (+ optional rocprofv3 kernel stats): gpu_onpolicy.sh <tag> [prof]
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_agents.py tests/test_gpu_dropin_run.py -x -q \
  -k "policy_heads or categorical or pair or onpolicy or a2c or ppo or in_place or dropin or deferred or fold or sqnorm or gather_rows or fused_rollout or linear_bwd or fc4_and or fc4_policy or rollout_fc4" > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
timeout 300 python tools/bench_agents.py --seconds 3 --cases ${CASES:-a2c_pixel_16,ppo_pixel_8} > $OUT/bench_agents.jsonl 2>&1
grep '"case"' $OUT/bench_agents.jsonl | cut -c1-220
if [ "$2" = "prof" ]; then bash tools/gpu_prof_agents.sh $TAG a2c_pixel_16,ppo_pixel_8 > $OUT/prof.log 2>&1; fi
