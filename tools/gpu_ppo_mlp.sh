#!/bin/bash
# one gpurun call for the ppo_mlp path: its GPU tests, then the three ppo_continuous agent lines
# usage: tools/gpu_ppo_mlp.sh <tag> [pytest -k expr]
TAG=${1:-r05a}
K=${2:-}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ -n "$K" ]; then
  timeout 900 python -m pytest tests/test_gpu_ppo_mlp.py -q -m gpu -k "$K" 2>&1 | tail -60 | tee $OUT/tests.log
else
  timeout 900 python -m pytest tests/test_gpu_ppo_mlp.py -q -m gpu 2>&1 | tail -60 | tee $OUT/tests.log
fi
timeout 600 python tools/bench_agents.py --cases ppo_continuous_16,ppo_continuous_16_host,ppo_continuous_16_generic --seconds 4 2>&1 | grep -v Warning | tee $OUT/bench_agents_ppo_continuous.jsonl
