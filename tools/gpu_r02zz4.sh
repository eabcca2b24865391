#!/bin/bash
TAG=${1:-r02zz4}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== tests"
timeout 500 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "fast_path or pixel_agents or per_async or env_switches or launcher or schedule_oracle" > $OUT/pytest.log 2>&1
grep -E "passed|failed" $OUT/pytest.log | tail -2 | cut -c1-200; grep -E "^(FAILED|ERROR)" $OUT/pytest.log | head -5 | cut -c1-250
echo "== agents"
timeout 300 python tools/bench_agents.py --seconds 3 --cases c51_pixel_uniform_device,qr_dqn_pixel_uniform_device,c51_pixel_per_device,dqn_pixel_per_device > $OUT/bench_agents.jsonl 2> $OUT/bench_agents.err
cut -c1-200 $OUT/bench_agents.jsonl; tail -3 $OUT/bench_agents.err | cut -c1-200
