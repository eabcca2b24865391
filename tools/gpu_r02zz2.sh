#!/bin/bash
TAG=${1:-r02zz2}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== PER / replay tests"
timeout 400 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "per or prioritized or sumtree or replay or pixel_agents or fast_path or env_switches" > $OUT/pytest_per.log 2>&1
grep -E "passed|failed" $OUT/pytest_per.log | tail -2 | cut -c1-200; grep -E "^(FAILED|ERROR)" $OUT/pytest_per.log | head -5 | cut -c1-250
echo "== agents"
timeout 300 python tools/bench_agents.py --seconds 3 --cases dqn_pixel_per_device,c51_pixel_per_device,dqn_pixel_per,c51_pixel_uniform_device,a2c_pixel_16,ppo_pixel_8 > $OUT/bench_agents.jsonl 2> $OUT/bench_agents.err
cut -c1-200 $OUT/bench_agents.jsonl
echo "== host profile: dqn_pixel_per_device"; timeout 200 python tools/prof_agents.py dqn_pixel_per_device 3000 2>/dev/null | cut -c1-150 | head -24
