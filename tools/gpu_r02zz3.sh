#!/bin/bash
TAG=${1:-r02zz3}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== tests"
timeout 500 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "linear or a2c or ppo or onpolicy or launcher or data_parallel or two_ranks" > $OUT/pytest.log 2>&1
grep -E "passed|failed" $OUT/pytest.log | tail -2 | cut -c1-200; grep -E "^(FAILED|ERROR)" $OUT/pytest.log | head -5 | cut -c1-250
echo "== agents"
timeout 300 python tools/bench_agents.py --seconds 3 --cases a2c_pixel_16,ppo_pixel_8,dqn_pixel_uniform_generic > $OUT/bench_agents.jsonl 2> $OUT/bench_agents.err
cut -c1-200 $OUT/bench_agents.jsonl
DRA_LINEAR_GEMV=0 timeout 300 python tools/bench_agents.py --seconds 3 --cases a2c_pixel_16,ppo_pixel_8 > $OUT/bench_agents_nogemv.jsonl 2> $OUT/bench_agents_nogemv.err
echo "(DRA_LINEAR_GEMV=0)"; cut -c1-200 $OUT/bench_agents_nogemv.jsonl
