#!/bin/bash
TAG=${1:-r02zz5}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== tests"
timeout 500 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "c51 or qr or loss or fast_path or pixel_agents or env_switches or linear" > $OUT/pytest.log 2>&1
grep -E "passed|failed" $OUT/pytest.log | tail -2 | cut -c1-200; grep -E "^(FAILED|ERROR)|Fatal" $OUT/pytest.log | head -5 | cut -c1-250
echo "== agents"
timeout 300 python tools/bench_agents.py --seconds 3 --cases c51_pixel_uniform_device,qr_dqn_pixel_uniform_device,c51_pixel_per_device 2>/dev/null | cut -c1-150
echo "(DRA_HEAD_GEMV=1)"; DRA_HEAD_GEMV=1 timeout 300 python tools/bench_agents.py --seconds 3 --cases c51_pixel_uniform_device,qr_dqn_pixel_uniform_device 2>/dev/null | cut -c1-150
