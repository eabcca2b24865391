#!/bin/bash
OUT=gpurun_out/last; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "qr" > $OUT/pytest_qr.log 2>&1; grep -E "passed|failed" $OUT/pytest_qr.log | tail -1; grep -E "^(FAILED|ERROR)|Fatal" $OUT/pytest_qr.log | head -3 | cut -c1-200
timeout 100 python tools/bench_agents.py --seconds 3 --cases qr_dqn_pixel_uniform_device 2>/dev/null | cut -c1-150
timeout 400 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -1; grep -E "^(FAILED|ERROR)|Fatal" $OUT/pytest_gpu.log | head -3 | cut -c1-200
