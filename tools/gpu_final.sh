#!/bin/bash
# The short end-of-round call: the whole GPU suite, smoke(), bench.py (default + the driver's command), the agent-level lines.
# usage: gpurun -- 'bash tools/gpu_final.sh <tag>'
TAG=${1:-final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rm -f gpurun_out/parity_errors.jsonl
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -1; tail -5 $OUT/pytest_gpu.log > $OUT/pytest_gpu_tail.txt
python tools/parity_summary.py gpurun_out/parity_errors.jsonl > $OUT/parity_errors.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; head -c 300 $OUT/bench.json; echo
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; head -c 250 $OUT/bench_driver_cmd.json; echo
timeout 400 python tools/bench_agents.py > $OUT/bench_agents.jsonl 2> $OUT/bench_agents.err; cut -c1-160 $OUT/bench_agents.jsonl
