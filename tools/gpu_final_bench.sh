#!/bin/bash
OUT=gpurun_out/final; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; head -c 250 $OUT/bench.json; echo
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; head -c 200 $OUT/bench_driver_cmd.json; echo
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(json.dumps(d['roofline'])[:900])"
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
