#!/bin/bash
OUT=gpurun_out/last3; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(round(d['value'],1), json.dumps(d['cpu_baseline'])[:900])"
