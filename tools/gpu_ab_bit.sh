#!/bin/bash
# A/B of ONE variant bit on one box: its bit-identity test, then interleaved long runs (tools/diag_lane.py) with the bit cleared and set.
# usage: gpurun -- 'bash tools/gpu_ab_bit.sh <tag> <ops.VAR_ name> <pytest -k expression>'
TAG=$1; BIT=$2; K=$3; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "$K" > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log | cut -c1-300
OFF=$(python -c "from deeprl_amd import ops; print(ops.get_tuning() & ~ops.$BIT)")
ON=$(python -c "from deeprl_amd import ops; print(ops.get_tuning() | ops.$BIT)")
for rep in 1 2 3; do for V in $OFF $ON; do
  DRA_TUNING=$V python tools/diag_lane.py 4000 2>>$OUT/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'variant':$V,'$BIT': bool($V == $ON),'updates_per_s':d['updates_per_s'],'us_per_step':d['us_per_step']}))" | tee -a $OUT/ab_$BIT.jsonl
done; done
tail -2 $OUT/err.txt
