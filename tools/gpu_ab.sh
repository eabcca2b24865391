#!/bin/bash
# quick A/B round: usage gpurun -- 'bash tools/gpu_ab.sh <tag> <masks> [trace-mask] [pytest -k expr]'
TAG=$1; MASKS=$2; TM=${3:-}; KEXPR=${4:-async}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "$KEXPR" > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
timeout 600 python tools/ab_variants.py --masks $MASKS --rounds 3 --steps 1000 > $OUT/ab.jsonl 2> $OUT/ab.err; cut -c1-330 $OUT/ab.jsonl; tail -3 $OUT/ab.err
if [ -n "$TM" ]; then
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -- python $R/bench.py --variant $TM --steps 600 --warmup 100 --no-cpu-baseline > $R/$OUT/prof_bench.json 2> $R/$OUT/prof_bench.err)
  python tools/prof_summary.py $OUT/prof > $OUT/stats.txt 2>&1; python tools/prof_timeline.py $OUT/prof 200 2 > $OUT/timeline.txt 2>&1
  find $OUT/prof -name "*.db" -size +20M -delete
  head -24 $OUT/stats.txt
fi
