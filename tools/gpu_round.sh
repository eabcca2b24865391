#!/bin/bash
# One gpurun call: GPU parity tests, variant A/B, bench, rocprofv3 kernel stats -- everything under gpurun_out/.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_round.sh <tag> [variant-mask-for-bench]'
TAG=${1:-r01}
VAR=${2:-127}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt
echo "== new-kernel tests" ; timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "fused or one_pass or segs" -p no:cacheprovider > $OUT/pytest_new.log 2>&1; tail -5 $OUT/pytest_new.log
echo "== all gpu tests" ; timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log
echo "== variant A/B" ; timeout 400 python tools/ab_variants.py --masks 0,1,3,7,15,31,63,127,16,32,96 --rounds 3 --steps 500 > $OUT/ab_variants.jsonl 2> $OUT/ab_variants.err; cat $OUT/ab_variants.jsonl | cut -c1-400
echo "== bench variant $VAR" ; timeout 300 python bench.py --variant $VAR > $OUT/bench_v$VAR.json 2> $OUT/bench_v$VAR.err; cat $OUT/bench_v$VAR.json
echo "== bench variant 0" ; timeout 300 python bench.py --variant 0 --no-cpu-baseline > $OUT/bench_v0.json 2> $OUT/bench_v0.err; cat $OUT/bench_v0.json
echo "== rocprofv3 kernel trace" ; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --variant $VAR --steps 600 --warmup 100 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof_bench.err)
python tools/prof_summary.py $OUT/prof > $OUT/rocprofv3_kernel_stats.txt 2>&1; head -40 $OUT/rocprofv3_kernel_stats.txt
python tools/prof_timeline.py $OUT/prof 200 2 > $OUT/timeline.txt 2>&1; head -70 $OUT/timeline.txt
find $OUT/prof -name "*.db" -size +20M -delete
echo "== done"
