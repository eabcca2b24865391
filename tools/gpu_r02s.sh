#!/bin/bash
TAG=${1:-r02s}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
echo "== gpu tests"; timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log | cut -c1-200
export DEEPRL_AMD_LIB=$R/deeprl_amd/lib/libdeeprl_amd_trace.so
run() { name=$1; shift; env "$@" timeout 200 python tools/phase_trace.py > $OUT/phase_$name.json 2> $OUT/phase_$name.err; tail -2 $OUT/phase_$name.err | grep -v amdgpu.ids; python tools/phase_summary.py $OUT/phase_$name.json | grep -E "^==|chain|env step|_bwd"; }
run acu64 DRA_ACTOR_CUS=64
unset DEEPRL_AMD_LIB
for cus in 64 72; do
echo "== bench acus $cus"; DRA_ACTOR_CUS=$cus timeout 200 python bench.py --no-cpu-baseline > $OUT/bench_acus$cus.json 2> $OUT/bench_acus$cus.err; head -c 170 $OUT/bench_acus$cus.json; echo; tail -3 $OUT/bench_acus$cus.err | grep -v amdgpu
done
