#!/bin/bash
TAG=${1:-r02e}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
echo "== gpu tests (agents: learner)"; timeout 1200 python -m pytest tests/test_gpu_agents.py -q -m gpu -x -p no:cacheprovider -k "fused or async" > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log
export DEEPRL_AMD_LIB=$R/deeprl_amd/lib/libdeeprl_amd_trace.so
run() { name=$1; shift; env "$@" timeout 200 python tools/phase_trace.py > $OUT/phase_$name.json 2> $OUT/phase_$name.err; tail -2 $OUT/phase_$name.err | grep -v amdgpu.ids; python tools/phase_summary.py $OUT/phase_$name.json | grep -E "^==|chain|env step|actor"; }
run base A=1
run acu64 DRA_ACTOR_CUS=64
unset DEEPRL_AMD_LIB
for cus in 96 80 64; do
echo "== bench acus $cus"; DRA_ACTOR_CUS=$cus timeout 200 python bench.py --no-cpu-baseline > $OUT/bench_acus$cus.json 2> $OUT/bench_acus$cus.err; head -c 170 $OUT/bench_acus$cus.json; echo; tail -3 $OUT/bench_acus$cus.err | grep -v amdgpu
done
echo "== done"
