#!/bin/bash
TAG=${1:-r02t}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export DEEPRL_AMD_LIB=$R/deeprl_amd/lib/libdeeprl_amd_trace.so
run() { name=$1; shift; env "$@" timeout 200 python tools/phase_trace.py > $OUT/phase_$name.json 2> $OUT/phase_$name.err; tail -2 $OUT/phase_$name.err | grep -v amdgpu.ids; python tools/phase_summary.py $OUT/phase_$name.json | grep -E "^==|chain|env step|actor|rmsprop"; }
run base DRA_ACTOR_CUS=64
run nt DRA_ACTOR_CUS=64 DRA_NT_OPT=1
unset DEEPRL_AMD_LIB
for nt in 0 1; do
echo "== bench nt $nt"; DRA_NT_OPT=$nt DRA_ACTOR_CUS=64 timeout 200 python bench.py --no-cpu-baseline --no-parity-check > $OUT/bench_nt$nt.json 2> $OUT/bench_nt$nt.err; head -c 170 $OUT/bench_nt$nt.json; echo
done
echo "== bench a2c_pixel"; timeout 300 python bench.py --workload a2c_pixel --steps 400 --warmup 100 > $OUT/bench_a2c.json 2> $OUT/bench_a2c.err; cat $OUT/bench_a2c.json | cut -c1-400; tail -2 $OUT/bench_a2c.err | grep -v amdgpu
echo "== bench ppo_pixel"; timeout 300 python bench.py --workload ppo_pixel --steps 100 --warmup 40 > $OUT/bench_ppo.json 2> $OUT/bench_ppo.err; cat $OUT/bench_ppo.json | cut -c1-400; tail -2 $OUT/bench_ppo.err | grep -v amdgpu
