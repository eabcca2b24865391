#!/bin/bash
TAG=${1:-r02b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export DEEPRL_AMD_LIB=$R/deeprl_amd/lib/libdeeprl_amd_trace.so
run() { name=$1; shift; env "$@" timeout 200 python tools/phase_trace.py > $OUT/phase_$name.json 2> $OUT/phase_$name.err; tail -2 $OUT/phase_$name.err; python tools/phase_summary.py $OUT/phase_$name.json | grep -E "^==|chain|env step|conv._bwd|rmsprop|conv._fwd"; }
run base A=1
run oneround DRA_WG3_MTG=6 DRA_DGRAD_PT=4
run oneround_persist DRA_WG3_MTG=6 DRA_DGRAD_PT=4 DRA_CONV_PT_BATCH=32
run persist DRA_CONV_PT_BATCH=32
run oneround_persist_acu64 DRA_WG3_MTG=6 DRA_DGRAD_PT=4 DRA_CONV_PT_BATCH=32 DRA_ACTOR_CUS=64
run acu64 DRA_ACTOR_CUS=64
unset DEEPRL_AMD_LIB
echo "== bench oneround+persist"; DRA_WG3_MTG=6 DRA_DGRAD_PT=4 DRA_CONV_PT_BATCH=32 timeout 200 python bench.py --no-cpu-baseline > $OUT/bench_oneround_persist.json 2> $OUT/bench_oneround_persist.err; head -c 200 $OUT/bench_oneround_persist.json; echo
echo "== done"
