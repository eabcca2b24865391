#!/bin/bash
TAG=${1:-r02w}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
echo "== gather tests"; timeout 600 python -m pytest tests/test_gpu_properties.py tests/test_gpu_kernels.py -q -m gpu -x -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2
for i in 1 2; do echo "== bench $i"; timeout 300 python bench.py --no-cpu-baseline --no-parity-check > $OUT/bench$i.json 2> $OUT/bench$i.err; python - <<PY
import json
d=json.load(open("$OUT/bench$i.json")); print(d["value"], d["host"])
PY
done
export DEEPRL_AMD_LIB=$R/deeprl_amd/lib/libdeeprl_amd_trace.so
timeout 200 python tools/phase_trace.py > $OUT/phase_async.json 2> $OUT/phase.err; python tools/phase_summary.py $OUT/phase_async.json | grep -E "head|chain|env step|gather"
unset DEEPRL_AMD_LIB
echo "== kernel microbench gather"; for s in 1 0; do DRA_GATHER_SAMPLE=$s timeout 300 python tools/bench_kernels.py > $OUT/kernel_microbench_s$s.json 2> $OUT/kmb.err; python - <<PY
import json
d=json.load(open("$OUT/kernel_microbench_s$s.json")); print("sample-shape=$s", {k: (round(v["read_plus_write_GBps"]), round(v["frac_hbm_total_8TBps"],3), round(v["frac_hbm_read_8TBps"],3)) for k, v in d.items() if k.startswith("gather")})
PY
done
