#!/bin/bash
# round 3, call f: XCD-aware block order (A/B + HBM traffic), true resume, data-parallel graphs
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=gpurun_out/r03f
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_resume.py tests/test_gpu_data_parallel.py tests/test_gpu_env_switches.py -q -m gpu -p no:cacheprovider 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_agents.py tests/test_gpu_pixel_onpolicy.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -5
for X in 1 0 1 0; do
  echo "== DRA_XCD_ORDER=$X"
  DRA_XCD_ORDER=$X timeout 300 python tools/ab_variants.py --masks 1765887 --rounds 3 --steps 1500 2>>$OUT/ab.err | cut -c1-900 | tee -a $OUT/ab_xcd$X.jsonl
done
for X in 1; do
  (cd /tmp && DRA_XCD_ORDER=$X timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch$X -- python $R/tools/pmc_workload.py --steps 40 --variant 787199 > $R/$OUT/pmc_fetch$X.log 2>&1); tail -1 $OUT/pmc_fetch$X.log
  (cd /tmp && DRA_XCD_ORDER=$X timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write$X -- python $R/tools/pmc_workload.py --steps 40 --variant 787199 > $R/$OUT/pmc_write$X.log 2>&1); tail -1 $OUT/pmc_write$X.log
  python tools/pmc_traffic.py $OUT/pmc_fetch$X $OUT/pmc_write$X > $OUT/pmc_traffic_xcd$X.json 2> $OUT/pmc_traffic$X.err
  python - <<PY
import json
d=json.load(open("$OUT/pmc_traffic_xcd$X.json"))
print({k:(round(v['fetch_bytes']/1e6,2),round(v['write_bytes']/1e6,2)) for k,v in d['kernels'].items()})
PY
  find $OUT/pmc_fetch$X $OUT/pmc_write$X -name "*.db" -size +20M -delete
  find $OUT/pmc_fetch$X $OUT/pmc_write$X -name "*kernel_trace*" -size +20M -delete
done
echo "== a2c_pixel 1 process vs 2 ranks on this box"
timeout 200 python bench.py --workload a2c_pixel --steps 300 --warmup 30 2>/dev/null | tail -1 | cut -c1-400 | tee $OUT/bench_a2c_1.json
timeout 300 python bench.py --workload a2c_pixel --gpus 2 --steps 300 --warmup 30 2>/dev/null | tail -1 | cut -c1-400 | tee $OUT/bench_a2c_2rank.json
timeout 200 python bench.py --workload ppo_pixel --steps 30 --warmup 5 2>/dev/null | tail -1 | cut -c1-400 | tee $OUT/bench_ppo_1.json
timeout 300 python bench.py --workload ppo_pixel --gpus 2 --steps 30 --warmup 5 2>/dev/null | tail -1 | cut -c1-400 | tee $OUT/bench_ppo_2rank.json
tail -n 3 $OUT/ab.err
