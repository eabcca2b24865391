#!/bin/bash
# SQ counter passes over the conv forward kernels (8 SQ counters per pass).  usage: gpurun -- 'bash tools/pmc_conv.sh <tag> [batches]'
TAG=${1:-pmc_conv}; B=${2:-1024}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$OUT/p1 -- python $R/tools/pmc_conv_workload.py $B > $R/$OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $R/$OUT/p2 -- python $R/tools/pmc_conv_workload.py $B > $R/$OUT/p2.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_fwd" in r["Kernel_Name"]:
                key = (r["Kernel_Name"].split("V2Geom<")[1].split(">")[0], r["Grid_Size"])
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, cs in sorted(acc.items()):
        print(p, key, {k: round(sum(v) / len(v)) for k, v in cs.items()})
PY
tail -2 $OUT/p1.log; tail -2 $OUT/p2.log
find $OUT -name "*kernel_trace*" -size +5M -delete
