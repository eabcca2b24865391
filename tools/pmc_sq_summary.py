#!/usr/bin/env python
"""rocprofv3 --pmc SQ counter CSVs (one directory per pass) -> per-kernel averages + derived ratios.

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES is per SE;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles (64 per v_mfma_f32_32x32x2_f32) summed over SIMDs.
Derived: wait_frac = WAIT_ANY / WAVE_CYCLES (parked on s_waitcnt / barrier), issue_stall_frac = WAIT_INST_ANY /
WAVE_CYCLES, active_frac = ACTIVE_INST_ANY / WAVE_CYCLES, mfma_cycles_per_simd = MFMA_BUSY / (4 * CUs that ran)."""
import csv
import glob
import json
import sys
from collections import defaultdict

from pmc_traffic import GROUPS


def load(path):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    merged = defaultdict(dict)
    for d in sys.argv[1:]:
        for name, cs in load(d).items():
            for c, v in cs.items():
                merged[name][c] = sum(v) / len(v)
                merged[name]["_launches"] = len(v)
    out = {}
    for group, subs in GROUPS + [("actor_head_env", ["actor_head_env"]), ("actor_conv", ["conv_fwd_v2_kernel"])]:
        for name, cs in merged.items():
            if any(s in name for s in subs):
                key = group if group not in out else group + "|" + name[:60]
                c = {k: round(v, 1) for k, v in cs.items()}
                wc = cs.get("SQ_WAVE_CYCLES", 0)
                if wc:
                    c["wait_frac"] = round(cs.get("SQ_WAIT_ANY", 0) / wc, 3)
                    c["issue_stall_frac"] = round(cs.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
                    c["active_frac"] = round(cs.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
                c["kernel"] = name[:140]
                out[key] = c
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
