#!/usr/bin/env python
"""rocprofv3 --pmc counter CSVs -> HBM bytes per launch per kernel.

    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> > profiles/rNN_pmc_traffic.json

Units and corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB, derived from the
L2's memory-side request counters; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane)
coalesced streaming read, other widths and WRITE_SIZE are uncalibrated.  The gather launch of
tools/pmc_workload.py with 32768 samples moves a known number of bytes, so its measured/known ratios are
reported as `calibration` and the corrected figures use them:  fetch_bytes = FETCH_SIZE*1024 / fetch_ratio.
"""
import csv
import glob
import json
import sys

GROUPS = [  # learner kernel group -> substring(s) of the rocprof kernel name
    ("gather", ["ring_gather_kernel"]),
    ("conv1_fwd", ["conv_fwd_v2_kernel<V2Geom<4, 84"]),
    ("conv2_fwd", ["conv_fwd_v2_kernel<V2Geom<32, 20"]),
    ("conv3_fwd", ["conv_fwd_v2_kernel<V2Geom<64, 9"]),
    ("fc4_fwd", ["LinFwd<", "LinFwdSlabsOne<"]),
    ("head_loss", ["head_fused_kernel"]),
    ("fc4_bwd", ["LinDgradOne<", "igemm_kernel<LinDgrad<", "multi_kernel<IgemmRole<LinDgrad<"]),
    ("fc4_bwd_w", ["igemm_kernel<LinWgrad<"]),
    ("conv3_bwd", ["multi_kernel<ConvDgradLin<ConvGeom<64, 9, 64, 3, 1>", "multi_kernel<ConvDgradOne<ConvGeom<64, 9, 64, 3, 1>"]),
    ("conv2_bwd", ["multi_kernel<ConvDgradLin<ConvGeom<32, 20, 64, 4, 2>", "multi_kernel<ConvDgradOne<ConvGeom<32, 20, 64, 4, 2>"]),
    ("conv1_bwd_w", ["multi_kernel<ConvWgradOne<ConvGeom<4, 84, 32, 8, 4>"]),
    ("conv_fwd_chain", ["conv_fwd_chain_kernel"]),     # DRA_VAR_FWD_CHAIN: conv1 + conv2 + conv3 forward of both nets (+ the fc4 riders)
    ("conv_bwd_chain", ["bwd_chain_kernel"]),          # DRA_VAR_BWD_CHAIN: conv3 / conv2 / conv1 backward + the two slab folds
    ("grad_norm", ["grad_sqnorm_kernel", "grad_fold_norm_kernel", "fold_norm_kernel", "clip_step_kernel"]),
    ("rmsprop_step", ["rmsprop_step_kernel", "late_step_kernel"]),
    ("actor_fc4", ["actor_fc4_kernel", "actor_fc4_planes", "actor_c3fc4_kernel"]),
    ("actor_persist", ["actor_persist_kernel"]),
]


def load(path, counter):
    rows = {}
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                name = r["Kernel_Name"]
                grid = int(r.get("Grid_Size", 0) or 0)
                rows.setdefault((name, grid), []).append(float(r["Counter_Value"]))
    return rows


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {"units": "bytes per launch; raw = counter KiB * 1024", "kernels": {}}
    # calibration launch: the gather with 32768 samples (grid = 32768 * 5 workgroups * 256 threads)
    known_rd, known_wr = 32 * 1024 * 5 * 7056, 2 * 32 * 1024 * 4 * 7056
    cal = {"fetch_ratio": None, "write_ratio": None, "known_read_bytes": known_rd, "known_write_bytes": known_wr}
    for (name, grid), v in fetch.items():
        if "ring_gather_kernel" in name and grid == 32 * 1024 * 5 * 256:
            cal["fetch_raw_bytes"] = 1024 * sum(v) / len(v)
            cal["fetch_ratio"] = cal["fetch_raw_bytes"] / known_rd
    for (name, grid), v in write.items():
        if "ring_gather_kernel" in name and grid == 32 * 1024 * 5 * 256:
            cal["write_raw_bytes"] = 1024 * sum(v) / len(v)
            cal["write_ratio"] = cal["write_raw_bytes"] / known_wr
    out["calibration"] = cal
    fr = cal["fetch_ratio"] or 0.5
    wr = cal["write_ratio"] or 1.0
    for group, subs in GROUPS:
        acc = {"fetch": [], "write": [], "names": set()}
        for table, key in ((fetch, "fetch"), (write, "write")):
            for (name, grid), v in table.items():
                if any(s in name for s in subs) and not ("ring_gather_kernel" in name and grid == 32 * 1024 * 5 * 256):
                    acc[key] += v
                    acc["names"].add(name.split("(")[0][:120])
        if not acc["fetch"] and not acc["write"]:
            continue
        f = 1024 * sum(acc["fetch"]) / max(1, len(acc["fetch"]))
        w = 1024 * sum(acc["write"]) / max(1, len(acc["write"]))
        out["kernels"][group] = {"launches_fetch_pass": len(acc["fetch"]), "fetch_raw_bytes": f, "write_raw_bytes": w,
                                 "fetch_bytes": f / fr, "write_bytes": w / wr, "hbm_bytes": f / fr + w / wr,
                                 "names": sorted(acc["names"])}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
