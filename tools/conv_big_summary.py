#!/usr/bin/env python
"""conv_big_bwd.py's event timings + the rocprofv3 kernel trace + the PMC pass of the same command -> one JSON line per batch:
per kernel group (conv<layer> fwd / bwd) the HIP-event microseconds, the rocprofv3 average duration of the launch's dominant
kernel, FLOP-derived fractions of the fp32-MFMA peak from both, and the counter-derived MFMA utilisation
    SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles),   kernel cycles = rocprofv3 duration x 2.4 GHz (and GRBM_GUI_ACTIVE),
which counts ISSUED MFMAs (tile padding included), beside VALU instructions per MFMA."""
import csv
import glob
import json
import sqlite3
import sys

PEAK, SIMDS, CLOCK = 157.3e12, 1024, 2.4e9
GEOM = {1: "V2Geom<4, 84, 32, 8, 4>|ConvGeom<4, 84, 32, 8, 4>|conv1_fwd_u8", 2: "Geom<32, 20, 64, 4, 2>", 3: "Geom<64, 9, 64, 3, 1>"}


def main():
    ev = json.load(open(sys.argv[1]))
    dbs = glob.glob(sys.argv[2] + "/**/*_results.db", recursive=True)
    rows = []
    if dbs:
        cur = sqlite3.connect(dbs[0]).cursor()
        rows = cur.execute("select name, count(*), avg(end-start), sum(end-start) from kernels group by name").fetchall()
    pmc = {}
    for f in glob.glob(sys.argv[3] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            pmc.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    out = {"batch": ev["batch"], "oneshot_wgrad": ev["oneshot_wgrad"], "dgrad_scatter": ev.get("dgrad_scatter", False), "kernels": {}}
    for key, e in ev["kernels"].items():
        layer, kind = int(key[4]), key.split("_")[1]
        pats = GEOM[layer].split("|")
        def mine(name):
            if not any(p in name for p in pats):
                return False
            is_fwd = "conv_fwd" in name or "conv1_fwd" in name
            return is_fwd if kind == "fwd" else (not is_fwd and ("multi_kernel" in name or "igemm" in name or "wgrad" in name or "dgrad" in name))
        ks = [r for r in rows if mine(r[0])]
        rec = dict(e)
        if ks:
            calls = max(r[1] for r in ks)
            per_call_ns = sum(r[3] for r in ks) / calls          # all kernels of the group per call (bwd may be > 1 launch)
            rec["rocprofv3_us"] = round(per_call_ns / 1e3, 1)
            rec["rocprofv3_frac"] = round(e["flops"] / (per_call_ns * 1e-9) / PEAK, 3)
            rec["rocprofv3_kernels"] = [r[0][:90] for r in ks]
            busy = sum(sum(pmc.get(r[0], {}).get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / max(1, len(pmc.get(r[0], {}).get("SQ_VALU_MFMA_BUSY_CYCLES", [1]))) for r in ks)
            if busy:
                rec["mfma_busy_cycles"] = busy
                rec["mfma_util_counter"] = round(busy / (SIMDS * per_call_ns * 1e-9 * CLOCK), 3)
                gui = sum(sum(pmc.get(r[0], {}).get("GRBM_GUI_ACTIVE", [0])) / max(1, len(pmc.get(r[0], {}).get("GRBM_GUI_ACTIVE", [1]))) for r in ks)
                if gui:
                    rec["grbm_gui_active"] = gui
                valu = sum(sum(pmc.get(r[0], {}).get("SQ_INSTS_VALU", [0])) / max(1, len(pmc.get(r[0], {}).get("SQ_INSTS_VALU", [1]))) for r in ks)
                mf = sum(sum(pmc.get(r[0], {}).get("SQ_INSTS_MFMA", [0])) / max(1, len(pmc.get(r[0], {}).get("SQ_INSTS_MFMA", [1]))) for r in ks)
                if mf:
                    rec["valu_per_mfma"] = round(valu / mf, 2)
                # issued MFMA FLOP / algorithmic FLOP: the fp32 MFMAs run 64 FLOP per busy cycle whatever their shape (32x32x2: 4096 in
                # 64 cycles, 16x16x4: 2048 in 32 -- the scatter-form input gradient of round 6 uses the latter)
                rec["issued_mfma_flop_frac"] = round(busy * 64 / max(e["flops"], 1), 3)
        out["kernels"][key] = rec
    print(json.dumps(out))


if __name__ == "__main__":
    main()
