#!/usr/bin/env python
"""One line per kernel family of a tools/phase_trace.py JSON + the two chain lengths."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e)
        continue
    print("==", f, "mode", d["mode"], "update_cus", d.get("update_cus"), "actor_cus", d.get("actor_cus"))
    ks = d["kernels"]
    for k, v in ks.items():
        print("%-16s wgs %5d start %7.2f span %6.2f wg_mean %5.2f  starts %s  phases %s" % (
            k, v["workgroups"], v["t_start_us"], v["span_us"], v["wg_dur_us"]["mean"], v["wg_start_us_hist(0,1,2,4,6,8,12,16+)"],
            {a.split()[0]: b["mean"] for a, b in v["phases_us"].items()}))
    upd = [k for k in ks if not k.startswith("actor") and k != "gather"]
    if upd:
        t0 = min(ks[k]["t_start_us"] for k in upd)
        t1 = max(ks[k]["t_start_us"] + ks[k]["span_us"] for k in upd)
        print("update chain %.1f us (sum of kernel spans %.1f)" % (t1 - t0, sum(ks[k]["span_us"] for k in upd)))
    act = [k for k in ks if k.startswith("actor")]
    if len(act) == 5:
        t0 = ks["actor_conv1"]["t_start_us"]
        t1 = ks["actor_head_env"]["t_start_us"] + ks["actor_head_env"]["span_us"]
        print("actor env step (last of the agent step) %.1f us (sum of kernel spans %.1f)" % (t1 - t0, sum(ks[k]["span_us"] for k in act)))
