"""The bench line's contract, checked on the newest committed `python bench.py` output (profiles/r*_bench.json) and on
bench.py's own helpers that read committed profiler summaries -- no GPU needed.  The driver parses this line; a field that
goes missing here goes missing there."""
import glob
import importlib.util
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest(pattern):
    def key(p):
        m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(p))
        return (int(m.group(1)), len(m.group(2)), m.group(2))
    files = [p for p in glob.glob(os.path.join(ROOT, "profiles", pattern)) if re.match(r"r\d+[a-z]*_bench\.json$", os.path.basename(p))]
    assert files, pattern
    return max(files, key=key)


def test_committed_bench_line_has_the_contract_fields():
    path = _newest("r*_bench.json")
    b = json.load(open(path))
    for k, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                   ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str)):
        assert isinstance(b[k], typ), (path, k)
    assert "vs_baseline" in b and b["vs_baseline"] is None            # BASELINE.md holds no published number for this metric
    assert b["metric"] == "gradient-updates/sec" and b["scaling"] == "weak" and b["dtype"] == "f32" and b["data"] == "synthetic"
    assert isinstance(b["config"]["workload"], str) and "model" not in b["config"]
    assert abs(b["value"] - b["n_gpus"] * 1e3 / b["ms_per_step"]) / b["value"] < 1e-6
    r = b["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 0
    # round 6: the dominant kernel of the TIMED pipeline, by name -- the chained forward launch (conv1 + conv2 + conv3 of both nets),
    # replayed alone in the same run; the chained backward and rounds 2-5's per-layer headline (conv2's backward launch) beside it
    assert r["kernel"] == "conv_fwd_chain" and r["peak"] == 157.3 and r["algorithmic_flops"] == 990380032
    assert r["graph_replay"]["kernel_us"] > 5 and r["frac_kernel_alone_warm"] >= r["frac"]
    bc = r["backward_chain"]
    assert bc["kernel"] == "conv_bwd_chain" and 0 < bc["frac"] < 1 and bc["algorithmic_flops"] == 780664832
    pl = r["per_layer_launch"]
    assert pl["kernel"] == "conv2_bwd_x" and 0 < pl["frac"] < 1
    assert [m["kernel"] for m in pl["mfma_kernels"]] and all(0 < m["frac"] < 1 for m in pl["mfma_kernels"])
    assert pl["longest_kernel"]["bound"] == "fabric/MALL"
    c = b["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str) and c["unit"]
    assert 0.5 < c["port_vs_reference"]["port_over_reference"] < 2.0
    assert b["parity_check"]["ok"] is True


def test_bench_reads_the_newest_committed_profiler_summaries():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    k = mod.rocprof_kernel("conv2_bwd_x", 339738624)
    assert k is not None and 5e-3 < k["avg_ms"] < 3e-2 and 0.05 < k["frac"] < 0.6 and k["file"].startswith("profiles/")
    assert mod.rocprof_kernel("rmsprop_step", 0)["avg_ms"] > 5e-3
    t = mod.pmc_traffic("conv2_bwd_x")
    assert t is not None and 4e6 < t < 4e7
    # the chained launches of the timed pipeline (round 6)
    kc = mod.rocprof_kernel("conv_fwd_chain", 990380032)
    assert kc is not None and 2e-2 < kc["avg_ms"] < 6e-2 and 0.05 < kc["frac"] < 0.6
    kb = mod.rocprof_kernel("conv_bwd_chain", 780664832)
    assert kb is not None and 1.5e-2 < kb["avg_ms"] < 5e-2
    tc = mod.pmc_traffic("conv_fwd_chain")
    assert tc is not None and 2e7 < tc < 2e8          # (with the deferred fc4 optimizer segment's riders: ~59 MB algorithmic)


def test_newest_bench_line_round5_fields():
    """Round 5 additions to the driver's record: every passing parity step is judged, the headline fraction names its source and
    keeps the live event-pair reading beside it, and the other BASELINE configs' agent lines travel with the headline."""
    b = json.load(open(_newest("r*_bench.json")))
    pc = b["parity_check"]
    assert pc["steps_judged"] == sum(1 for s in pc["steps"] if s["within_tolerance"] or not s.get("excused", False))
    assert pc["steps_judged"] == pc["steps_checked"] or any(s.get("excused") for s in pc["steps"])
    r = b["roofline"]
    assert "frac_source" in r and r["frac_hip_events"] is None                  # (a chained launch has no per-launch event pair)
    pl = r["per_layer_launch"]
    assert pl["frac_hip_events"] > 0 if "frac_hip_events" in pl else pl["frac"] > 0
    other = b["agent_api"]["other_configs"]
    for name in ("categorical_dqn_pixel", "categorical_dqn_pixel_prioritized_replay", "quantile_regression_dqn_pixel", "a2c_pixel_16",
                 "ppo_pixel_8", "ppo_continuous_16"):
        assert other[name]["env_steps_per_s"] > 0 and other[name]["updates_per_s"] > 0, name
    # BASELINE configs[2] on the device (>= 150 k env-steps/s and >= 50 k minibatch updates/s)
    assert other["ppo_continuous_16"]["env_steps_per_s"] >= 150e3 and other["ppo_continuous_16"]["updates_per_s"] >= 50e3


def test_comm_check_refuses_more_ranks_than_gpus():
    """`bench.py --gpus N --comm-check` on a box with fewer GPUs prints one JSON line saying so and exits 3 (nothing launched)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("box has two GPUs: the check would run")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm-check"], capture_output=True, text=True,
                       timeout=300)
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 3 and rec["comm_check"] and rec["enough_devices"] is False and rec["n_gpus_requested"] == 2
