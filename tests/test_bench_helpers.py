"""CPU checks of bench.py's helpers (the contract file must not break on a detail): the rocprofv3 summary parser behind
`roofline.rocprofv3`, and the worker mode of the CPU replica baseline."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_rocprof_summary_parser_reads_the_committed_profile():
    b = _bench()
    rec = b.rocprof_kernel("conv2_bwd_x", 339738624)
    assert rec is not None and rec["file"].startswith("profiles/") and rec["calls"] > 100
    assert 5e-3 < rec["avg_ms"] < 3e-2                      # a 10-20 us kernel
    assert abs(rec["frac"] - 339738624 / (rec["avg_ms"] * 1e-3) / 1e12 / 157.3) < 1e-12
    assert b.rocprof_kernel("no_such_group", 1.0) is None


def test_cpu_replica_worker_prints_one_json_line():
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-replica-worker", "1.0", "--ring", "600"], env=env,
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-1500:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["updates"] >= 1 and rec["seconds"] >= 1.0
