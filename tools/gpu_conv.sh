export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "conv" 2>&1 | tail -3
python - <<'PY'
import json, os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import deeprl_amd as d
from deeprl_amd import ops
from bench_kernels import timeit
d.select_device(0); dev = d.Config.DEVICE
for layer, (c, h, oc, k, s) in ops._CONV_GEOM.items():
    o = (h - k) // s + 1
    for batch in (32, 128, 256, 1024):
        x = (torch.randint(0, 256, (batch, c, h, h), device=dev, dtype=torch.uint8) if layer == 1 else torch.rand(batch, c, h, h, device=dev))
        wt = torch.randn(c * k * k, oc, device=dev) * 0.05; bb = torch.zeros(oc, device=dev)
        t = timeit(lambda: ops.conv_fwd_koc(layer, [x], [wt], [bb], u8_coef=1.0 / 255 if layer == 1 else None), n=30)
        fl = 2.0 * batch * o * o * oc * c * k * k
        print(json.dumps({"conv": layer, "batch": batch, "pt_batch": os.environ.get("DRA_CONV_PT_BATCH", "128"), "us": round(t * 1e6, 1), "TFLOPs": round(fl / t / 1e12, 1), "frac": round(fl / t / 157.3e12, 3)}))
PY
