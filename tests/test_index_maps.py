"""CPU check of the HIP kernels' index arithmetic: tools/emulate_oneshot.py transcribes, lane by lane, the staging
maps, LDS layouts, K splits and per-MFMA operand addresses of the one-round-trip kernels (conv forward incl. the
multi-tile / persistent shapes, conv input / weight gradients, fc4 forward / input gradient) and compares the
emulated outputs with torch autograd in fp64.  The kernels themselves are tested on the GPU; this keeps their
addressing honest in the GPU-less container."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_emulated_index_maps_agree_with_autograd(capsys):
    spec = importlib.util.spec_from_file_location("emulate_oneshot", os.path.join(ROOT, "tools", "emulate_oneshot.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main()                      # raises AssertionError listing the maps that disagree
    out = capsys.readouterr().out
    assert "FAIL" not in out and out.count(" ok") >= 20
