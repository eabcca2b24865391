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


def test_xcd_order_is_a_bijection_that_keeps_groups_on_one_xcd():
    """csrc/common.h xcd_order(): workgroups are dispatched round-robin over 8 XCDs (blockIdx mod 8); the renumbering must
    be a bijection on the grid and must put every sharing group of the remapped range on ONE XCD.  Transcribed here for the
    grids the update launches (conv forward 64 x 13 / 12 / 8, input gradients 32 x 8 / 6, weight gradients 32 x 4 / 8 x 18,
    fc4 forward 28 x 8 with its natural-order remainder) and a role that does not start on a multiple of 8."""
    def xcd_order(bid, first, n_groups, per_group):
        gpx = n_groups // 8
        main = gpx * 8 * per_group
        if bid >= main:
            return bid
        xcd, j = (bid + first) & 7, bid >> 3
        return (xcd * gpx + j // per_group) * per_group + (j % per_group)

    for n_groups, per_group, first in ((64, 13, 0), (64, 12, 0), (64, 8, 0), (32, 8, 0), (32, 6, 0), (32, 4, 256), (8, 18, 192),
                                       (28, 8, 0), (32, 10, 3), (5, 7, 0), (16, 1, 5)):
        total = n_groups * per_group
        got = [xcd_order(b, first, n_groups, per_group) for b in range(total)]
        assert sorted(got) == list(range(total)), (n_groups, per_group, first)
        main = (n_groups // 8) * 8 * per_group
        xcds = {}
        for b in range(main):
            xcds.setdefault(got[b] // per_group, set()).add((b + first) & 7)
        assert all(len(v) == 1 for v in xcds.values()), (n_groups, per_group, first)
        if n_groups >= 8:
            assert len(xcds) == (n_groups // 8) * 8
