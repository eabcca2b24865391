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


def test_conv1_full_k_throughput_kernel_maps():
    """conv_v2.hip conv1_fwd_u8_tp_kernel, transcribed: (i) the launcher's (samples per group, tile parts, groups) and the
    kernel's (group, wave) -> tiles enumeration cover every (sample, position tile) exactly once for the batches the GPU
    tests run and the regime boundaries; (ii) operand o = 4 j + q of a tile reads byte c * 7056 + (4 oh + kh) * 84 + 4 ow + kw
    of the sample, with (c, kh, kw) running over all K = 256 taps exactly once, quarter q = (channel pair q & 1, tap half
    q >> 1) in the latency shape's order (areg[q][j] = weight row (2 (q & 1) + h) * 64 + (q >> 1) * 32 + j); (iii) evaluated in
    fp64 on a small batch, the four-chain sum equals F.conv2d."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    H, OH, P, TPS, IMG, HH = 84, 20, 400, 13, 4 * 84 * 84, 84 * 84

    def plan(batch, per=512):
        spg = 2 if batch >= 2 * per else 1
        tp = 1 if (spg == 2 or batch >= per) else (2 if 2 * batch >= per else 4)
        n_groups = ((batch + spg - 1) // spg) * tp
        return spg, tp, n_groups

    for batch in (384, 389, 511, 512, 1023, 1024, 1025, 128, 131, 255):
        spg, tp, n_groups = plan(batch)
        seen = np.zeros((batch, TPS), dtype=np.int64)
        for g in range(n_groups):
            b0 = (g // tp) * spg
            ns = min(spg, batch - b0)
            part = g - (g // tp) * tp
            t_lo, t_hi = (part * TPS // tp, (part + 1) * TPS // tp) if tp > 1 else (0, ns * TPS)
            assert tp == 1 or ns == 1
            for wave in range(4):
                for tile in range(t_lo + wave, t_hi, 4):
                    s = tile // TPS
                    seen[b0 + s, tile - s * TPS] += 1
        assert (seen == 1).all(), batch

    def boff(o):
        j, q = o >> 2, o & 3
        t = (q >> 1) * 32 + j
        kh, kw = divmod(t, 8)
        return (q & 1) * 2 * HH + kh * H + kw

    taps = set()
    for h in range(2):
        for o in range(128):
            j, q = o >> 2, o & 3
            off = h * HH + boff(o)
            c, rem = divmod(off, HH)
            kh, kw = divmod(rem, H)
            assert c == 2 * (q & 1) + h and kh * 8 + kw == (q >> 1) * 32 + j and kw < 8
            taps.add((c, kh, kw))
    assert len(taps) == 256

    rs = np.random.RandomState(0)
    x = rs.randint(0, 256, size=(2, 4, H, H)).astype(np.uint8)
    w = rs.standard_normal((32, 4, 8, 8))
    wt = w.transpose(1, 2, 3, 0).reshape(256, 32)             # KOC: row (c * 8 + kh) * 8 + kw
    lut = np.arange(256) / 255.0
    smem = x.reshape(2, IMG)
    y = np.zeros((2, 32, P))
    for s in range(2):
        for tile in range(TPS):
            for li in range(32):
                p = min(tile * 32 + li, P - 1)
                oh, ow = divmod(p, OH)
                acc = np.zeros((4, 32))
                for h in range(2):
                    base = h * HH + oh * 4 * H + ow * 4
                    for o in range(128):
                        j, q = o >> 2, o & 3
                        row = (2 * (q & 1) + h) * 64 + (q >> 1) * 32 + j
                        acc[q] += wt[row] * lut[smem[s, base + boff(o)]]
                if tile * 32 + li < P:
                    y[s, :, tile * 32 + li] = (acc[0] + acc[1]) + (acc[2] + acc[3])
    ref = F.conv2d(torch.from_numpy(x.astype(np.float64) / 255.0), torch.from_numpy(w), stride=4).numpy().reshape(2, 32, P)
    np.testing.assert_allclose(y, ref, rtol=1e-12, atol=1e-12)


def test_conv2_wide_staging_equals_the_row_shaped_map():
    """conv_v2.hip V2StageWide (128-bit staging of conv2's whole fp32 image): float4 number idx of the contiguous sample lands
    as two 64-bit LDS writes -- elements 0 / 2 at dst, dst + 1 and elements 1 / 3 at dst + WPH, dst + WPH + 1 -- exactly where
    the row-shaped staging puts the same pixels: img[c * CS + row * RW + lds_col(col)], lds_col(iw) = (iw % S) * WPH + iw / S.
    conv3 (S = 1): lds_col is the identity and RW = H, CS = H * H: the LDS image is the memory image."""
    C, H, S = 32, 20, 2
    WPH = (H + S - 1) // S
    RW = S * WPH
    NR = H
    CS = NR * RW
    where = {}
    for idx in range(C * H * H // 4):
        e = 4 * idx
        c, rem = divmod(e, H * H)
        row, col = divmod(rem, H)
        assert col % 4 == 0
        dst = c * CS + row * RW + col // 2
        assert dst % 2 == 0 and (dst + WPH) % 2 == 0          # 64-bit aligned writes
        for k, a in ((0, dst), (2, dst + 1), (1, dst + WPH), (3, dst + WPH + 1)):
            where[(c, row, col + k)] = a
    assert len(where) == C * H * H and len(set(where.values())) == C * H * H
    for (c, row, iw), a in where.items():
        assert a == c * CS + row * RW + (iw % S) * WPH + iw // S
    C3, H3 = 64, 9
    assert (1 * ((H3 + 0) // 1)) == H3 and (C3 * H3 * H3) % 4 == 0      # RW == H, whole float4s: a straight copy


def test_round4_small_kernel_maps():
    """Transcriptions of the round-4 latency kernels' work decompositions (the kernels themselves run in the GPU suite):
    (i) linear_gemv_rows_kernel (igemm.hip): 8 waves = 2 output rows x 4 K quarters, lane l of quarter q holds float4
        v0 + l + 64 i -- every float4 of every row is covered exactly once for the K the GPU tests use, and the
        (q0 + q1) + (q2 + q3) combination of the quarter sums is the dot product;
    (ii) linear_heads_rows_kernel: one wave per input row, k = lane + 64 i, outputs in chunks of eight, head 0 first;
    (iii) dist_head_dgrad_kernel (learner.hip): workgroup (sample, 128-column block), threads (r, c): rows n = r, r + 2, ... in
        chunks of eight + a tail, the two row classes added as (even) + (odd): every row of the action's block exactly once;
    (iv) HeadWgradRole with the action array (oneshot.h): the matching samples found 64 at a time as a bit mask, taken eight per
        round in ascending order -- the dense sum over the batch with the other samples' (zero) terms left out;
    (v) linear_pair_bwd_kernel: workgroups [0, B) = input rows, [B, B + 2 (O0 + O1)) = (output row, half of K)."""
    import numpy as np
    rs = np.random.RandomState(4)
    # (i)
    for K in (3136, 1024, 2052, 4096, 516):
        nv = K // 4
        nvq = (nv + 3) // 4
        R = 2 if ((nvq + 63) // 64) <= 2 else 4
        assert R * 64 >= nvq
        seen = np.zeros(nv, dtype=np.int64)
        x, w = rs.standard_normal(K), rs.standard_normal(K)
        parts = []
        for q in range(4):
            v0, v1 = q * nvq, min(nv, q * nvq + nvq)
            lane_sums = np.zeros(64)
            for lane in range(64):
                for i in range(R):
                    v = v0 + lane + 64 * i
                    if v < v1:
                        seen[v] += 1
                        lane_sums[lane] += float(np.dot(w[4 * v:4 * v + 4], x[4 * v:4 * v + 4]))
            parts.append(lane_sums.sum())
        assert (seen == 1).all(), K
        np.testing.assert_allclose((parts[0] + parts[1]) + (parts[2] + parts[3]), float(np.dot(w, x)), rtol=1e-12)
    # (ii)
    for K, O0, O1 in ((512, 4, 1), (512, 18, 1), (64, 3, 2), (17, 1, 1), (400, 9, 9)):
        x = rs.standard_normal(K)
        w = rs.standard_normal((O0 + O1, K))
        got = np.full(O0 + O1, np.nan)
        order = []
        for oc in range(0, O0 + O1, 8):
            for u in range(8):
                o = oc + u
                if o >= O0 + O1:
                    continue
                lane_sums = [sum(x[l + 64 * i] * w[o][l + 64 * i] for i in range(8) if l + 64 * i < K) for l in range(64)]
                got[o] = sum(lane_sums)
                order.append(o)
        assert order == list(range(O0 + O1))
        np.testing.assert_allclose(got, w @ x, rtol=1e-10, atol=1e-12)
    # (iii)
    for N in (51, 200, 8, 1, 17):
        d, w = rs.standard_normal(N), rs.standard_normal((N, 4))
        acc, rows = [np.zeros(4), np.zeros(4)], [[], []]
        for r in (0, 1):
            n = r
            while n + 14 < N:
                for u in range(8):
                    rows[r].append(n + 2 * u)
                    acc[r] += d[n + 2 * u] * w[n + 2 * u]
                n += 16
            while n < N:
                rows[r].append(n)
                acc[r] += d[n] * w[n]
                n += 2
        assert sorted(rows[0] + rows[1]) == list(range(N)) and all(v % 2 == 0 for v in rows[0]) and all(v % 2 for v in rows[1])
        np.testing.assert_allclose(acc[0] + acc[1], d @ w, rtol=1e-10, atol=1e-12)
    # (iv)
    for B, A, group in ((32, 204, 51), (100, 800, 200), (7, 12, 3), (1024, 8, 4)):
        n_act = A // group
        action = rs.randint(0, n_act, size=B)
        dq = np.zeros((B, A))
        for b in range(B):
            dq[b, action[b] * group:(action[b] + 1) * group] = rs.standard_normal(group)
        h4 = rs.standard_normal(B)
        for a in (0, A // 2, A - 1):
            mine = a // group
            acc, taken = 0.0, []
            for b0 in range(0, B, 64):
                mask = [b0 + l for l in range(64) if b0 + l < B and action[b0 + l] == mine]
                while mask:
                    chunk, mask = mask[:8], mask[8:]
                    for b in chunk:
                        taken.append(b)
                        acc += dq[b, a] * h4[b]
            assert taken == [b for b in range(B) if action[b] == mine]
            np.testing.assert_allclose(acc, float(dq[:, a] @ h4), rtol=1e-10, atol=1e-12)
    # (v)
    for B, O0, O1 in ((80, 4, 1), (1, 18, 1), (33, 3, 2)):
        jobs = []
        for bid in range(B + 2 * (O0 + O1)):
            if bid < B:
                jobs.append(("dx", bid))
            else:
                r = bid - B
                row, half = r >> 1, r & 1
                jobs.append(("dw0", row, half) if row < O0 else ("dw1", row - O0, half))
        assert [j for j in jobs if j[0] == "dx"] == [("dx", b) for b in range(B)]
        assert sorted(j for j in jobs if j[0] == "dw0") == [("dw0", o, h) for o in range(O0) for h in (0, 1)]
        assert sorted(j for j in jobs if j[0] == "dw1") == [("dw1", o, h) for o in range(O1) for h in (0, 1)]


def test_scatter_input_gradient_maps():
    """csrc/dgrad_scatter.h (round 6), transcribed lane by lane for conv3 (NS = 1, 2) and conv2 (NS = 1): the weight / gradient
    operand maps (A row m <-> channel 16 cb + (m >> 2) + 4 (m & 3); step j = 4 v + e of lane group kq <-> oc = 16 v + 4 kq + e),
    the image address of every (lane, output register, tap), and the dump area of lanes past the last position.  Checked:
      * the determinism argument -- within ONE tap's read-add-write the 64 lanes x 4 registers of a wave touch distinct LDS words,
        and two different waves of a workgroup never touch the same word at all;
      * dump addresses stay inside [IMG, LDS_FLOATS) and never alias an image word;
      * MFMA semantics over these maps + the col2im reproduce F.conv2d's input gradient in float64 (incl. a last workgroup with
        fewer samples and a last tile with idle lanes)."""
    import numpy as np
    import torch
    import torch.nn.functional as F

    def geom(c, h, oc, kh, s):
        oh = (h - kh) // s + 1
        return dict(C=c, H=h, OC=oc, KH=kh, S=s, OH=oh, P=oh * oh, HW=h * h)

    for (c, h, oc, kh, s), ns, batch in (((64, 9, 64, 3, 1), 2, 3), ((64, 9, 64, 3, 1), 1, 2), ((32, 20, 64, 4, 2), 1, 2)):
        g = geom(c, h, oc, kh, s)
        C, H, OC, KH, S, OH, P, HW = (g[k] for k in ("C", "H", "OC", "KH", "S", "OH", "P", "HW"))
        NCB, KST = C // 16, OC // 4
        NPS = 4 // NCB
        KHN = KH // NPS
        NTAP = KHN * KH
        HWP = HW if HW & 1 else HW + 4          # (DRA_SCAT_HWPAD: float4 rows for the epilogue)
        IMG = ns * C * HWP
        ROWF = KH * OC
        WREG = 16 * (ROWF + 4)
        DUMP = 64 + 12 * HWP + (KH - 1) * (H + 1) + 4
        LDS_FLOATS = (max(IMG + DUMP, 4 * WREG) + 3) & ~3
        assert LDS_FLOATS * 4 <= 160 * 1024
        rs = np.random.RandomState(c + ns)
        w = rs.standard_normal((oc, c, kh, kh))
        dy = rs.standard_normal((batch, oc, OH, OH))
        wt = np.ascontiguousarray(w.transpose(1, 2, 3, 0)).reshape(c * kh * kh, oc)       # [(c,kh,kw)][oc]
        xt = torch.zeros((batch, c, h, h), dtype=torch.float64, requires_grad=True)
        F.conv2d(xt, torch.from_numpy(w), None, stride=s).backward(torch.from_numpy(dy))
        want = xt.grad.numpy()
        got = np.zeros_like(want)
        n_groups = (batch + ns - 1) // ns
        for grp in range(n_groups):
            b0 = grp * ns
            nsmp = min(ns, batch - b0)
            nq = nsmp * P
            lds = np.zeros(LDS_FLOATS)
            owner = np.full(LDS_FLOATS, -1)          # which wave touched an image word
            for wave in range(4):
                cb, ps = wave % NCB, wave // NCB
                lanes = np.arange(64)
                n, kq = lanes & 15, lanes >> 4
                # A operand: a[t][j] of lane (n, kq)
                ch_a = 16 * cb + (n >> 2) + 4 * (n & 3)
                tiles = (nq + 15) // 16
                for tile in range(tiles):
                    q = tile * 16 + n
                    ok = q < nq
                    qc = np.minimum(q, nq - 1)
                    smp, p = qc // P, qc % P
                    ohh, oww = p // OH, p % OH
                    ao = np.where(ok, (smp * C + 16 * cb + kq) * HWP + (ohh * S + ps) * H + oww * S, IMG + lanes)
                    for t in range(NTAP):
                        kh_, kw_ = ps + NPS * (t // KH), t % KH
                        # the 16 x 16 x 4 MFMA chain of this tap: D[row][col] = sum_k A[row][k] B[k][col]; lane (col = n, 4 kq + r rows)
                        acc = np.zeros((64, 4))
                        for j in range(KST):
                            ocs = 16 * (j >> 2) + 4 * np.arange(4) + (j & 3)          # oc of lane group kq = 0..3 in step j
                            A = np.zeros((16, 4))                                       # [row m][k = kq]
                            for m in range(16):
                                chm = 16 * cb + (m >> 2) + 4 * (m & 3)
                                A[m] = wt[(chm * KH + kh_) * KH + kw_, ocs]
                            B = np.zeros((4, 16))                                       # [k = kq][col n]
                            for col in range(16):
                                lane0 = col                                             # lanes (col, kq) share position col
                                if tile * 16 + col < nq:
                                    B[:, col] = dy[b0 + smp[lane0], ocs, ohh[lane0], oww[lane0]]
                            D = A @ B
                            for r in range(4):
                                acc[:, r] += D[4 * kq + r, n]
                        off = NPS * (t // KH) * H + (t % KH)
                        addr = ao[:, None] + 4 * np.arange(4)[None, :] * HWP + off      # [lane][r]
                        assert len(np.unique(addr)) == 256, "one tap's read-add-write: 64 lanes x 4 registers, distinct words"
                        img = addr[ok]
                        assert img.size == 0 or (img.min() >= 0 and img.max() < IMG)
                        dump = addr[~ok]
                        assert dump.size == 0 or (dump.min() >= IMG and dump.max() < LDS_FLOATS)
                        assert np.all((owner[img] == -1) | (owner[img] == wave)), "two waves never share an image word"
                        owner[img] = wave
                        lds[addr] += acc
                        # output register r of lane group kq is channel 16 cb + kq + 4 r = the channel of A row 4 kq + r
                        assert np.array_equal(ch_a[(4 * kq[:, None] + np.arange(4)[None, :])], 16 * cb + kq[:, None] + 4 * np.arange(4)[None, :])
            for si in range(nsmp):
                for ch in range(C):
                    base = (si * C + ch) * HWP
                    got[b0 + si, ch] = lds[base:base + HW].reshape(H, H)
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10)
