#!/usr/bin/env python
"""CPU emulation of the lane / slot / LDS index maps of deeprl_amd/csrc/oneshot.h.

There is no GPU in the authoring container, so every one-pass kernel's addressing is transcribed
here lane by lane (numpy, one workgroup at a time: staging -> "LDS" array -> per-wave MFMA operand
fetch -> 4-wave fold -> guarded store) and compared with torch's autograd on random inputs.  This
checks the mathematics of the maps (slot <-> reduction index bijections, padding, phase decomposition,
slab layout); the real parity tests run on the GPU (tests/test_gpu_kernels.py).

    python tools/emulate_oneshot.py
"""
import numpy as np
import torch
import torch.nn.functional as F


def mfma_row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def mfma_acc(acc, a, b):
    """acc[32][32] += sum_h a[(i,h)] * b[(j,h)]; a, b indexed [h][lane32]."""
    acc += np.outer(a[0], b[0]) + np.outer(a[1], b[1])


class Geom:
    def __init__(self, C, H, OC, KH, S):
        self.C, self.H, self.OC, self.KH, self.S = C, H, OC, KH, S
        self.OH = (H - KH) // S + 1
        self.P = self.OH * self.OH
        self.KK = KH * KH
        self.K = C * self.KK
        self.HW = H * H


G1 = Geom(4, 84, 32, 8, 4)
G2 = Geom(32, 20, 64, 4, 2)
G3 = Geom(64, 9, 64, 3, 1)


def emu_conv_wgrad(G, ROWS, MTG, RW, CSPAD, dy, x, B):
    """ConvWgradOne<G, ROWS, MTG, RW, CSPAD>::run for every block -> (dw slabs [n_slabs][K][OC], db slabs)."""
    S, KH, OC, C, H, OH = G.S, G.KH, G.OC, G.C, G.H, G.OH
    OWP = (OH + 1) & ~1
    NPAIR = OWP // 2
    NJ, NCHUNK = ROWS * NPAIR, OH // ROWS
    MTILES = G.K // 32
    NGRP, NTL = MTILES // MTG, OC // 32
    TILES = MTG * NTL
    NR = (ROWS - 1) * S + KH
    CS = NR * RW + CSPAD
    NCHMAX = (MTG * 32) // G.KK if (MTG * 32) % G.KK == 0 else (MTG * 32 + G.KK - 2) // G.KK + 1
    NCH = min(NCHMAX, C)
    IMG = NCH * CS + RW
    LDB, NPOS = OC + 1, ROWS * OWP
    n_slabs = B * NCHUNK
    dw = np.full((n_slabs, G.K, OC), np.nan)
    db = np.full((n_slabs, OC), np.nan)
    assert RW >= (OWP - 1) * S + KH
    for bid in range(B * NCHUNK * NGRP):
        grp = bid % NGRP
        r = bid // NGRP
        chunk, bi = r % NCHUNK, r // NCHUNK
        k0 = grp * MTG * 32
        c_lo = k0 // G.KK
        c_hi = min((k0 + MTG * 32 - 1) // G.KK, C - 1)
        nch = c_hi - c_lo + 1
        assert nch <= NCH
        ir0 = chunk * ROWS * S
        img = np.full(IMG, np.nan)
        for e in range(NCH * CS + RW):
            cl, rem = divmod(e, CS)
            rr, cc = divmod(rem, RW)
            inside = cl < nch and rr < NR and cc < H
            img[e] = x[bi, c_lo + cl, ir0 + rr, cc] if inside else 0.0
        dyl = np.full(NPOS * LDB, np.nan)
        for e in range(OC * NPOS):
            oc, pos = divmod(e, NPOS)
            ohl, ow = divmod(pos, OWP)
            dyl[pos * LDB + oc] = dy[bi, oc, chunk * ROWS + ohl, ow] if ow < OH else 0.0
        li = np.arange(32)
        slab = bi * NCHUNK + chunk
        for wave in range(4):
            for t in range((TILES + 3) // 4):
                tile = wave + 4 * t
                if tile >= TILES:
                    continue
                mt, nt = tile // NTL, tile % NTL
                k = k0 + mt * 32 + li
                c, kr = k // G.KK, k % G.KK
                kh, kw = kr // KH, kr % KH
                acc = np.zeros((32, 32))
                for j in range(NJ):
                    ohl, jw = j // NPAIR, j % NPAIR
                    a = np.zeros((2, 32))
                    b = np.zeros((2, 32))
                    for h in range(2):
                        a[h] = img[(c - c_lo) * CS + kh * RW + kw + h * S + ohl * S * RW + 2 * jw * S]
                        b[h] = dyl[h * LDB + nt * 32 + li + (ohl * OWP + 2 * jw) * LDB]
                    mfma_acc(acc, a, b)
                assert not np.isnan(acc).any()
                dw[slab, k0 + mt * 32:k0 + mt * 32 + 32, nt * 32:nt * 32 + 32] = acc
        if grp == 0:
            for oc in range(OC):
                db[slab, oc] = sum(dyl[pos * LDB + oc] for pos in range(NPOS))
    return dw, db


def emu_conv_dgrad_lin(G, PT, dy, wt, xact, B):
    """oneshot_lin.h ConvDgradLin<G, PT>::run for every block: the gradient image stays [OC][P] in "LDS", out-of-image
    (lane, tap) pairs read a block of zeros behind it.  -> dx, and the flat LDS addresses every operand read used."""
    S, KH, OC, C, H, OH, P = G.S, G.KH, G.OC, G.C, G.H, G.OH, G.P
    KP = (KH + S - 1) // S
    NPH, HP = S * S, (H + S - 1) // S
    PP = HP * HP
    TPP = (PP + 31) // 32
    TGP = (TPP + PT - 1) // PT
    OCW, OCH, NT, MT = OC // 4, OC // 8, KP * KP, C // 32
    NSRC = OC * P
    ZERO = ((OCH - 1) * P + 1 + 3) & ~3
    dx = np.full((B, C, H, H), np.nan, dtype=np.float64)
    wtf = wt.reshape(-1)
    li = np.arange(32)
    for bid in range(B * NPH * TGP * MT):
        mt = bid % MT
        r = bid // MT
        grp = r % TGP
        r //= TGP
        phi, bi = r % NPH, r // NPH
        ph, pw = phi // S, phi % S
        c0, p0 = mt * 32, grp * PT * 32
        np_ = min(32 * PT, PP - p0)
        lds = np.full(NSRC + ZERO, np.nan)
        lds[:NSRC] = dy[bi].reshape(-1)             # straight copy, float4 by float4
        lds[NSRC:] = 0.0
        for t in range(PT):
            if 32 * t >= np_:
                continue
            pj = np.minimum(32 * t + li, np_ - 1)
            ih2, iw2 = (p0 + pj) // HP, (p0 + pj) % HP
            acc = np.zeros((32, 32))
            for wave in range(4):
                for tp in range(NT):
                    kh2, kw2 = tp // KP, tp % KP
                    kh, kw = kh2 * S + ph, kw2 * S + pw
                    rr, cc = ih2 - kh2, iw2 - kw2
                    ok = (rr >= 0) & (rr < OH) & (cc >= 0) & (cc < OH)
                    for jj in range(OCH):
                        a = np.zeros((2, 32))
                        b = np.zeros((2, 32))
                        for h in range(2):
                            oc = wave * OCW + h * OCH + jj
                            a[h] = wtf[((c0 + li) * G.KK + kh * KH + kw) * OC + oc]
                            boff = np.where(ok, (wave * OCW + h * OCH) * P + rr * OH + cc, NSRC)
                            addr = boff + jj * P
                            assert (addr >= 0).all() and (addr < NSRC + ZERO).all()
                            b[h] = lds[addr]
                        mfma_acc(acc, a, b)
            ih, iw = ih2 * S + ph, iw2 * S + pw
            for l in range(32):
                if 32 * t + l < np_ and ih[l] < H and iw[l] < H:
                    for m in range(32):
                        c = c0 + m
                        dx[bi, c, ih[l], iw[l]] = acc[m][l] * (1.0 if xact[bi, c, ih[l], iw[l]] > 0 else 0.0)
    return dx


def emu_conv_wgrad_lin(G, MTG, dy, x, B):
    """oneshot_lin.h ConvWgradLin<G, MTG>::run for every block: the input channels of a k-tile group and the sample's gradient
    block stay in "LDS" as they lie in memory (the image from a 16-byte aligned start: `shift`); the reduction runs over the
    output positions in memory order, two per MFMA.  -> (dw slabs [B][K][OC], db slabs [B][OC])."""
    S, KH, OC, C, H, OH, P, HW = G.S, G.KH, G.OC, G.C, G.H, G.OH, G.P, G.HW
    NJ = (P + 1) // 2
    ODD = P & 1
    MTILES = G.K // 32
    NGRP, NTL = MTILES // MTG, OC // 32
    TILES = MTG * NTL
    NCHMAX = (MTG * 32) // G.KK if (MTG * 32) % G.KK == 0 else (MTG * 32 + G.KK - 2) // G.KK + 1
    NCH = min(NCHMAX, C)
    IMGF = (NCH * HW + 3 + 3) & ~3
    NSRC = OC * P
    xf = x.reshape(-1)
    n_x4 = x.size // 4
    dw = np.full((B, G.K, OC), np.nan)
    db = np.full((B, OC), np.nan)
    li = np.arange(32)

    def pos_off(p):
        return (p // OH) * S * H + (p % OH) * S
    for bid in range(B * NGRP):
        grp, bi = bid % NGRP, bid // NGRP
        k0 = grp * MTG * 32
        c_lo = k0 // G.KK
        c_hi = min((k0 + MTG * 32 - 1) // G.KK, C - 1)
        nch = c_hi - c_lo + 1
        assert nch <= NCH
        xstart = (bi * C + c_lo) * HW
        shift = xstart & 3
        nvi = (nch * HW + shift + 3) >> 2
        assert nvi * 4 <= IMGF
        img = np.full(IMGF, np.nan)
        for f in range(nvi):
            g4 = min((xstart >> 2) + f, n_x4 - 1)
            img[4 * f:4 * f + 4] = xf[4 * g4:4 * g4 + 4]
        dyl = np.full(NSRC + 4, np.nan)
        dyl[:NSRC] = dy[bi].reshape(-1)
        dyl[NSRC:] = 0.0
        for wave in range(4):
            for t in range((TILES + 3) // 4):
                tile = wave + 4 * t
                if tile >= TILES:
                    continue
                mt, nt = tile // NTL, tile % NTL
                k = k0 + mt * 32 + li
                c, kr = k // G.KK, k % G.KK
                kh, kw = kr // KH, kr % KH
                abase = shift + (c - c_lo) * HW + kh * H + kw
                acc = np.zeros((32, 32))
                for j in range(NJ):
                    last_odd = ODD and j == NJ - 1
                    wrap = ((2 * j + 1) % OH) == 0
                    a = np.zeros((2, 32))
                    b = np.zeros((2, 32))
                    for h in range(2):
                        if last_odd:
                            aaddr = abase + pos_off(2 * j)
                            b[h] = dyl[NSRC] if h else dyl[(nt * 32 + li) * P + 2 * j]
                        else:
                            aaddr = abase + h * ((S * H - (OH - 1) * S) if wrap else S) + pos_off(2 * j)
                            b[h] = dyl[(nt * 32 + li) * P + h + 2 * j]
                        assert (aaddr >= 0).all() and (aaddr < 4 * nvi).all()
                        a[h] = img[aaddr]
                    mfma_acc(acc, a, b)
                assert not np.isnan(acc).any()
                dw[bi, k0 + mt * 32:k0 + mt * 32 + 32, nt * 32:nt * 32 + 32] = acc
        if grp == 0:
            for oc in range(OC):
                db[bi, oc] = sum(dyl[oc * P + pos] for pos in range(P))
    return dw, db


def emu_lin_dgrad(O, dy, w, xact, B, I):
    """LinDgradOne<O>: dx[b][i] = relu'(xact) * sum_o dy[b][o] w[o][i]."""
    KW, NJ, LDA = O // 4, O // 8, O + 1
    tiles_n = (I + 31) // 32
    dx = np.full((B, I), np.nan)
    li = np.arange(32)
    for bid in range(tiles_n * ((B + 31) // 32)):
        bm, bn = bid // tiles_n, bid % tiles_n
        m0, n0 = bm * 32, bn * 32
        ncol = np.minimum(n0 + li, I - 1)
        lds = np.zeros(32 * LDA)
        for e in range(32 * O):
            row, col = divmod(e, O)
            lds[row * LDA + col] = dy[min(m0 + row, B - 1), col]
        acc = np.zeros((32, 32))
        for wave in range(4):
            for j in range(NJ):
                a = np.zeros((2, 32))
                b = np.zeros((2, 32))
                for h in range(2):
                    kb = wave * KW + h * NJ
                    a[h] = lds[li * LDA + kb + j]
                    b[h] = w[kb + j, ncol]
                mfma_acc(acc, a, b)
        for m in range(32):
            for l in range(32):
                if m0 + m < B and n0 + l < I:
                    dx[m0 + m, n0 + l] = acc[m][l] * (1.0 if xact[m0 + m, n0 + l] > 0 else 0.0)
    return dx


def emu_lin_wgrad_one(NI, dy, x, B, O, I):
    """oneshot_lin.h LinWgradOne<NI>::run for every block: both operands straight from memory to registers (lane <-> o / i),
    the reduction over the batch rows two per MFMA.  -> (dw [O][I], db [O])."""
    tiles_i = (I + 31) // 32
    tiles_o, groups_i = O // 32, (tiles_i + NI - 1) // NI
    dw = np.full((O, I), np.nan)
    db = np.full(O, np.nan)
    li = np.arange(32)
    for bid in range(tiles_o * groups_i):
        gi, to = bid % groups_i, bid // groups_i
        o0 = to * 32
        for wave in range(4):
            for t in range((NI + 3) // 4):
                it = wave + 4 * t
                i0 = (gi * NI + it) * 32
                if not (it < NI and i0 < I):
                    continue
                acc = np.zeros((32, 32))
                for j in range(16):
                    a = np.zeros((2, 32))
                    b = np.zeros((2, 32))
                    for h in range(2):
                        row = 2 * j + h
                        a[h] = dy[min(row, B - 1), np.minimum(o0 + li, O - 1)] if row < B else 0.0
                        b[h] = x[min(row, B - 1), np.minimum(i0 + li, I - 1)]
                    mfma_acc(acc, a, b)
                for m in range(32):
                    for n in range(32):
                        if o0 + m < O and i0 + n < I:
                            dw[o0 + m, i0 + n] = acc[m][n]
        if gi == 0:
            for m in range(32):
                if o0 + m < O:
                    db[o0 + m] = sum(dy[b, o0 + m] for b in range(B))
    return dw, db


def emu_lin_fwd_slabs(I, KS, x, w, B, O):
    KPS = I // KS
    KW, NJ, LD = KPS // 4, KPS // 8, KPS + 1
    tiles_n, tiles_m = (O + 31) // 32, (B + 31) // 32
    slabs = np.full((KS, B, O), np.nan)
    li = np.arange(32)
    for bid in range(tiles_n * tiles_m * KS):
        bn = bid % tiles_n
        r = bid // tiles_n
        bm = r % tiles_m
        s = (r // tiles_m) % KS
        m0, n0, k0 = bm * 32, bn * 32, s * KPS
        xs, ws = np.zeros(32 * LD), np.zeros(32 * LD)
        for row in range(32):
            xs[row * LD:row * LD + KPS] = x[min(m0 + row, B - 1), k0:k0 + KPS]
            ws[row * LD:row * LD + KPS] = w[min(n0 + row, O - 1), k0:k0 + KPS]
        acc = np.zeros((32, 32))
        for wave in range(4):
            for j in range(NJ):
                a = np.zeros((2, 32))
                b = np.zeros((2, 32))
                for h in range(2):
                    a[h] = xs[li * LD + wave * KW + h * NJ + j]
                    b[h] = ws[li * LD + wave * KW + h * NJ + j]
                mfma_acc(acc, a, b)
        for m in range(32):
            for l in range(32):
                if m0 + m < B and n0 + l < O:
                    slabs[s, m0 + m, n0 + l] = acc[m][l]
    return slabs


def emu_conv_fwd(G, PT, x, wt, U8=False):
    """conv_fwd_v2_kernel / conv_fwd_v2_persist_kernel (conv_v2.hip): tile groups of PT 32-position tiles, the
    division-free staging map (LR lanes per image row, 64/LR rows per pass, stride-phase de-interleaved LDS
    columns), the 4-wave K split and the per-MFMA operand addresses `bptr + off`.  x [B][C][H][H], wt [K][OC]."""
    B = x.shape[0]
    S, KH, OH, P, KK, C, H, OC = G.S, G.KH, G.OH, G.P, G.KK, G.C, G.H, G.OC
    TPS = (P + 31) // 32
    WPH = (H + S - 1) // S
    RW = S * WPH
    CP = C // 2
    CPW = CP // 4 if CP >= 4 else 1
    TSPLIT = 1 if CP >= 4 else 4 // CP
    TW = KK // TSPLIT
    NJ = CPW * TW
    TPG = (TPS + PT - 1) // PT
    OROWS = (32 * PT - 1 + OH - 1) // OH + 1
    NR = min((OROWS - 1) * S + KH, H)
    CS = NR * RW
    LR = 32 if (U8 or H > 16) else 16
    RP = 64 // LR
    LPT = (NR + RP - 1) // RP
    CPT = (C + 3) // 4
    COLS = H // 4 if U8 else H
    lds_col = lambda iw: (iw % S) * WPH + iw // S
    y = np.zeros((B, OC, P))
    for bi in range(B):
        for grp in range(TPG):
            p0 = grp * PT * 32
            npos = min(32 * PT, P - p0)
            if npos <= 0:
                continue
            oh0, oh1 = p0 // OH, (p0 + npos - 1) // OH
            ir0, nrows = oh0 * S, (oh1 - oh0) * S + KH
            assert nrows <= NR
            lds = np.full(C * CS, np.nan)
            for wave in range(4):
                for ci in range(CPT):
                    c = wave + 4 * ci
                    for lane in range(64):
                        rsub, cl = lane // LR, lane % LR
                        for q in range(LPT):
                            r = RP * q + rsub
                            if cl < COLS and r < nrows and c < C:
                                if U8:   # one u32 word = 4 pixels: pixel b of word cl is stride phase b, index cl
                                    for b in range(4):
                                        lds[c * CS + r * RW + cl + b * WPH] = x[bi, c, ir0 + r, 4 * cl + b]
                                else:
                                    lds[c * CS + r * RW + lds_col(cl)] = x[bi, c, ir0 + r, cl]
            for oc0 in range(0, OC, 32):
                acc = np.zeros((PT, 32, 32))
                for wave in range(4):
                    cp0 = wave * CPW if CP >= 4 else wave % CP
                    t0 = 0 if CP >= 4 else (wave // CP) * TW
                    for j in range(NJ):
                        cpl, tp = j // TW, j % TW
                        kh, kw = tp // KH, tp % KH
                        off = 2 * cpl * CS + kh * RW + (kw % S) * WPH + kw // S
                        a = np.stack([wt[((2 * cp0 + h) * KK + t0) + (2 * cpl * KK + tp), oc0:oc0 + 32] for h in range(2)])
                        for t in range(PT):
                            b = np.empty((2, 32))
                            for h in range(2):
                                for li in range(32):
                                    pj = min(32 * t + li, npos - 1)
                                    poh, pw = (p0 + pj) // OH, (p0 + pj) % OH
                                    bptr = (2 * cp0 + h) * CS + ((poh - oh0) * S + t0 // KH) * RW + pw
                                    b[h, li] = lds[bptr + off]
                            mfma_acc(acc[t], a, b)
                for t in range(PT):
                    for li in range(32):
                        if 32 * t + li < npos:
                            y[bi, oc0:oc0 + 32, p0 + 32 * t + li] = acc[t][:, li]
    return y.reshape(B, OC, OH, OH)


def check(name, got, want, tol=1e-9):
    assert not np.isnan(got).any(), name + ": unwritten outputs"
    err = np.abs(got - want).max() / max(1e-30, np.abs(want).max())
    print("%-28s max rel err %.2e %s" % (name, err, "ok" if err < tol else "FAIL"))
    FAILED.extend([name] if not err < tol else [])
    assert err < tol, name


FAILED = []


def main():
    torch.manual_seed(0)
    rs = np.random.RandomState(0)
    for name, G, wg in (("conv3", G3, (7, 3, 10, 1)), ("conv2", G2, (9, 4, 24, 4)), ("conv1", G1, (4, 4, 88, 0))):
        B = 2
        x = torch.randn(B, G.C, G.H, G.H, dtype=torch.float64)
        x = x * (x > -0.3)  # some exact zeros / negatives for the relu mask
        w = torch.randn(G.OC, G.C, G.KH, G.KH, dtype=torch.float64, requires_grad=True)
        xr = x.clone().requires_grad_(True)
        y = F.conv2d(xr, w, stride=G.S)
        dy = torch.randn_like(y)
        y.backward(dy)
        wt = w.detach().permute(1, 2, 3, 0).reshape(G.K, G.OC).numpy()
        want_dw = w.grad.permute(1, 2, 3, 0).reshape(G.K, G.OC).numpy()
        if name != "conv1":       # oneshot_lin.h: operands in LDS as they lie in memory, geometry in the operand base addresses
            want_dx = xr.grad.numpy() * (x.numpy() > 0)
            pt_lin = 2 if name == "conv2" else 1
            check(name + " dgrad lin", emu_conv_dgrad_lin(G, pt_lin, dy.numpy(), wt, x.numpy(), B), want_dx)
            dwl, dbl = emu_conv_wgrad_lin(G, 4 if name == "conv2" else 2, dy.numpy(), x.numpy(), B)
            check(name + " wgrad lin", dwl.sum(0), want_dw)
            check(name + " bias  lin", dbl.sum(0), dy.sum((0, 2, 3)).numpy())
        else:                     # oneshot.h ConvWgradOne: conv1 (transposing staging, 4-row chunks)
            dw, db = emu_conv_wgrad(G, *wg, dy.numpy(), x.numpy(), B)
            check(name + " wgrad one-pass", dw.sum(0), want_dw)
            check(name + " bias  one-pass", db.sum(0), dy.sum((0, 2, 3)).numpy())
    B, I, O = 5, 96, 64
    dyl, wl = rs.randn(B, O), rs.randn(O, I)
    xa = rs.randn(B, I)
    check("linear dgrad one-pass", emu_lin_dgrad(O, dyl, wl, xa, B, I), (dyl @ wl) * (xa > 0))
    for B, I, O in ((32, 200, 64), (5, 96, 32), (1, 40, 64)):      # fc4's weight gradient at batch <= 32, ragged input tiles
        dyl, xl = rs.randn(B, O), rs.randn(B, I)
        dwl, dbl = emu_lin_wgrad_one(8, dyl, xl, B, O, I)
        check("linear wgrad register-only B=%d" % B, dwl, dyl.T @ xl)
        check("linear bias  register-only B=%d" % B, dbl, dyl.sum(0))
    B, I, O, KS = 3, 64, 40, 2
    xl, wl = rs.randn(B, I), rs.randn(O, I)
    check("linear fwd slabs one-pass", emu_lin_fwd_slabs(I, KS, xl, wl, B, O).sum(0), xl @ wl.T)
    # the instantiations the learner launches: fc4 forward with 8 / 14 / 28 K slices (3136 inputs) and the distributional
    # heads' one-pass contraction (512 inputs, the whole reduction in one workgroup)
    for I, KS in ((3136, 8), (3136, 14), (3136, 28), (512, 1)):
        B, O = 3, 40
        xl, wl = rs.randn(B, I), rs.randn(O, I)
        check("linear fwd slabs one-pass I=%d KS=%d" % (I, KS), emu_lin_fwd_slabs(I, KS, xl, wl, B, O).sum(0), xl @ wl.T)
    for name, G, pts, u8 in (("conv1", G1, (1, 2), True), ("conv2", G2, (1, 3), False), ("conv3", G3, (1, 2), False)):
        x = rs.randint(0, 256, size=(2, G.C, G.H, G.H)).astype(np.float64) if u8 else rs.randn(2, G.C, G.H, G.H)
        w = rs.randn(G.OC, G.C, G.KH, G.KH)
        want = F.conv2d(torch.tensor(x), torch.tensor(w), stride=G.S).numpy()
        wt = np.transpose(w, (1, 2, 3, 0)).reshape(G.K, G.OC)
        for pt in pts:
            check("%s fwd PT=%d staging+operands" % (name, pt), emu_conv_fwd(G, pt, x, wt, U8=u8), want)
    if FAILED:
        raise AssertionError("index maps disagree with autograd: %s" % FAILED)
    print("all index maps agree with autograd")


if __name__ == "__main__":
    main()


# ---------------------------------------------------------------------------------------------------------------
# fold_norm_kernel (optim.hip): slab fold + sums of squares, transcribed workgroup by workgroup / thread by thread.
# Returns (folded gradient, partials[blocks], owners[float4 index] = how many (block, thread) pairs own that element).
def emu_clip_step(grad, segs, plain_iters_cap=2048):
    """grad: f32 [n]; segs: list of (begin, count, slabs [ns, stride], stride, ns) with begin / count multiples of 4."""
    n = grad.size
    n4_total = n // 4
    g4 = grad.astype(np.float32).reshape(n4_total, 4).copy()
    owners = np.zeros(n4_total, dtype=np.int64)
    first_block, blocks = [], 0
    for (begin, count, slabs, stride, ns) in segs:
        first_block.append(blocks)
        per = 64 if ns <= 32 else 16
        blocks += (count // 4 + per - 1) // per
    first_block.append(blocks)
    end = segs[-1][0] + segs[-1][1] if segs else 0
    plain_begin4, plain_count4 = end // 4, (n - end) // 4
    pb = (plain_count4 + 1023) // 1024
    room = plain_iters_cap - blocks
    iters = 1
    if pb > room:
        iters = (pb + room - 1) // room
        pb = (pb + iters - 1) // iters
    total = blocks + pb
    partials = np.zeros(total, dtype=np.float64)
    f32 = np.float32
    for bid in range(total):
        acc = np.zeros(256, dtype=np.float32)                 # per-thread fp32 accumulators
        if bid < first_block[len(segs)]:
            sg = 0
            while sg + 1 < len(segs) and bid >= first_block[sg + 1]:
                sg += 1
            begin, count, slabs, stride, ns = segs[sg]
            b = bid - first_block[sg]
            n4 = count // 4
            sl4 = slabs.astype(np.float32).reshape(ns, stride // 4, 4)
            begin4 = begin // 4
            if ns <= 32:
                e0 = b * 64
                s_part = np.zeros((4, 16, 16, 4), dtype=np.float32)
                for tid in range(256):
                    g, el = tid >> 4, tid & 15
                    v0, v1 = g < ns, g + 16 < ns
                    for u in range(4):
                        i = e0 + 16 * u + el
                        ic = i if i < n4 else n4 - 1
                        pp = np.zeros(4, dtype=np.float32)
                        if v0:
                            pp = pp + sl4[g, ic]
                        if v1:
                            pp = pp + sl4[g + 16, ic]
                        s_part[u, g, el] = pp
                for tid in range(64):
                    u, el = tid >> 4, tid & 15
                    r = s_part[u, 0, el].copy()
                    for q in range(1, 16):
                        r = r + s_part[u, q, el]
                    io = e0 + tid
                    if io < n4:
                        g4[begin4 + io] = r
                        acc[tid] = f32(acc[tid] + f32(f32(f32(r[0] * r[0]) + f32(r[1] * r[1])) + f32(r[2] * r[2])) + f32(r[3] * r[3]))
                        owners[begin4 + io] += 1
            else:
                s_part = np.zeros((16, 16, 4), dtype=np.float32)
                for tid in range(256):
                    g, el = tid >> 4, tid & 15
                    i = b * 16 + el
                    ic = i if i < n4 else n4 - 1
                    pp = np.zeros(4, dtype=np.float32)
                    s0 = g
                    while s0 < ns:
                        for u in range(10):
                            s_ = s0 + 16 * u
                            if s_ < ns:
                                pp = pp + sl4[s_, ic]
                        s0 += 160
                    s_part[g, el] = pp
                for tid in range(16):
                    el = tid
                    r = s_part[0, el].copy()
                    for q in range(1, 16):
                        r = r + s_part[q, el]
                    i = b * 16 + el
                    if i < n4:
                        g4[begin4 + i] = r
                        acc[tid] = f32(acc[tid] + f32(f32(f32(r[0] * r[0]) + f32(r[1] * r[1])) + f32(r[2] * r[2])) + f32(r[3] * r[3]))
                        owners[begin4 + i] += 1
        else:
            b = bid - first_block[len(segs)]
            for it in range(iters):
                for tid in range(256):
                    i0 = (it * pb + b) * 1024 + tid
                    for v in range(4):
                        i = i0 + 256 * v
                        if i < plain_count4:
                            r = g4[plain_begin4 + i]
                            acc[tid] = f32(acc[tid] + f32(f32(f32(r[0] * r[0]) + f32(r[1] * r[1])) + f32(r[2] * r[2])) + f32(r[3] * r[3]))
                            owners[plain_begin4 + i] += 1
        partials[bid] = float(acc.astype(np.float64).sum())
    return g4.reshape(-1), partials, owners
