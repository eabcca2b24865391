"""CPU restatement of the Atari frame preprocessing the reference gets from baselines / OpenCV (TEST INFRASTRUCTURE: only
tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this).

PARITY UNPINNED BY THE REFERENCE: deep_rl/component/envs.py:39-47 calls baselines.common.atari_wrappers.wrap_deepmind
(baselines @8e56dd per SURVEY.md; absent from /root/reference) whose WarpFrame is
    frame = cv2.cvtColor(frame, cv2.COLOR_RGB2GRAY); frame = cv2.resize(frame, (84, 84), interpolation=cv2.INTER_AREA)
on the MaxAndSkipEnv observation (elementwise max of the last two raw frames).  Neither package is installed, so the
functions below restate OpenCV's published algorithms (imgproc/color_rgb: fixed-point luminance; imgproc/resize.cpp:
computeResizeAreaTab + resizeArea_ for a non-integer scale) and are checked against an independent float64 area average
(tests/test_oracle_vs_golden.py), not against cv2 itself."""
import math

import numpy as np


def rgb2gray_u8(rgb):
    """cv2.COLOR_RGB2GRAY for uint8: (R*4899 + G*9617 + B*1868 + (1 << 13)) >> 14."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def resize_area_tab(ssize, dsize):
    """OpenCV computeResizeAreaTab (cn = 1): per destination index the list of (source index, float32 weight)."""
    scale = ssize / dsize
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        row = []
        if sx1 - fsx1 > 1e-3:
            row.append((sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            row.append((sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            row.append((sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
        tab.append(row)
    return tab


def resize_area_u8(gray, out_h=84, out_w=84):
    """cv2.resize(gray, (out_w, out_h), interpolation=cv2.INTER_AREA) for a uint8 image and a non-integer scale (resizeArea_):
    fp32 throughout, horizontal pass per source row in table order, vertical accumulation sum = beta*buf / sum += beta*buf,
    saturate_cast<uchar> (round half to even)."""
    h, w = gray.shape
    xt, yt = resize_area_tab(w, out_w), resize_area_tab(h, out_h)
    g = gray.astype(np.float32)
    out = np.zeros((out_h, out_w), dtype=np.uint8)
    for dy in range(out_h):
        total = np.zeros(out_w, dtype=np.float32)
        for k, (sy, beta) in enumerate(yt[dy]):
            buf = np.zeros(out_w, dtype=np.float32)
            for dx in range(out_w):
                acc = np.float32(0.0)
                for sx, alpha in xt[dx]:
                    acc = np.float32(acc + np.float32(g[sy, sx] * alpha))
                buf[dx] = acc
            total = (beta * buf).astype(np.float32) if k == 0 else (total + (beta * buf).astype(np.float32)).astype(np.float32)
        out[dy] = np.clip(np.rint(total), 0, 255).astype(np.uint8)
    return out


def atari_preprocess(raw2):
    """raw2 [n_env][2][H][W][3] uint8 -> [n_env][84][84] uint8: max of the two frames, luminance, INTER_AREA resize."""
    raw2 = np.asarray(raw2, dtype=np.uint8)
    mx = np.maximum(raw2[:, 0], raw2[:, 1])
    return np.stack([resize_area_u8(rgb2gray_u8(mx[i])) for i in range(mx.shape[0])])
