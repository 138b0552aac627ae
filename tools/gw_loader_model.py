#!/usr/bin/env python3
"""CPU model of the LDS-DMA loader of csrc/hcq_wgrad_grp.hip (addresses only): for every step of a job, every DMA
instruction and lane it reproduces the kernel's (buffer offset | out-of-range, LDS byte address) and checks
  * every in-range access lies inside its tensor,
  * the staged images hold exactly what the fragment reads expect: dy[row][w0 .. w0+15] and, per tap, the window
    x[row][w0 + floor4(wo) ..] with zeros outside the row / image,
  * two lanes that write the same LDS bytes write the same values.
Run after touching the loader, BEFORE the GPU (an out-of-range access there is a memory fault, not a test failure)."""
import itertools
import sys

import numpy as np

OOB = 0xFFFFFFF0
BIAS = 64


def model(OA, IB, KW, XP, N, H, W, dil, pad, kh_off=0, seed=0):
    XC = XP // 4
    XR = 64 // XC
    DYB = 8 * OA * 64
    XTB = 8 * IB * XP * 4 + (64 if 64 % XC else 0)
    NDYI = DYB // 1024
    XTI = 8 * IB // XR
    KD = (NDYI + KW * XTI) // 4
    assert NDYI % 4 == 0 and XTI % 4 == 0
    rng = np.random.RandomState(seed)
    x = rng.randn(N, 8 * IB, H, W).astype(np.float32)
    dy = rng.randn(N, 8 * OA, H, W).astype(np.float32)
    xf, dyf = x.reshape(-1), dy.reshape(-1)
    rs = H * W * 4
    woff0, wstep, hoff = -pad, dil, kh_off
    al = [(woff0 + t * wstep) & ~3 for t in range(KW)]
    lanes = np.arange(64)
    swz = (lanes & 3) ^ ((lanes >> 4) & 3)                               # piece a lane FETCHES for the piece lane & 3 it writes
    ldy = (lanes >> 2) * rs + swz * 16
    lch = swz if XC == 4 else lanes % XC
    lx = (lanes // XC) * rs + lch * 16
    lx_last = np.where(lanes < XR * XC, lx, OOB)
    nchecked = 0
    for n, h, wi in itertools.product(range(N), range(H), range(W // 16)):
        w0 = wi * 16
        hi_ = h + hoff
        rowok = 0 <= hi_ < H
        dy_base = ((n * 8 * OA) * H + h) * W + w0                      # element index of the descriptor base
        x_base = ((n * 8 * IB) * H + hi_) * W + (w0 - BIAS)
        lds_dy = np.full(DYB // 4, np.nan, np.float32)
        lds_x = np.full(KW * XTB // 4, np.nan, np.float32)
        lo, hi = [], []
        for t in range(KW):
            c0 = w0 + al[t]
            l = (-c0) >> 2 if c0 < 0 else 0
            hh = min((W - 4 - c0) >> 2, XC - 1)
            none = (not rowok) or hh < l
            lo.append((1 << 20) if none else l)
            hi.append(0 if none else hh - l)
        for K in range(KD):
            for wave in range(4):
                Q0 = 4 * K
                if Q0 < NDYI:
                    q = Q0 + wave
                    off = ldy + q * 16 * rs
                    idx = dy_base + off // 4
                    assert idx.min() >= 0 and idx.max() + 3 < dyf.size
                    dst = q * 256 + lanes * 4
                    for ln in range(64):
                        lds_dy[dst[ln]:dst[ln] + 4] = dyf[idx[ln]:idx[ln] + 4]
                else:
                    T = (Q0 - NDYI) // XTI
                    e = Q0 - NDYI - T * XTI + wave
                    v = lx_last if e == XTI - 1 else lx
                    ok = ((lch - lo[T]) & 0xFFFFFFFF) <= hi[T]
                    v = np.where(ok, v, OOB)
                    soff = e * XR * rs + (BIAS + al[T]) * 4
                    dst = (T * XTB + e * XR * XC * 16) // 4 + lanes * 4
                    assert dst.max() + 4 <= (T + 1) * XTB // 4
                    for ln in range(64):
                        if v[ln] == OOB:
                            val = np.zeros(4, np.float32)
                        else:
                            idx = x_base + (int(v[ln]) + soff) // 4
                            assert 0 <= idx and idx + 3 < xf.size, (n, h, wi, K, wave, ln, idx)
                            # the piece must lie inside ONE row of the tensor, at the column the window says
                            row = e * XR + ln // XC
                            col = w0 + al[T] + 4 * int(lch[ln])
                            assert row < 8 * IB and 0 <= col <= W - 4
                            assert idx == ((n * 8 * IB + row) * H + hi_) * W + col
                            val = xf[idx:idx + 4]
                        old = lds_x[dst[ln]:dst[ln] + 4]
                        assert np.all(np.isnan(old)) or np.array_equal(old, val), "two lanes disagree on the same LDS bytes"
                        lds_x[dst[ln]:dst[ln] + 4] = val
        # ---- what the fragment reads expect
        exp_dy = dy[n, :, h, w0:w0 + 16]
        img = lds_dy.reshape(8 * OA, 4, 4)
        R = np.arange(8 * OA)
        unsw = np.stack([img[R, c ^ ((R >> 2) & 3)] for c in range(4)], 1).reshape(8 * OA, 16)    # the readers' view
        assert np.array_equal(unsw, exp_dy)
        for t in range(KW):
            img = lds_x[t * XTB // 4: t * XTB // 4 + 8 * IB * XP].reshape(8 * IB, XP)
            wo = woff0 + t * wstep
            ofs = wo & 3
            cols = w0 + wo + np.arange(16)
            exp = np.zeros((8 * IB, 16), np.float32)
            if rowok:
                m = (cols >= 0) & (cols < W)
                exp[:, m] = x[n, :, hi_, cols[m]].T
            if XC == 4:
                R = np.arange(8 * IB)
                i3 = img.reshape(8 * IB, 4, 4)
                img = np.stack([i3[R, c ^ ((R >> 2) & 3)] for c in range(4)], 1).reshape(8 * IB, 16)
            got = img[:, ofs:ofs + 16]
            assert np.array_equal(got, exp), (n, h, wi, t)
        nchecked += 1
    return nchecked


if __name__ == "__main__":
    cases = [
        (48, 24, 3, 20, 1, 1, 128, 1, 1, 0), (48, 24, 3, 20, 1, 1, 128, 55, 55, 0), (48, 24, 3, 20, 2, 1, 128, 13, 13, 0),
        (48, 24, 3, 20, 1, 1, 128, 2, 2, 0), (48, 24, 3, 20, 1, 1, 128, 34, 34, 0),
        (24, 48, 1, 16, 2, 1, 128, 1, 0, 0),
        (24, 24, 3, 20, 1, 3, 128, 1, 1, -1), (24, 24, 3, 20, 1, 3, 128, 1, 1, 0), (24, 24, 3, 20, 2, 2, 128, 1, 1, 1),
        (48, 48, 1, 20, 1, 1, 128, 1, 1, 0), (48, 48, 1, 20, 1, 1, 128, 1, -1, 0),
    ]
    for c in cases:
        print(c, model(*c), "steps ok", flush=True)
