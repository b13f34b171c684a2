"""Deterministic synthetic frames used by the tests and by bench.py.

This is our own generator, not reference code: a 32-bit LCG
(`s = s*1664525 + 1013904223`, one draw `s >> 8` per pixel in raster order, seed 12345 + frame
index) added to a smooth ramp ("natural-like", mode 1) or used directly (mode 0, stress).
8-bit values are stored widened to uint16, exactly how the reference's own CLI feeds
`icer_compress_image_uint16` (example/src/icer_util.c:163-168).

The colour generator draws three values per pixel (R, G, B order) and converts them with the
integer RGB->YCbCr formulas the reference's example apps use before calling the YUV encoder
(example/inc/color_util.h:27-29, app-side code; restated here only to build inputs).
"""
from __future__ import annotations

import numpy as np

LCG_A = np.uint32(1664525)
LCG_C = np.uint32(1013904223)
DEFAULT_SEED = 12345


def lcg_draws(n: int, seed: int) -> np.ndarray:
    """First `n` outputs (state >> 8) of the LCG, vectorised by jump-ahead (mod 2**32)."""
    with np.errstate(over="ignore"):
        a_pow = np.empty(n + 1, dtype=np.uint32)
        a_pow[0] = 1
        if n:
            a_pow[1:] = LCG_A
            np.cumprod(a_pow, dtype=np.uint32, out=a_pow)      # a^k mod 2^32
        geo = np.cumsum(a_pow[:-1], dtype=np.uint32)          # sum_{j<k+1} a^j mod 2^32
        state = a_pow[1:] * np.uint32(seed & 0xFFFFFFFF) + LCG_C * geo
    return state >> np.uint32(8)


def _ramp(w: int, h: int, base: float, kx: float, ky: float) -> np.ndarray:
    x = np.arange(w, dtype=np.float64) / w - 0.5
    y = np.arange(h, dtype=np.float64) / h - 0.5
    # same association order as the C expression  base + kx*(x/W-0.5) + ky*(y/H-0.5)
    return ((base + kx * x[None, :]) + ky * y[:, None]).astype(np.int64)


def gray_frame(w: int, h: int, seed: int = DEFAULT_SEED, mode: int = 1) -> np.ndarray:
    """(h, w) uint16 frame with 8-bit range content."""
    r = lcg_draws(w * h, seed).reshape(h, w)
    if mode == 0:
        v = (r & np.uint32(255)).astype(np.int64)
    elif mode == 1:
        v = _ramp(w, h, 128.0, 60.0, 50.0) + (r % np.uint32(17)).astype(np.int64) - 8
    elif mode == 2:
        # 12-bit-range stress content for true 16-bit paths (beyond what 9 planes code losslessly)
        v = (r & np.uint32(4095)).astype(np.int64)
        return v.astype(np.uint16)
    else:
        raise ValueError("mode must be 0, 1 or 2")
    return np.clip(v, 0, 255).astype(np.uint16)


def gray_batch(n: int, w: int, h: int, seed: int = DEFAULT_SEED, mode: int = 1) -> np.ndarray:
    return np.stack([gray_frame(w, h, seed + k, mode) for k in range(n)])


def color_frame_yuv(w: int, h: int, seed: int = DEFAULT_SEED):
    """Three (h, w) uint16 planes Y, Cb, Cr built from a synthetic RGB frame."""
    r3 = lcg_draws(3 * w * h, seed).reshape(h, w, 3)
    n = (r3 % np.uint32(17)).astype(np.int64) - 8
    x = np.arange(w, dtype=np.float64) / w - 0.5
    y = np.arange(h, dtype=np.float64) / h - 0.5
    zx, zy = np.zeros_like(x), np.zeros_like(y)
    rr = ((128.0 + 60.0 * x[None, :]) + 50.0 * y[:, None]).astype(np.int64) + n[..., 0]
    gg = ((100.0 + 80.0 * y[:, None]) + zx[None, :]).astype(np.int64) + n[..., 1]
    bb = ((140.0 - 70.0 * x[None, :]) + zy[:, None]).astype(np.int64) + n[..., 2]
    rr, gg, bb = (np.clip(c, 0, 255) for c in (rr, gg, bb))
    clip = lambda v: np.clip(v, 0, 255)
    yy = clip((19595 * rr + 38470 * gg + 7471 * bb) >> 16)
    cb = clip(((36962 * (bb - yy)) >> 16) + 128)
    cr = clip(((46727 * (rr - yy)) >> 16) + 128)
    return yy.astype(np.uint16), cb.astype(np.uint16), cr.astype(np.uint16)


def gray_frame_12bit(w: int, h: int, seed: int = DEFAULT_SEED, mode: int = 1) -> np.ndarray:
    """(h, w) uint16 frame with 12-bit content: the 8-bit frame in the upper bits, four more LCG bits below.  Its wavelet
    coefficients exceed the 9 coded bit planes here and there (category 3, magnitudes >= 512), which 8-bit content never does."""
    low = (lcg_draws(w * h, (seed ^ 0x5A5A5A5A) & 0xFFFFFFFF).reshape(h, w) & np.uint32(15)).astype(np.uint16)
    return (gray_frame(w, h, seed, mode) << 4 | low).astype(np.uint16)


def gray_frame_u8(w: int, h: int, seed: int = 12345, mode: int = 1) -> np.ndarray:
    """Input for the uint8 twins (int8 storage, 7 bit planes): the gray frame >> 2, i.e. 6-bit data."""
    return (gray_frame(w, h, seed, mode) >> 2).astype(np.uint8)


def color_frame_yuv_u8(w: int, h: int, seed: int = 12345):
    """Y, U, V planes of color_frame_yuv >> 2 as uint8."""
    return tuple((p >> 2).astype(np.uint8) for p in color_frame_yuv(w, h, seed))


def gray_frames_torch(n: int, w: int, h: int, seed: int, device, mode: int = 1):
    """gray_batch(n, w, h, seed, mode) built on `device` with torch (bench.py: the large batch configurations would take
    minutes to generate on the host): the same LCG by jump-ahead in wrapping 64-bit integer arithmetic, the same float64
    ramp.  Returns an (n, h, w) int16 tensor (8-bit content, as the uint16 API takes it).  tests/test_synth.py checks it
    against the numpy generator."""
    import torch
    npix = w * h
    m32 = 0xFFFFFFFF
    a_pow = torch.full((npix,), int(LCG_A), dtype=torch.int64, device=device)
    a_pow = torch.cumprod(a_pow, 0) & m32                     # a^k mod 2^32, k = 1..npix (int64 wraps mod 2^64: consistent)
    ones = torch.ones(1, dtype=torch.int64, device=device)
    geo = torch.cumsum(torch.cat([ones, a_pow[:-1]]), 0) & m32  # sum_{j<k} a^j mod 2^32
    x = torch.arange(w, dtype=torch.float64, device=device) / w - 0.5
    y = torch.arange(h, dtype=torch.float64, device=device) / h - 0.5
    ramp = ((128.0 + 60.0 * x[None, :]) + 50.0 * y[:, None]).to(torch.int64)
    out = torch.empty((n, h, w), dtype=torch.int16, device=device)
    for k in range(n):
        state = (a_pow * ((seed + k) & m32) + int(LCG_C) * geo) & m32
        r = (state >> 8).reshape(h, w)
        v = (r & 255) if mode == 0 else ramp + (r % 17) - 8
        out[k] = torch.clamp(v, 0, 255).to(torch.int16)
    return out
