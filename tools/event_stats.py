#!/usr/bin/env python3
"""Events of the level-1 coding units of a synthetic frame by coder bin and bit plane (CPU build of the kernel source, tests/emu).
   python tools/event_stats.py [W H STAGES]      -- what the pipeline's waves see: how many events per chunk go to the uncoded bin 0,
   the variable-to-variable bins 1..7 and the Golomb bins 8..16."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icer_compression_amd import synth  # noqa: E402


def main():
    w, h, st = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (1024, 1024, 4)))
    L = C.CDLL(os.path.join(ROOT, "tests", "emu", "libcoder_emu.so"))
    u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
    L.emu_dwt.argtypes = [u16p, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
    L.emu_unit_bin_hist.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    img = synth.gray_frame(w, h, synth.DEFAULT_SEED, 1).copy()
    assert L.emu_dwt(img, w, h, st, 0) == 0
    v = img.view(np.int16).astype(np.int32)
    sm = np.where(v < 0, 0x8000 | (-v), v).astype(np.uint16)       # sign-magnitude words of the detail bands
    hw, hh = w // 2, h // 2
    for name, sb, (y0, x0) in (("HL", 1, (0, hw)), ("HH", 3, (hh, hw))):
        seg = np.ascontiguousarray(sm[y0:y0 + hh, x0:x0 + hw])
        print(f"level-1 {name} {hw}x{hh}: per chunk of 64 pixels")
        print("  lsb  events  bin0  v2v(1-7)  golomb(8-16)  signs  blank%  no-v2v%  sign-in-golomb%")
        for lsb in range(9):
            hist = (C.c_ulonglong * 40)()
            L.emu_unit_bin_hist(seg.ctypes.data, hw, hh, hw, sb, lsb, hist)
            n = hist[34]
            mag = [hist[b] for b in range(17)]
            sg = [hist[17 + b] for b in range(17)]
            tot = sum(mag) + sum(sg)
            b0 = mag[0] + sg[0]
            v2v = sum(mag[1:8]) + sum(sg[1:8])
            gol = sum(mag[8:]) + sum(sg[8:])
            print(f"  {lsb:3d}  {tot / n:6.1f} {b0 / n:5.1f}  {v2v / n:8.1f}  {gol / n:12.1f}  {sum(sg) / n:5.1f}  {100 * hist[35] / n:6.1f}  {100 * hist[36] / n:7.1f}  {100 * hist[37] / n:7.1f}")


if __name__ == "__main__":
    main()
