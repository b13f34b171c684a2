#!/usr/bin/env python3
"""Per-phase cycle breakdown of code_units_wg_kernel (profiling build, s_memtime ticks per wave; level-1 units only,
one row per bit plane).   python tools/wg_phase_profile.py [w h stages segments]      (needs a GPU)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icer_compression_amd import api, build, synth  # noqa: E402

NAMES = ["wait at barriers", "A pixels -> events", "B counts + bins", "C per-bin summaries, walks", "D+E states, records",
         "flush test", "F slots + commit", "drain 1: lengths", "drain 2: pack", "drain 3: store", "exact chunk", None, "blank runs (closed form)"]


def main():
    w, h, st, sg = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (4096, 4096, 5, 10)))
    api.LIB_PATH = build.build_profiling_library()
    lib = api.load_library()
    lib.icerx_prof_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    enc = api.Encoder(w, h, 1, st, 0, sg, max_frames=1)
    frame = synth.gray_frame(w, h, 12345, int(os.environ.get("MODE", "1")))[None]
    enc.encode_host(frame, 2 * w * h)
    NT = 32
    out = (C.c_uint64 * (9 * NT))()
    lib.icerx_prof_read(enc.handle, out, 1)
    t = np.array(list(out), dtype=np.float64).reshape(9, NT)
    waves = int(os.environ.get("ICER_WG_WAVES", "16"))
    print(f"{w}x{h} st={st} seg={sg}: level-1 units, cycles per window and wave (kcyc), by bit plane; {waves} waves per workgroup")
    print(" " * 34 + "  ".join(f"lsb{p}" for p in range(9)))
    tot = np.zeros(9)
    for k, name in enumerate(NAMES):
        if name is None:
            continue
        row = t[:, k] / np.maximum(t[:, 11], 1) / 1e3         # (each wave adds its window count to bucket 11)
        tot += row
        print(f"  {name:32s}" + "  ".join(f"{v:4.1f}" for v in row))
    print(f"  {'total per window':32s}" + "  ".join(f"{v:4.1f}" for v in tot))
    print(f"  {'per chunk':32s}" + "  ".join(f"{v / waves:4.2f}" for v in tot))
    # totals of wave 0's clock per bit plane (all level-1 units of the plane together), for planes that are mostly blank runs
    print("sum over the level-1 units of a plane, kcyc per wave (bucket / waves):")
    for k, name in enumerate(NAMES):
        if name is None:
            continue
        print(f"  {name:32s}" + "  ".join(f"{v / waves / 1e3:7.0f}" for v in t[:, k]))
    print(f"  {'windows (general path)':32s}" + "  ".join(f"{v / waves:7.0f}" for v in t[:, 11]))


if __name__ == "__main__":
    main()
