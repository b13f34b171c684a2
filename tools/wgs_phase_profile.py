#!/usr/bin/env python3
"""Per-phase cycle breakdown of code_units_list_kernel -- the two-wave window coder that takes the all-but-blank units beside
the pipeline kernel -- on a lone frame (split launch), level-1 units, one column per bit plane (profiling build, s_memtime ticks).
    python tools/wgs_phase_profile.py [w h stages segments]      (needs a GPU)"""
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
    lib.icerx_prof_read_wgs.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    enc = api.Encoder(w, h, 1, st, 0, sg, max_frames=1)
    frame = synth.gray_frame(w, h, 12345, 1)[None]
    enc.encode_host(frame, 2 * w * h)
    out = (C.c_uint64 * (9 * 32))()
    lib.icerx_prof_read_wgs(enc.handle, out, 1)
    enc.encode_host(frame, 2 * w * h)
    lib.icerx_prof_read_wgs(enc.handle, out, 1)
    li = enc.launch_info()
    t = np.array(list(out), dtype=np.float64).reshape(9, 32)
    waves = int(os.environ.get("ICER_HIP_LIST_WAVES", "4"))          # (the instance the launch of a lone frame uses)
    print(f"{w}x{h} st={st} seg={sg}, launch {li}: level-1 units coded by the two-wave window coder, sums over the units of a plane")
    print("(cycles of BOTH waves added up; a unit's wall time is about the sum / 2 / units of the plane)")
    print(" " * 34 + "  ".join(f"{'lsb%d' % p:>9s}" for p in range(9)))
    for k, name in enumerate(NAMES):
        if name is None:
            continue
        print(f"  {name:32s}" + "  ".join(f"{v / 1e3:9.0f}" for v in t[:, k]) + "   kcyc")
    tot = t[:, [k for k, n in enumerate(NAMES) if n]].sum(axis=1)
    print(f"  {'total':32s}" + "  ".join(f"{v / 1e3:9.0f}" for v in tot) + "   kcyc")
    print(f"  {'windows (general path)':32s}" + "  ".join(f"{v / waves:9.0f}" for v in t[:, 11]))
    print(f"  {'blank runs':32s}" + "  ".join(f"{v:9.0f}" for v in t[:, 13]))
    with np.errstate(divide="ignore", invalid="ignore"):
        gen = t[:, [1, 2, 3, 4, 5, 6, 7, 8, 9, 10]].sum(axis=1) + t[:, 0]
        print(f"  {'kcyc per window (incl. waits)':32s}" + "  ".join(f"{v:9.1f}" for v in np.nan_to_num(gen / np.maximum(t[:, 11], 1) / 1e3)))
        print(f"  {'kcyc per blank run':32s}" + "  ".join(f"{v:9.1f}" for v in np.nan_to_num(t[:, 12] / np.maximum(t[:, 13], 1) / 1e3)))


if __name__ == "__main__":
    main()
