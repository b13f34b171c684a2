#!/usr/bin/env python3
"""Per-phase cycle breakdown of code_units_kernel (profiling build with s_memtime counters).
   python tools/phase_profile.py [w h stages segments]      (needs a GPU)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icer_compression_amd import api, build, synth  # noqa: E402

NAMES = ["ctx wave: context+loads", "ctx wave: counts", "ctx wave: fold+bin", "ctx wave: wait (queue full)",
         "walk wave: wait (events/verdict)", "walk wave: compaction", "walk wave: walk",
         "asm wave: wait (events)", "asm wave: bin0+golomb+wait walker+slots", "asm wave: drain", "asm wave: exact path",
         "asm wave: stage flush"]


def main():
    w, h, st, sg = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (4096, 4096, 5, 10)))
    api.LIB_PATH = build.build_profiling_library()
    lib = api.load_library()
    lib.icerx_prof_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    enc = api.Encoder(w, h, 1, st, 0, sg, max_frames=1)
    frame = synth.gray_frame(w, h, 12345, int(os.environ.get("MODE", "1")))[None]
    enc.encode_host(frame, 2 * w * h)
    out = (C.c_uint64 * 12)()
    lib.icerx_prof_read(enc.handle, out, 1)
    enc.timing_enable(True)
    enc.encode_host(frame, 2 * w * h)
    ms, _ = enc.timing_read()
    lib.icerx_prof_read(enc.handle, out, 1)
    tot = sum(out)
    print(f"{w}x{h} st={st} seg={sg}: code_units {ms['code_units']:.2f} ms; summed wave-cycles {tot/1e6:.1f} M")
    for n, v in zip(NAMES, out):
        print(f"  {n:22s} {v/1e6:10.1f} Mcyc  {100.0*v/max(tot,1):5.1f} %")


if __name__ == "__main__":
    main()
