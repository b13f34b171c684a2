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

NAMES = ["pixel wave: context+loads", "pixel wave: wait (queue full)",
         "count wave: wait (pixels)", "count wave: counts", "count wave: fold+bin", "count wave: wait (queue full)",
         "walk wave: wait (events/verdict)", "walk wave: walks (two halves)", "walk wave: join + results", "walk wave: (unused)",
         "golomb wave: wait (events/verdict)", "golomb wave: bins 0, 8-16", "golomb wave: hand-over",
         "merge wave: wait (events)", "merge wave: walker reads + ring slots", "merge wave: drain", "merge wave: exact path",
         "merge wave: retire", "merge wave: event reads + wait golomb", "merge wave: golomb reads + wait walker",
         "records wave: idle / poll", "records wave: v2v records", "drain wave: pop + pack", "drain wave: store payload",
         "drain wave: idle / poll", "compaction wave: wait (events)", "compaction wave: ranks + bit strings"]
NT = 32


def trace_report(tr, w, h, st, sg):
    """start / end of every workgroup of the frame (100 MHz clock), where it ran, and who shared its CU"""
    t0 = int(tr[:, 0].min())
    start = (tr[:, 0].astype(np.int64) - t0) / 1e5          # ms
    end = (tr[:, 1].astype(np.int64) - t0) / 1e5
    hw = tr[:, 2]
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 15
    cu = (hw.astype(np.int64) >> 8) & 15
    sh = (hw.astype(np.int64) >> 12) & 1
    se = (hw.astype(np.int64) >> 13) & 7
    place = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print(f"workgroup trace: {len(tr)} workgroups, last end {end.max():.2f} ms, distinct CUs used {len(set(place.tolist()))}")
    nbig = 27 * sg
    order = np.argsort(-end)[:12]
    print("  latest finishers: launch position, start ms, end ms, xcc/se/sh/cu, big units on the same CU")
    for b in order:
        mates = [int(x) for x in np.nonzero((place == place[b]) & (np.arange(len(tr)) < nbig))[0] if x != b]
        print(f"   pos {int(b):5d} unit {int(tr[b, 3]):5d}  {start[b]:6.2f} -> {end[b]:6.2f}   {int(xcc[b])}/{int(se[b])}/{int(sh[b])}/{int(cu[b])}   {mates}")
    big_end = end[:nbig]
    cnt = np.array([np.sum(place[:nbig] == place[b]) for b in range(nbig)])
    for k in sorted(set(cnt.tolist())):
        print(f"  big units sharing their CU with {k - 1} other big unit(s): {np.sum(cnt == k)}, mean end {big_end[cnt == k].mean():.2f} ms, max {big_end[cnt == k].max():.2f} ms")
    hist, edges = np.histogram(end, bins=10, range=(0, end.max()))
    print("  end-time histogram (ms): " + "  ".join(f"<{edges[i + 1]:.1f}:{hist[i]}" for i in range(10)))


def main():
    w, h, st, sg = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (4096, 4096, 5, 10)))
    api.LIB_PATH = build.build_profiling_library()
    lib = api.load_library()
    lib.icerx_prof_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    enc = api.Encoder(w, h, 1, st, 0, sg, max_frames=1)
    frame = synth.gray_frame(w, h, 12345, int(os.environ.get("MODE", "1")))[None]
    enc.encode_host(frame, 2 * w * h)
    out = (C.c_uint64 * (9 * NT))()
    lib.icerx_prof_read(enc.handle, out, 1)
    enc.timing_enable(True)
    enc.encode_host(frame, 2 * w * h)
    ms, _ = enc.timing_read()
    ntr = min(enc.info()["units_per_frame"], 4096)
    tr = (C.c_uint64 * (4 * ntr))()
    lib.icerx_prof_trace.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    lib.icerx_prof_trace(enc.handle, tr, ntr)
    lib.icerx_prof_read(enc.handle, out, 1)
    trace_report(np.array(tr, dtype=np.uint64).reshape(ntr, 4), w, h, st, sg)
    nunits = 3 * sg                                     # level-1 units per plane (HL, LH, HH x segments)
    chunks = (w // 2) * (h // 2) * 3 / 64.0             # 64-pixel chunks per plane over those units
    print(f"{w}x{h} st={st} seg={sg}: code_units {ms['code_units']:.2f} ms; level-1 units only, per bit plane:")
    print("   cycles per chunk and wave (kcyc)  " + "".join(f"  lsb{p}" for p in range(9)))
    for k, n in enumerate(NAMES):
        print(f"  {n:40s}" + "".join(f"{out[p * NT + k] / chunks / 1e3:6.2f}" for p in range(9)))
    print(f"  {'exact-path chunks %':40s}" + "".join(f"{100.0 * out[p * NT + 29] / max(out[p * NT + 31], 1):6.2f}" for p in range(9)))
    print(f"  {'doubtful chunks % (quick test failed)':40s}" + "".join(f"{100.0 * out[p * NT + 30] / max(out[p * NT + 31], 1):6.2f}" for p in range(9)))
    tot = [sum(out[p * NT + k] for k in range(27)) for p in range(9)]
    print(f"  {'per-wave total (= unit latency / chunk)':40s}" + "".join(f"{t / 8 / chunks / 1e3:6.2f}" for t in tot))


if __name__ == "__main__":
    main()
