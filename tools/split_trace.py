#!/usr/bin/env python3
"""Workgroup trace of a single-frame launch with sub-range splitting (profiling build): when every workgroup of
code_units_kernel started and ended, by kind (sub-range workgroup / whole unit) -- where the launch's time goes.
   python tools/split_trace.py [w h stages segments]      (needs a GPU; ICER_HIP_SPLIT etc. as usual)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icer_compression_amd import api, build, synth  # noqa: E402


def main():
    w, h, st, sg = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (4096, 4096, 5, 10)))
    api.LIB_PATH = os.environ.get("ICER_HIP_PROF_LIB") or build.build_profiling_library()       # (a profiling build of another source tree: experiments)
    lib = api.load_library()
    enc = api.Encoder(w, h, 1, st, 0, sg, max_frames=1)
    frame = synth.gray_frame(w, h, 12345, 1)[None]
    enc.encode_host(frame, 2 * w * h)
    enc.timing_enable(True)
    enc.encode_host(frame, 2 * w * h)
    ms, _ = enc.timing_read()
    ntr = 4096
    tr = (C.c_uint64 * (4 * ntr))()
    lib.icerx_prof_trace.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    lib.icerx_prof_trace(enc.handle, tr, ntr)
    t = np.array(tr, dtype=np.uint64).reshape(ntr, 4)
    t = t[t[:, 1] > 0]
    t0 = int(t[:, 0].min())
    start = (t[:, 0].astype(np.int64) - t0) / 1e5
    end = (t[:, 1].astype(np.int64) - t0) / 1e5
    is_sub = (t[:, 3] & np.uint64(0x80000000)) != 0
    end_chunk = (t[:, 3] >> np.uint64(32)).astype(np.int64)
    print(f"{w}x{h}: code_units {ms['code_units']:.2f} ms; {len(t)} workgroups ran to their end ({int(is_sub.sum())} sub-range workgroups)")
    for name, sel in (("sub-range workgroups", is_sub), ("whole-unit / first-sub-range workgroups", ~is_sub)):
        if not sel.any():
            continue
        d = end[sel] - start[sel]
        print(f"  {name}: n {int(sel.sum())}, start mean {start[sel].mean():.2f} max {start[sel].max():.2f} ms; duration mean {d.mean():.2f} p50 {np.median(d):.2f} "
              f"p90 {np.percentile(d, 90):.2f} max {d.max():.2f} ms; end max {end[sel].max():.2f} ms")
        hist, edges = np.histogram(end[sel], bins=10, range=(0, end.max()))
        print("     end-time histogram (ms): " + "  ".join(f"<{edges[i + 1]:.1f}:{hist[i]}" for i in range(10)))
        hist, edges = np.histogram(start[sel], bins=10, range=(0, end.max()))
        print("     start-time histogram (ms): " + "  ".join(f"<{edges[i + 1]:.1f}:{hist[i]}" for i in range(10)))
    order = np.argsort(-end)[:12]
    print("  latest finishers: kind, id, start -> end ms")
    for b in order:
        print(f"     {'sub ' if is_sub[b] else 'unit'} {int(t[b, 3] & np.uint64(0x7FFFFFFF)):5d}  {start[b]:6.2f} -> {end[b]:6.2f}   stopped at chunk {int(end_chunk[b])}")
    print("  stage ms:", {k: round(v, 3) for k, v in ms.items()})


if __name__ == "__main__":
    main()
