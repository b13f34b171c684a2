#!/usr/bin/env python3
"""Timeline of code_units_list_kernel in a single-frame launch (profiling build): which workgroup coded which unit of the
window coder's list, from when to when -- beside the pipeline kernel's workgroup trace (tools/split_trace.py).
   python tools/list_trace.py [w h stages segments]      (needs a GPU; ICER_HIP_* knobs as usual)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icer_compression_amd import api, build, synth  # noqa: E402


def main():
    w, h, st, sg = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (4096, 4096, 5, 10)))
    api.LIB_PATH = os.environ.get("ICER_HIP_PROF_LIB") or build.build_profiling_library()
    lib = api.load_library()
    enc = api.Encoder(w, h, 1, st, 0, sg, max_frames=1)
    frame = synth.gray_frame(w, h, 12345, 1)[None]
    enc.encode_host(frame, 2 * w * h)
    n = 4096
    lt = (C.c_uint64 * (4 * n))()
    lib.icerx_prof_list_trace.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int, C.c_int]
    lib.icerx_prof_list_trace(enc.handle, lt, n, 1)
    enc.timing_enable(True)
    enc.encode_host(frame, 2 * w * h)
    ms, _ = enc.timing_read()
    lib.icerx_prof_list_trace(enc.handle, lt, n, 0)
    ntr = 4096
    tr = (C.c_uint64 * (4 * ntr))()
    lib.icerx_prof_trace.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    lib.icerx_prof_trace(enc.handle, tr, ntr)
    pt = np.array(tr, dtype=np.uint64).reshape(ntr, 4)
    pt = pt[pt[:, 1] > 0]
    t = np.array(lt, dtype=np.uint64).reshape(n, 4)
    t = t[t[:, 1] > 0]
    t0 = int(min(t[:, 0].min(), pt[:, 0].min() if len(pt) else t[:, 0].min()))
    start = (t[:, 0].astype(np.int64) - t0) / 1e5
    end = (t[:, 1].astype(np.int64) - t0) / 1e5
    wg = (t[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    unit = (t[:, 2] >> np.uint64(32)).astype(np.int64)
    lsb = (t[:, 3] & np.uint64(0xFF)).astype(np.int64)
    level = ((t[:, 3] >> np.uint64(8)) & np.uint64(0xFF)).astype(np.int64)
    sub = ((t[:, 3] >> np.uint64(16)) & np.uint64(0xFF)).astype(np.int64)
    seg = ((t[:, 3] >> np.uint64(24)) & np.uint64(0xFF)).astype(np.int64)
    chunks = (t[:, 3] >> np.uint64(32)).astype(np.int64)
    d = end - start
    print(f"{w}x{h} st={st} seg={sg}: launch {enc.launch_info()}, code_units stage {ms['code_units']:.2f} ms")
    if len(pt):
        pe = (pt[:, 1].astype(np.int64) - t0) / 1e5
        print(f"  pipeline kernel: {len(pt)} workgroups, first start {(int(pt[:, 0].min()) - t0) / 1e5:.2f}, last end {pe.max():.2f} ms")
    print(f"  list kernel: {len(t)} units by {len(np.unique(wg))} workgroups; first start {start.min():.2f}, last end {end.max():.2f} ms; unit time total {d.sum():.1f} ms, "
          f"mean {d.mean():.3f}, max {d.max():.2f} ms")
    print("  units by duration (top 24): level lsb subband segment chunks  workgroup  start -> end  (ms)")
    for k in np.argsort(-d)[:24]:
        print(f"     L{level[k]} lsb{lsb[k]} sb{sub[k]} seg{seg[k]:2d} {chunks[k]:6d}   wg{wg[k]:4d}   {start[k]:6.2f} -> {end[k]:6.2f}  ({d[k]:.2f})")
    print("  by (level, lsb): units, mean / max duration ms, mean start ms")
    for lv in sorted(set(level)):
        for b in sorted(set(lsb[level == lv])):
            m = (level == lv) & (lsb == b)
            print(f"     L{lv} lsb{b}: {int(m.sum()):4d}  {d[m].mean():.3f} / {d[m].max():.3f}   start {start[m].mean():.2f}")
    # the busiest workgroups
    tot = {}
    for k in range(len(t)):
        tot.setdefault(int(wg[k]), []).append(k)
    busiest = sorted(tot, key=lambda g: -end[tot[g]].max())[:8]
    print("  workgroups that end last: workgroup, units, busy ms, first start -> last end")
    for g in busiest:
        ks = tot[g]
        print(f"     wg{g:4d}: {len(ks):3d} units, busy {d[ks].sum():.2f} ms, {start[ks].min():.2f} -> {end[ks].max():.2f};  " +
              ", ".join(f"L{level[k]}b{lsb[k]}({d[k]:.2f})" for k in sorted(ks, key=lambda k: start[k])[:10]))


if __name__ == "__main__":
    main()
