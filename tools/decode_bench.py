#!/usr/bin/env python3
"""First measurement of the decoder library (libicer_hip_dec.so) on a GPU box -- NOT bench.py's metric.

    python tools/decode_bench.py [--size 4096] [--stages 5] [--segments 10] [--reps 3]

Encodes one synthetic gray frame with the HIP encoder, decodes the stream through the lib_icer-shaped entry point
(host buffers: includes the PCIe copies and the device allocations of this first version), checks the result against
the input (the stream is lossless for this data) and prints one JSON line with the decode rate.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icer_compression_amd import api, decoder, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--stages", type=int, default=5)
    ap.add_argument("--segments", type=int, default=10)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    w = h = a.size
    img = synth.gray_frame(w, h, 12345, 1)
    rc, stream, _ = api.compress([img], a.stages, 0, a.segments, 2 * w * h)
    assert rc == 0, rc
    times = []
    for _ in range(a.reps):
        t = time.perf_counter()
        drc, dw, dh, planes = decoder.decompress(stream, 1, a.stages, 0, a.segments, bufsize=w * h)
        times.append(time.perf_counter() - t)
    ok = drc == 0 and (dw, dh) == (w, h) and np.array_equal(planes[0].reshape(h, w), img)
    best = min(times)
    print(json.dumps({"what": "decode, host buffers, first device version", "w": w, "h": h, "stages": a.stages,
                      "segments": a.segments, "stream_bytes": len(stream), "round_trip_exact": bool(ok),
                      "seconds": [round(t, 4) for t in times], "Mpix_per_s": round(w * h / best / 1e6, 2)}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
