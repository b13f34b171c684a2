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
    ap.add_argument("--batch", type=int, default=8)
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
    # batch of identical streams through the decoder object (host buffers): what the kernels do when there is enough to do
    batch = {}
    if a.batch > 1:
        dec = decoder.Decoder(1, a.stages, 0, a.segments)
        tb = []
        for _ in range(a.reps):
            t = time.perf_counter()
            brc, res = dec.decode_host([stream] * a.batch, w * h)
            tb.append(time.perf_counter() - t)
        bok = brc == 0 and all(r[0] == 0 and np.array_equal(r[3][0].reshape(h, w), img) for r in res)
        batch = {"batch": a.batch, "batch_exact": bool(bok), "batch_seconds": [round(t, 4) for t in tb],
                 "batch_Mpix_per_s": round(a.batch * w * h / min(tb) / 1e6, 2)}
        ok = ok and bok
    print(json.dumps({"what": "decode, host buffers, first device version", "w": w, "h": h, "stages": a.stages,
                      "segments": a.segments, "stream_bytes": len(stream), "round_trip_exact": bool(ok),
                      "kernel": "wave-per-chain" if os.environ.get("ICER_DEC_WAVE") == "1" else "thread-per-chain",
                      "seconds": [round(t, 4) for t in times], "Mpix_per_s": round(w * h / best / 1e6, 2), **batch}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
