#!/usr/bin/env python3
"""Decode benchmark of libicer_hip_dec.so (HISTORY.md 6b (summary: DESIGN.md 8)) -- NOT bench.py's metric.

    python tools/decode_bench.py [--batch 16] [--reps 3] [--no-cpu-baseline]

Workload: the stream the HIP encoder makes of the BASELINE configs[1] frame (4096 x 4096 gray, 5 stages, filter A, 10
segments, lossless: the reference golden, checked), resident in HBM; decoded to uint16 planes that stay in HBM
(icerx_decode_device).  One JSON line: single-stream latency rate, rate with `--batch` streams per call (what the
kernels do when there is enough to do: one wavefront per segment-subband chain, 160 chains per stream), the HBM roofline
fraction of the batch run (algorithmic bytes = stream read once + planes written once) and the reference decoder
(oracle/_ref) on one host core on the same stream.  Every decoded frame is compared with the encoder's input.
"""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

W = H = 4096
STAGES, FILT, SEGMENTS = 5, 0, 10
HBM_PEAK_GBPS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    import torch
    from icer_compression_amd import api, decoder, synth
    assert torch.cuda.is_available(), "needs a HIP device"
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)                                   # torch's HIP runtime first (see tests/conftest.py)
    img = synth.gray_frame(W, H, 12345, 1)
    rc, stream, _ = api.compress([img], STAGES, FILT, SEGMENTS, 2 * W * H)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["C2_4096_gray_5st_10seg"]
    assert rc == 0 and len(stream) == gold["size"] and "%08x" % zlib.crc32(stream) == gold["crc32"], "encoder stream is not the golden"
    want = torch.from_numpy(img.view(np.int16)).to(dev)

    def run(n, reps):
        d_data = torch.from_numpy(np.frombuffer(stream * n, dtype=np.uint8).copy()).to(dev)
        d_out = torch.zeros((n, H * W), dtype=torch.int16, device=dev)
        dec = decoder.Decoder(1, STAGES, FILT, SEGMENTS)
        offs, lens = [k * len(stream) for k in range(n)], [len(stream)] * n
        times = []
        for _ in range(reps + 1):                                # (first call: allocations)
            torch.cuda.synchronize()
            t = time.perf_counter()
            rc2, rcs, ws, hs = dec.decode_device(n, d_data.data_ptr(), offs, lens, d_out.data_ptr(), W * H)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t)
        ok = rc2 == 0 and all(r == 0 for r in rcs) and all(bool(torch.equal(d_out[k].view(H, W), want)) for k in range(n))
        dec.close()
        return min(times[1:]), ok

    t1, ok1 = run(1, a.reps)
    tb, okb = run(a.batch, a.reps) if a.batch > 1 else (None, True)
    alg = float(len(stream) + W * H * 2)                         # per stream: read the stream, write the planes
    line = {"metric": "Mpixels/s decode (bit-exact), 4096x4096 gray", "unit": "Mpixels/s", "higher_is_better": True,
            "value": round(W * H / t1 / 1e6, 2), "ms_per_frame": round(t1 * 1e3, 2), "dtype": "int16", "data": "synthetic",
            "config": {"workload": "stream of BASELINE configs[1] (4096x4096 gray, 5 stages, filter A, 10 segments, lossless, 9 948 227 bytes = "
                                   "reference golden) resident in HBM -> uint16 planes in HBM; icerx_decode_device, whole call incl. the host's packet walk",
                       "kernel": {"0": "one thread per chain (decode_chains_kernel)", "1": "one wavefront per chain, one lane per bit plane (decode_chains_wave_kernel)"}.get(
                           os.environ.get("ICER_DEC_WAVE", ""), "one wavefront per bit plane, wave-uniform decisions (decode_chains_planes_kernel)"),
                       "parity": bool(ok1 and okb)},
            "roofline": {"bound": "hbm", "achieved": round(alg / t1 / 1e9, 4), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(alg / t1 / 1e9 / HBM_PEAK_GBPS, 7), "algorithmic_bytes_per_frame": alg,
                         "note": "whole call; a chain is a serial adaptive decode (one decision at a time per bit plane), far from bandwidth-bound"}}
    if tb:
        line["batched"] = {"streams_per_call": a.batch, "value": round(a.batch * W * H / tb / 1e6, 2), "unit": "Mpixels/s",
                           "ms_per_call": round(tb * 1e3, 2), "roofline_frac": round(a.batch * alg / tb / 1e9 / HBM_PEAK_GBPS, 7)}
    if not a.no_cpu_baseline:
        from oracle import binding
        binding.build()
        if binding.have_reference():
            ref, kind = binding.Reference(), "reference"
            t = time.perf_counter()
            out = ref.decompress(stream, 1, STAGES, FILT, SEGMENTS)
            dt = time.perf_counter() - t
            good = out[0] == 0 and np.array_equal(np.asarray(out[1][0]).reshape(H, W), img)
        else:
            ref, kind = binding.Oracle(), "port"
            t = time.perf_counter()
            out = ref.decompress(stream, 1, STAGES, FILT, SEGMENTS, bufsize=W * H)
            dt = time.perf_counter() - t
            good = out[0] == 0 and np.array_equal(np.asarray(out[3][0]).reshape(H, W), img)
        line["cpu_baseline"] = {"value": round(W * H / dt / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": kind,
                                "sample": f"the same stream, one full frame, {dt:.2f} s, decoded image equals the input: {bool(good)}"}
        line["speedup_vs_cpu_1thread"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
    print(json.dumps(line))
    return 0 if (ok1 and okb) else 1


if __name__ == "__main__":
    sys.exit(main())
