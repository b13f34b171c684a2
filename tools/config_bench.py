#!/usr/bin/env python3
"""Per-GPU share of the BASELINE.json configurations on ONE GPU (device-resident input, stream stays in HBM).
   python tools/config_bench.py            # C2, C3, C4 (32 of the 256 frames), C5 (8 of the 64 frames)
Prints one JSON line per configuration; parity is checked against tests/golden/golden.json where a golden exists."""
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from icer_compression_amd import api, synth  # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))


def run(name, w, h, channels, stages, segments, quota, frames, golden_first=None, steps=3):
    dev = torch.device("cuda", 0)
    if channels == 1:
        host = np.stack([synth.gray_frame(w, h, synth.DEFAULT_SEED + k, 1) for k in range(frames)])
    else:
        host = np.stack([np.stack(synth.color_frame_yuv(w, h, synth.DEFAULT_SEED + k)) for k in range(frames)])
    d = torch.from_numpy(host.view(np.int16)).to(dev)
    stride = min(quota, 2 * w * h * channels)
    out = torch.empty((frames, stride), dtype=torch.uint8, device=dev)
    sizes = torch.zeros(frames, dtype=torch.int64, device=dev)
    rcs = torch.zeros(frames, dtype=torch.int32, device=dev)
    enc = api.Encoder(w, h, channels, stages, 0, segments, max_frames=frames)
    enc.encode_torch(d, quota, out, sizes, rcs)
    torch.cuda.synchronize()
    ok = None
    if golden_first:
        g = GOLD[golden_first]
        s0 = out[0, : int(sizes[0])].cpu().numpy().tobytes()
        ok = int(rcs[0]) == g["rc"] and len(s0) == g["size"] and "%08x" % zlib.crc32(s0) == g["crc32"]
    t = time.perf_counter()
    for _ in range(steps):
        enc.encode_torch(d, quota, out, sizes, rcs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    print(json.dumps({"config": name, "frames_on_this_gpu": frames, "ms_per_launch": round(dt * 1e3, 3),
                      "Mpixels_per_s": round(frames * w * h / dt / 1e6, 1), "bytes_out": int(sizes.sum()),
                      "first_frame_matches_reference_golden": ok}), flush=True)
    enc.close()
    del d, out


if __name__ == "__main__":
    only = None                                  # --only C2,C4: a subset (measurements of launch variants)
    if "--only" in sys.argv:
        only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    if not only or "C2" in only:
        run("C2 4096^2 gray 5st 10seg lossless", 4096, 4096, 1, 5, 10, 2 * 4096 * 4096, 1, "C2_4096_gray_5st_10seg", steps=10 if only else 3)
    if not only or "C3" in only:
        run("C3 4096^2 YUV 5st 10seg quota 70000", 4096, 4096, 3, 5, 10, 70000, 1, "C3_4096_yuv_quota70000")
    if not only or "C4" in only:
        run("C4 2048^2 gray 4st 16seg, 32 of 256 frames", 2048, 2048, 1, 4, 16, 2 * 2048 * 2048, 32, "C4_2048_frame0")
    if not only or "C5" in only:
        run("C5 8192^2 gray 6st 32seg, 8 of 64 frames", 8192, 8192, 1, 6, 32, 2 * 8192 * 8192, 8, "C5_8192_frame0", steps=2)
