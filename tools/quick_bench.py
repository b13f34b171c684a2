#!/usr/bin/env python3
"""One geometry, device-resident, timed with HIP events around the stream: ms per launch + the coder stage's share.
   python tools/quick_bench.py W H STAGES SEGMENTS FRAMES [STEPS]      (ICER_HIP_LIB / ICER_HIP_* select variants)
Parity: every frame's stream against the oracle's CRC is NOT checked here (use the test-suite); frame 0 against a golden when one exists."""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from icer_compression_amd import api, synth  # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
KEYS = {(4096, 4096, 5, 10): "C2_4096_gray_5st_10seg", (2048, 2048, 4, 16): "C4_2048_frame0", (8192, 8192, 6, 32): "C5_8192_frame0"}


def main():
    w, h, st, sg, frames = (int(x) for x in sys.argv[1:6])
    steps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
    dev = torch.device("cuda", 0)
    d = synth.gray_frames_torch(frames, w, h, synth.DEFAULT_SEED, dev, 1)
    quota = int(os.environ.get("QB_QUOTA", 2 * w * h))           # (QB_QUOTA=<bytes>: a rate-limited encode, progressive mode below W*H/2)
    out = torch.empty((frames, quota), dtype=torch.uint8, device=dev)
    sizes = torch.zeros(frames, dtype=torch.int64, device=dev)
    rcs = torch.zeros(frames, dtype=torch.int32, device=dev)
    enc = api.Encoder(w, h, 1, st, 0, sg, max_frames=frames)
    for _ in range(2):
        enc.encode_torch(d, quota, out, sizes, rcs)
    torch.cuda.synchronize()
    ok = None
    g = GOLD.get(KEYS.get((w, h, st, sg), ""))
    if g:
        s0 = out[0, : int(sizes[0])].cpu().numpy().tobytes()
        ok = int(rcs[0]) == g["rc"] and len(s0) == g["size"] and "%08x" % zlib.crc32(s0) == g["crc32"]
    enc.timing_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        enc.encode_torch(d, quota, out, sizes, rcs)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    stage, _ = enc.timing_read()
    print(json.dumps({"geom": [w, h, st, sg, frames], "ms": round(ms, 3), "Mpix_s": round(frames * w * h / ms / 1e3, 1),
                      "stages_ms": {k: round(v, 3) for k, v in stage.items()}, "golden0": ok,
                      "env": {k: v for k, v in os.environ.items() if k.startswith("ICER_HIP")}}), flush=True)
    enc.close()


if __name__ == "__main__":
    main()
