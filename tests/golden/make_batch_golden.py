"""Regenerates tests/golden/batch_golden.json: (stream length, zlib CRC-32) of the UNTOUCHED reference encoder
(oracle/_ref/libicer_ref.so, built by oracle/Makefile from /root/reference) for EVERY frame of the two batch
configurations of BASELINE.json -- C4: 256 x 2048x2048 (seeds 12345 .. 12345+255, 4 stages, 16 segments) and
C5: 64 x 8192x8192 (seeds 12345 .. 12345+63, 6 stages, 32 segments), filter A, lossless quota 2*W*H, inputs from
icer_compression_amd.synth.gray_frame(mode 1).  bench.py checks every rank's streams of a multi-GPU run against it.

    python tests/golden/make_batch_golden.py [processes]      (authoring container; about 10 min on 8 cores)

The reference library is not re-entrant, hence worker PROCESSES."""
import json
import multiprocessing as mp
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CONFIGS = {"C4": dict(w=2048, h=2048, stages=4, segments=16, frames=256), "C5": dict(w=8192, h=8192, stages=6, segments=32, frames=64)}


def one(job):
    name, k = job
    from icer_compression_amd import synth
    from oracle.binding import Reference
    c = CONFIGS[name]
    img = synth.gray_frame(c["w"], c["h"], synth.DEFAULT_SEED + k, 1)
    rc, stream, _ = Reference().compress([img], c["stages"], 0, c["segments"], 2 * c["w"] * c["h"])
    return name, k, rc, len(stream), "%08x" % zlib.crc32(stream)


def main():
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    jobs = [("C5", k) for k in range(CONFIGS["C5"]["frames"])] + [("C4", k) for k in range(CONFIGS["C4"]["frames"])]
    out = {n: {"config": {**c, "filter": 0, "quota": 2 * c["w"] * c["h"], "base_seed": 12345, "mode": 1},
               "frames": [None] * c["frames"]} for n, c in CONFIGS.items()}
    t0 = time.time()
    with mp.Pool(procs) as pool:
        for i, (name, k, rc, size, crc) in enumerate(pool.imap_unordered(one, jobs)):
            assert rc == 0, (name, k, rc)
            out[name]["frames"][k] = [size, crc]
            if i % 16 == 0:
                print(f"{i + 1}/{len(jobs)} frames, {time.time() - t0:.0f} s", flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "batch_golden.json"), "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("done", time.time() - t0)


if __name__ == "__main__":
    main()
