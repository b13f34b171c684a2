#!/usr/bin/env python3
"""Randomised parity stress on a GPU (not collected by pytest): random geometries, filters, segment counts,
quotas and content, HIP path through the C ABI vs the oracle.  Meant to shake out rare cross-wave races.
   python tests/stress_gpu.py [seconds] [seed]          (ICER_HIP_CODER=pipe|wg pins the coding-unit kernel)
Fails on any mismatch and on any coding-unit time-out (icerx_process_stats), even though a time-out no longer fails the
call.  tests/test_gpu_stress.py runs a bounded slice of it under pytest -m gpu.
With ICER_STRESS_BATCH=<n> every gray 16-bit case becomes a batch of n frames (the same geometry, rolled content) through
icerx_compress_batch_uint16_devices -- the batch build of the pipeline kernel, four workgroups per compute unit, sub-batches on
three streams -- each frame against the oracle.
With ICER_STRESS_DECODE=1 every stream the encoder produced is also decoded by libicer_hip_dec.so (all three decode kernels)
and compared with the decoder oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icer_compression_amd import api, synth  # noqa: E402
from oracle.binding import Oracle  # noqa: E402


def _packets(stream):
    out, off = [], 0
    while off + 28 <= len(stream):
        n = 28 + (int.from_bytes(stream[off + 16: off + 20], "little") + 7) // 8
        out.append(stream[off: off + n])
        off += n
    return out


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    orc = Oracle()
    decode = os.environ.get("ICER_STRESS_DECODE") == "1"
    if decode:
        from icer_compression_amd import decoder
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        big = rng.random() < float(os.environ.get('ICER_STRESS_BIG', '0.03'))                       # now and then a large frame (long coding units, deep queues)
        w, h = (int(rng.integers(700, 2200)), int(rng.integers(700, 2200))) if big else (int(rng.integers(5, 700)), int(rng.integers(5, 700)))
        st = int(rng.integers(1, 7))
        while ((w + (1 << st) - 1) >> st) < 3 or ((h + (1 << st) - 1) >> st) < 3:
            st -= 1
        filt, sg = int(rng.integers(0, 7)), int(rng.integers(1, 33))
        kind = rng.integers(0, 5)
        if kind == 0:
            img = synth.gray_frame(w, h, int(rng.integers(1, 1 << 30)), 0)
        elif kind == 1:
            img = synth.gray_frame(w, h, int(rng.integers(1, 1 << 30)), 1)
        elif kind == 2:
            img = rng.integers(0, int(rng.choice([2, 16, 1024, 4096])), (h, w)).astype(np.uint16)
        elif kind == 3:
            img = (rng.integers(0, 256, (h, w)) * (rng.random((h, w)) < rng.choice([0.01, 0.1, 0.5]))).astype(np.uint16)
        else:
            img = np.full((h, w), int(rng.integers(0, 256)), np.uint16)
        quota = int(rng.choice([2 * w * h + 64, max(64, w * h // 3), max(40, w * h // 20), 100 + int(rng.integers(0, 3000))]))
        color = rng.random() < 0.2
        planes = [img] if not color else [img, np.roll(img, 3, 0), np.roll(img, 5, 1)]
        u8 = rng.random() < 0.2                          # the uint8 twins on 6-bit versions of the same planes
        if u8:
            planes = [(p >> int(rng.choice([0, 2, 4]))).astype(np.uint8) for p in planes]
        nb = int(os.environ.get("ICER_STRESS_BATCH", "0"))
        if nb > 1 and not u8 and not color:
            frames = np.stack([np.roll(img, (3 * k, 7 * k), (0, 1)) for k in range(nb)])
            cap = quota + 64                             # (a stream never exceeds its quota; 2wh + 64 is NOT a bound: 32 segments of a 66 x 8 frame are mostly packet headers)
            out, sizes, rcs = np.zeros((nb, cap), np.uint8), np.zeros(nb, np.uint64), np.zeros(nb, np.int32)
            rc = api.compress_batch(frames, st, filt, sg, quota, out, sizes, rcs, devices=[0])
            n += 1
            for k in range(nb):
                b = orc.compress([frames[k]], st, filt, sg, quota)
                # (an argument error -- too many segments / stages for the geometry -- is the call's return code, as the
                # reference returns it for every frame)
                if (rc != b[0] or b[0] not in (-3, -4)) if rc != 0 else (int(rcs[k]) != b[0] or out[k, : int(sizes[k])].tobytes() != b[1]):
                    bad += 1
                    print("BATCH MISMATCH", dict(w=w, h=h, stages=st, filt=filt, segments=sg, quota=quota, kind=int(kind), frame=k), "rc", rc, int(rcs[k]), b[0],
                          api.load_library().icerx_last_error() if rc else "", flush=True)
                    break
            continue
        try:
            a = (api.compress_u8 if u8 else api.compress)(planes, st, filt, sg, quota)
        except Exception as exc:                                   # noqa: BLE001
            print("EXCEPTION", w, h, st, filt, sg, quota, exc)
            bad += 1
            continue
        b = (orc.compress_u8 if u8 else orc.compress)(planes, st, filt, sg, quota)
        same = a[0] == b[0] and a[1] == b[1] and (a[0] not in (0, -5) or all(np.array_equal(p, q) for p, q in zip(a[2], b[2])))
        n += 1
        if decode and same and a[1] and len({p[7] >> 4 for p in _packets(a[1])}) >= len(planes):
            bits = 8 if u8 else 16
            want = orc.decompress(a[1], len(planes), st, filt, sg, bufsize=w * h, bits=bits)
            for mode in ("0", "1", "2"):
                os.environ["ICER_DEC_WAVE"] = mode
                got = decoder.decompress(a[1], len(planes), st, filt, sg, bufsize=w * h, bits=bits)
                if not (got[0] == want[0] and got[1:3] == want[1:3] and all(np.array_equal(x, y) for x, y in zip(got[3], want[3]))):
                    bad += 1
                    print("DECODE MISMATCH", dict(w=w, h=h, stages=st, filt=filt, segments=sg, quota=quota, kind=int(kind), color=bool(color),
                                                  u8=bool(u8), kernel=mode), "rc", got[0], want[0], flush=True)
        if not same:
            bad += 1
            print("MISMATCH", dict(w=w, h=h, stages=st, filt=filt, segments=sg, quota=quota, kind=int(kind), color=bool(color), u8=bool(u8)),
                  "rc", a[0], b[0], "len", len(a[1]), len(b[1]), api.load_library().icerx_last_error() if a[0] == -10 else "", flush=True)
    st = api.process_stats()
    print(f"stress: {n} cases, {bad} mismatches, {time.time() - t0:.1f} s, coder={os.environ.get('ICER_HIP_CODER', 'auto')}, "
          f"unit time-outs {st['unit_timeouts']}, batches re-coded by the barrier-only coder {st['fallback_batches']}, slot re-runs {st['slot_retries']}")
    sys.exit(1 if bad or st["unit_timeouts"] else 0)


if __name__ == "__main__":
    main()
