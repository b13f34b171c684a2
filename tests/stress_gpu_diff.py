#!/usr/bin/env python3
"""High-rate differential stress on a GPU (not collected by pytest): the two coding-unit kernels -- the wave pipeline
(both workgroup shapes and the batch default with the small workgroup coder beside it, taking turns) and the barrier-only workgroup coder -- encode the same random batches (random geometry, filter, segment
count, quota class, content made on the device) and must produce the same return codes, lengths and bytes; a sample of
the frames is also checked against the oracle.  No oracle call sits in the inner loop, so this runs thousands of
encodes per second: it is the campaign that looks for the hand-off stall of the pipeline (HISTORY.md 4.1).
   python tests/stress_gpu_diff.py [seconds] [seed]
Prints one summary line; exit code 1 on any difference or any coding-unit time-out (icerx_process_stats)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_encoder(api, coder, *a, **k):
    """coder: "wg", "pipe8" / "pipe11" (the wave pipeline with its workgroup shape pinned), or "auto": the library's own
    choice for a batch -- the pipeline with the all-but-blank units routed to the small workgroup coder beside it (threshold
    lowered to 80 % so that more units take that way), the workgroup coder in progressive mode"""
    if coder == "auto":
        os.environ["ICER_HIP_HYBRID"] = "80"
        os.environ["ICER_HIP_HYBRID_FRAMES"] = "1"
    else:
        os.environ["ICER_HIP_CODER"] = "pipe" if coder.startswith("pipe") else coder
        if coder.startswith("pipe") and coder[4:]:
            os.environ["ICER_HIP_PIPE_WAVES"] = coder[4:]
    try:
        return api.Encoder(*a, **k)
    finally:
        for v in ("ICER_HIP_CODER", "ICER_HIP_PIPE_WAVES", "ICER_HIP_HYBRID", "ICER_HIP_HYBRID_FRAMES"):
            os.environ.pop(v, None)


def main():
    import torch
    from icer_compression_amd import api
    from oracle.binding import Oracle
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    gen = torch.Generator(device="cuda").manual_seed(seed)
    dev = torch.device("cuda", 0)
    orc = Oracle()
    t0 = time.time()
    cases = frames_done = bad = checked = 0
    while time.time() - t0 < budget:
        klass = rng.random()
        if klass < 0.15:                                   # large frames, long coding units (the class of the round-1 stall)
            w, h, n = int(rng.integers(900, 2300)), int(rng.integers(900, 2300)), int(rng.integers(2, 7))
        elif klass < 0.5:
            w, h, n = int(rng.integers(200, 900)), int(rng.integers(200, 900)), int(rng.integers(4, 33))
        else:
            w, h, n = int(rng.integers(8, 200)), int(rng.integers(8, 200)), int(rng.integers(16, 129))
        st = int(rng.integers(1, 7))
        while ((w + (1 << st) - 1) >> st) < 3 or ((h + (1 << st) - 1) >> st) < 3:
            st -= 1
        filt, sg = int(rng.integers(0, 7)), int(rng.integers(1, 33))
        ch = 3 if rng.random() < 0.15 else 1
        quota = int(rng.choice([2 * w * h * ch + 64, max(64, w * h // 3), max(40, w * h // 20), 100 + int(rng.integers(0, 3000))]))
        # content on the device: noise of some amplitude thinned out to some density, on a smooth ramp or not
        amp = int(rng.choice([2, 16, 256, 256, 1024, 4096]))
        dens = float(rng.choice([0.01, 0.1, 0.5, 1.0]))
        x = torch.randint(0, amp, (n, ch, h, w), generator=gen, device=dev, dtype=torch.int32)
        x = x * (torch.rand((n, ch, h, w), generator=gen, device=dev) < dens)
        if rng.random() < 0.5:
            x = x + (torch.arange(w, device=dev, dtype=torch.int32)[None, None, None, :] * 100 // w)
        frames = x.clamp_(0, 32767).to(torch.int16).contiguous()
        res = []
        try:
            # the pipeline's two workgroup shapes and the batch default (both coders in one launch) take turns against the
            # workgroup coder
            for coder in (("pipe8", "pipe11", "auto")[cases % 3], "wg"):
                enc = make_encoder(api, coder, w, h, ch, st, filt, sg, max_frames=n)
                if enc.create_rc != 0:
                    res.append((enc.create_rc,))
                    continue
                out = torch.zeros((n, quota + 64), dtype=torch.uint8, device=dev)
                sizes = torch.zeros(n, dtype=torch.int64, device=dev)
                rcs = torch.zeros(n, dtype=torch.int32, device=dev)
                enc.encode_torch(frames, quota, out, sizes, rcs)
                torch.cuda.synchronize()
                res.append((0, out, sizes, rcs))
                enc.close()
        except Exception as exc:                                   # noqa: BLE001
            print("EXCEPTION", dict(w=w, h=h, n=n, st=st, filt=filt, sg=sg, ch=ch, quota=quota), exc, flush=True)
            bad += 1
            continue
        cases += 1
        a, b = res
        if a[0] != b[0]:
            same = False
        elif a[0] != 0:
            same = True                                     # both refused the geometry alike
        else:
            same = bool(torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]))
            if same:
                # bytes beyond a stream's length are unspecified: compare the used parts
                idx = torch.arange(a[1].shape[1], device=dev)[None, :] < a[2][:, None]
                same = bool(torch.equal(a[1] * idx, b[1] * idx))
            frames_done += 2 * n
            if same and cases % 25 == 0:                    # a sample against the oracle
                k = int(rng.integers(0, n))
                planes = [frames[k, c].cpu().numpy().view(np.uint16) for c in range(ch)]
                want = orc.compress(planes, st, filt, sg, quota)
                got = (int(a[3][k]), a[1][k, : int(a[2][k])].cpu().numpy().tobytes())
                checked += 1
                if (want[0], want[1]) != got:
                    same = False
        if not same:
            bad += 1
            print("MISMATCH", dict(w=w, h=h, n=n, st=st, filt=filt, sg=sg, ch=ch, quota=quota, amp=amp, dens=dens), flush=True)
    stt = api.process_stats()
    print(f"differential stress: {cases} batches, {frames_done} encodes (both coders), {checked} frames also against the oracle, {bad} mismatches, "
          f"{time.time() - t0:.1f} s, unit time-outs {stt['unit_timeouts']}, fall-back batches {stt['fallback_batches']}, slot re-runs {stt['slot_retries']}")
    sys.exit(1 if bad or stt["unit_timeouts"] else 0)


if __name__ == "__main__":
    main()
