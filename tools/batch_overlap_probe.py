#!/usr/bin/env python3
"""Device-resident batch of a BASELINE configuration as K concurrent launches of per_gpu / K frames (K encoders, K streams,
icerx_encode_device_async + icerx_encoder_wait) against the one launch of per_gpu frames bench.py times: do the tails of
the launches overlap?   python tools/batch_overlap_probe.py C4|C5 [K ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from icer_compression_amd import api
    name = sys.argv[1] if len(sys.argv) > 1 else "C4"
    ks = [int(x) for x in sys.argv[2:]] or [1, 2, 4]
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    wl = bench.Workload(name, 0, dev, 0)
    c = wl.cfg
    wl.step(); bad, _ = wl.verify()
    assert not bad
    res = {}
    for k in ks:
        per = (wl.B + k - 1) // k
        encs = [api.Encoder(wl.w, wl.h, 1, c["stages"], bench.FILT, c["segments"], max_frames=per, device=0) for _ in range(k)] if k > 1 else [wl.enc]
        streams = [torch.cuda.Stream(device=dev) for _ in range(k)]

        def step():
            if k == 1:
                wl.step(); return
            for i in range(k):
                lo, hi = i * per, min((i + 1) * per, wl.B)
                encs[i].encode_device_async_ptrs(wl.frames[lo:hi].data_ptr(), hi - lo, wl.quota, wl.out[lo:hi].data_ptr(), wl.out.stride(0),
                                                 wl.sizes[lo:hi].data_ptr(), wl.rcs[lo:hi].data_ptr(), streams[i].cuda_stream)
            for e in encs:
                e.wait()
        wl.sizes.zero_()
        step(); torch.cuda.synchronize()
        bad, _ = wl.verify()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        bad2, _ = wl.verify()
        pix = wl.B * wl.w * wl.h
        res[k] = {"launches_in_flight": k, "frames_each": per, "ms_min": round(min(ts) * 1e3, 3), "ms_mean": round(sum(ts) / len(ts) * 1e3, 3),
                  "mpix_s_mean": round(pix / (sum(ts) / len(ts)) / 1e6, 1), "mpix_s_best": round(pix / min(ts) / 1e6, 1), "parity": not bad and not bad2}
        if k > 1:
            for e in encs:
                e.close()
    print(json.dumps({"config": name, "results": res}))


if __name__ == "__main__":
    main()
