#!/usr/bin/env python3
"""Two encoders of a BASELINE batch configuration taking launches in turns (icerx_encode_device_async on two streams; the
host waits for launch k - 1 after it has submitted launch k): does the tail of one launch hide behind the next?
   python tools/batch_pingpong_probe.py C4|C5 [launches [encoders]]"""
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
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 2             # encoders taking the launches in turns
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    wls = [bench.Workload(name, r, dev, 0) for r in range(depth)]    # different shares of the configuration
    c = wls[0].cfg
    streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    for w in wls:
        w.step(); bad, _ = w.verify(); assert not bad
        w.sizes.zero_()

    def launch(i):
        w = wls[i]
        w.enc.encode_device_async_ptrs(w.frames.data_ptr(), w.B, w.quota, w.out.data_ptr(), w.out.stride(0), w.sizes.data_ptr(), w.rcs.data_ptr(), streams[i].cuda_stream)

    def run(pipelined):
        torch.cuda.synchronize(); t = time.perf_counter()
        if pipelined:
            for k in range(n):
                if k >= depth:
                    wls[k % depth].enc.wait()                     # (the launch this encoder took `depth` launches ago)
                launch(k % depth)
            for w in wls:
                w.enc.wait()
        else:
            for k in range(n):
                launch(k % depth); wls[k % depth].enc.wait()
        torch.cuda.synchronize()
        return time.perf_counter() - t
    run(True)
    res = {}
    for mode in (False, True, False, True):
        dt = run(mode)
        ok = all(not w.verify()[0] for w in wls)
        res.setdefault(f"{depth}_in_flight" if mode else "one_at_a_time", []).append({"ms_per_launch": round(dt / n * 1e3, 3), "mpix_s": round(n * wls[0].B * c["w"] * c["h"] / dt / 1e6, 1), "parity": ok})
    print(json.dumps({"config": name, "launches": n, "results": res}))


if __name__ == "__main__":
    main()
