#!/usr/bin/env python3
"""The host-fed batch call (icerx_compress_batch_uint16_devices, this GPU's share of C4 / C5 from page-locked memory) in a quiet
process and in a crowded one (two torch streams that have run kernels + a live decoder), with the environment as given:
   GPU_MAX_HW_QUEUES=4 ICER_HIP_STREAM_PRIO=0|1 python tools/host_batch_probe.py [C4|C5]     (needs a GPU)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "GPU_MAX_HW_QUEUES" in os.environ and os.environ["GPU_MAX_HW_QUEUES"] == "default":
    del os.environ["GPU_MAX_HW_QUEUES"]
    os.environ["ICER_BENCH_KEEP_HW_QUEUES"] = "1"
import bench  # noqa: E402  (sets GPU_MAX_HW_QUEUES=8 unless it is set)
if os.environ.get("ICER_BENCH_KEEP_HW_QUEUES") == "1":
    os.environ.pop("GPU_MAX_HW_QUEUES", None)


def rate(hw, n):
    hw.step()
    t0 = time.perf_counter()
    for _ in range(n):
        hw.step()
    dt = time.perf_counter() - t0
    bad, _ = hw.verify()
    c = hw.cfg
    return round(n * c["per_gpu"] * c["w"] * c["h"] / dt / 1e6, 1), not bad


def main():
    import torch
    from icer_compression_amd import decoder
    name = sys.argv[1] if len(sys.argv) > 1 else "C4"
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    n = 8 if name == "C4" else 4
    hw = bench.HostWorkload(name, 0, dev, 0)
    quiet, ok1 = rate(hw, n)
    hw.close()
    others = [torch.cuda.Stream(device=dev) for _ in range(2)]
    keep = []
    for st in others:
        with torch.cuda.stream(st):
            keep.append(torch.ones(1 << 20, device=dev) * 2)
    torch.cuda.synchronize(dev)
    c = bench.CONFIGS[name]
    dec = decoder.Decoder(1, c["stages"], 0, c["segments"])
    hw = bench.HostWorkload(name, 0, dev, 0)
    crowded, ok2 = rate(hw, n)
    hw.close()
    dec.close()
    print(json.dumps({"config": name, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "ICER_HIP_STREAM_PRIO": os.environ.get("ICER_HIP_STREAM_PRIO"),
                      "quiet_Mpix_s": quiet, "crowded_Mpix_s": crowded, "parity": ok1 and ok2}))


if __name__ == "__main__":
    main()
