#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ICER encoder.

Metric (BASELINE.json): Mpixels/s encode, bit-exact, 4096x4096 gray (+ achieved HBM GB/s of the
dominant kernel against the 8 TB/s roofline).  Workload = BASELINE.json configs[1]: one 4096x4096
8-bit synthetic frame (widened to uint16 like the reference's own callers), 5 DWT stages, filter A,
10 error-containment segments, lossless byte quota (2*W*H).  A "step" is one pass of the whole encode
hot path (DWT -> LL mean -> sign-magnitude -> all coding units -> quota scan -> final stream) over one
batch (default: ONE frame, as the config says) whose input already sits in HBM; the final stream stays
in HBM.  With --gpus N every rank encodes its own frame(s) per step (frames are independent; no
collective on the data path) -> weak scaling; value = all ranks' pixels / max-over-ranks time.

    python bench.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W = H = 4096
STAGES, FILT, SEGMENTS = 5, 0, 10
QUOTA = 2 * W * H
HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(frame: np.ndarray, expect_crc: str):
    """Reference CPU encoder (oracle/_ref, built from the untouched sources) on the host cores of this
    box, single thread (the library is non-reentrant), on the SAME 4096x4096 frame; falls back to our
    plain-C restatement (kind "port") when the reference build is absent."""
    from oracle import binding
    binding.build()
    if binding.have_reference():
        impl, kind = binding.Reference(), "reference"
    else:
        impl, kind = binding.Oracle(), "port"
    t = time.perf_counter()
    rc, stream, _ = impl.compress([frame], STAGES, FILT, SEGMENTS, QUOTA)
    dt = time.perf_counter() - t
    ok = rc == 0 and ("%08x" % zlib.crc32(stream)) == expect_crc
    return {"value": round(W * H / dt / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": kind,
            "sample": f"1 full {W}x{H} frame (same input as the GPU run), {dt:.2f} s, stream matches golden: {ok}"}


def cpu_all_cores():
    """the CPU encoder on ALL host cores of this box -- P worker processes (oracle/cpu_worker.py: the
    library is not re-entrant), one frame each (frame k uses seed 12345 + k), started together; value = P frames over the
    time of the slowest encode.  Bounded: every worker is killed after 120 s."""
    import subprocess
    from oracle import binding
    binding.build()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = max(1, min(cores, 32))
    cmd = [sys.executable, "-m", "oracle.cpu_worker"]
    ps = [subprocess.Popen(cmd + [str(12345 + k), str(W), str(H), str(STAGES), str(FILT), str(SEGMENTS)], cwd=ROOT,
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for k in range(procs)]
    times, ok = [], True
    deadline = time.time() + 120
    for p in ps:
        try:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
            rc, _, sec = out.split()
            ok = ok and rc == "0"
            times.append(float(sec))
        except Exception:
            p.kill()
            ok = False
    if not times:
        return {"error": "no CPU worker finished"}
    slowest = max(times)
    return {"value": round(len(times) * W * H / slowest / 1e6, 3), "unit": "Mpixels/s", "cores": procs,
            "kind": "reference" if binding.have_reference() else "port",
            "sample": f"{len(times)} of {procs} processes x 1 full {W}x{H} frame in parallel, slowest encode {slowest:.2f} s, all rc 0: {ok}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1, help="frames per rank per step (default 1 = BASELINE configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batched-probe", type=int, default=8,
                    help="also report throughput with this many frames per launch (secondary figure, 0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from icer_compression_amd import api, shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs one process per GPU: launch with torch.distributed.run "
                             f"--nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    B = args.batch
    seeds = shard.frame_seeds(synth.DEFAULT_SEED, rank, world, B)
    host_frames = np.stack([synth.gray_frame(W, H, s, 1) for s in seeds])
    frames = torch.from_numpy(host_frames.view(np.int16)).to(dev)             # resident in HBM before timing
    out = torch.empty((B, QUOTA), dtype=torch.uint8, device=dev)
    sizes = torch.zeros(B, dtype=torch.int64, device=dev)
    rcs = torch.zeros(B, dtype=torch.int32, device=dev)
    enc = api.Encoder(W, H, 1, STAGES, FILT, SEGMENTS, max_frames=B, device=local_rank)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # parity gate first (one extra untimed encode): rank 0's first frame is the golden C2 frame
    enc.encode_torch(frames, QUOTA, out, sizes, rcs)
    torch.cuda.synchronize(dev)
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as fh:
        gold = json.load(fh)["C2_4096_gray_5st_10seg"]
    h_sizes, h_rcs = sizes.cpu().numpy(), rcs.cpu().numpy()
    parity = bool((h_rcs == 0).all())
    if rank == 0:
        s0 = out[0, : int(h_sizes[0])].cpu().numpy().tobytes()
        parity = parity and len(s0) == gold["size"] and ("%08x" % zlib.crc32(s0)) == gold["crc32"]
    if not parity:
        raise SystemExit(f"rank {rank}: output is not bit-exact (rc={h_rcs.tolist()}, sizes={h_sizes.tolist()}); no number reported")

    # W untimed warm-up steps directly before the timed region (the GPU clocks fall back while the host checks parity)
    for _ in range(args.warmup):
        enc.encode_torch(frames, QUOTA, out, sizes, rcs)
    torch.cuda.synchronize(dev)

    enc.timing_enable(True)
    enc.timing_read(reset=True)
    barrier()
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        enc.encode_torch(frames, QUOTA, out, sizes, rcs)
        step_ms.append(round((time.perf_counter() - ts) * 1e3, 3))
    barrier()
    elapsed = time.perf_counter() - t0
    stage_ms, calls = enc.timing_read(reset=True)
    enc.timing_enable(False)

    elapsed_max = shard.max_over_ranks(elapsed, dev)

    # secondary figure: the same frame geometry with several frames per launch (what BASELINE configs[3]/[4]
    # do); one frame alone cannot fill 256 CUs because its largest coding units form a serial chain
    batched = None
    if args.batched_probe > 1:
        PB = args.batched_probe
        bf = torch.from_numpy(np.stack([synth.gray_frame(W, H, s, 1) for s in shard.frame_seeds(synth.DEFAULT_SEED, rank, world, PB)]).view(np.int16)).to(dev)
        bout = torch.empty((PB, QUOTA), dtype=torch.uint8, device=dev)
        bsizes = torch.zeros(PB, dtype=torch.int64, device=dev)
        brcs = torch.zeros(PB, dtype=torch.int32, device=dev)
        benc = api.Encoder(W, H, 1, STAGES, FILT, SEGMENTS, max_frames=PB, device=local_rank)
        benc.encode_torch(bf, QUOTA, bout, bsizes, brcs)
        barrier()
        tb = time.perf_counter()
        nb = max(2, args.steps // 2)
        for _ in range(nb):
            benc.encode_torch(bf, QUOTA, bout, bsizes, brcs)
        barrier()
        tb = shard.max_over_ranks(time.perf_counter() - tb, dev)
        ok = bool((brcs.cpu().numpy() == 0).all()) and (rank != 0 or int(bsizes[0]) == gold["size"])
        batched = {"frames_per_gpu_per_launch": PB, "value": round(world * PB * W * H * nb / tb / 1e6, 3), "unit": "Mpixels/s",
                   "ms_per_launch": round(tb / nb * 1e3, 3), "parity": ok}
        benc.close()
        del bf, bout

    if rank == 0:
        n_pix = world * B * W * H * args.steps
        value = n_pix / elapsed_max / 1e6
        # roofline of the dominant kernel (code_units_kernel): algorithmic bytes per launch =
        # SURVEY 8(d) per-frame figure (input planes read once + final stream written once) x frames per launch
        k_ms = stage_ms["code_units"] / max(calls, 1)
        alg_bytes = float(B * W * H * 2 + int(h_sizes.sum()))
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        line = {
            "metric": "Mpixels/s encode (bit-exact), 4096x4096 gray", "value": round(value, 3), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: {B} x {W}x{H} 8-bit gray (uint16 API) per GPU per step, "
                                   f"{STAGES} DWT stages, filter A, {SEGMENTS} segments, lossless quota 2*W*H; input and "
                                   "output stream resident in HBM", "frames_per_gpu_per_step": B,
                       "units_per_frame": enc.info()["units_per_frame"], "parity": "stream CRC equals reference golden"},
            "roofline": {"bound": "hbm", "kernel": "code_units_kernel", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(k_ms, 4)},
            "stage_ms_per_step": {k: round(v / max(calls, 1), 4) for k, v in stage_ms.items()},
            "step_ms": step_ms,
        }
        if batched:
            line["batched"] = batched
        pmc = os.path.join(ROOT, "profiles", "latest_pmc.json")
        if os.path.exists(pmc):                      # HBM traffic of code_units_kernel from the committed PMC passes
            with open(pmc) as fh:
                t = json.load(fh)
            line["roofline"]["traffic"] = t.get("traffic_bytes_per_launch")
            line["roofline"]["traffic_source"] = t.get("source")
        if world == 1:
            # PCIe-inclusive figure of the host-buffer entry point (never `value`): H2D frame + kernels + D2H stream
            t0h = time.perf_counter()
            reps = 3
            for _ in range(reps):
                (hrc, hstream), = enc.encode_host(host_frames[:1], QUOTA)
            th = (time.perf_counter() - t0h) / reps
            line["host_buffers"] = {"ms_per_frame": round(th * 1e3, 3), "value": round(W * H / th / 1e6, 3), "unit": "Mpixels/s",
                                    "note": "icerx_encode_host: pageable H2D of the frame, all kernels, D2H of size/rc/stream",
                                    "parity": hrc == 0 and ("%08x" % zlib.crc32(hstream)) == gold["crc32"]}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(host_frames[0], gold["crc32"])
            line["speedup_vs_cpu_1thread"] = round(value / line["cpu_baseline"]["value"], 2)
            try:
                line["cpu_all_cores"] = cpu_all_cores()
            except Exception as exc:                                   # reported, never fatal for the GPU numbers
                line["cpu_all_cores"] = {"error": repr(exc)}
        print(json.dumps(line), flush=True)
    enc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
