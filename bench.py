#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ICER encoder.

Metric (BASELINE.json): Mpixels/s encode, bit-exact, 4096x4096 gray (+ achieved HBM GB/s of the
dominant kernel against the 8 TB/s roofline).  Workload = BASELINE.json configs[1] ("C2"): one 4096x4096
8-bit synthetic frame (widened to uint16 like the reference's own callers), 5 DWT stages, filter A,
10 error-containment segments, lossless byte quota (2*W*H).  A "step" is one pass of the whole encode
hot path (DWT -> LL mean -> sign-magnitude -> all coding units -> quota scan -> final stream) over one
batch (default: ONE frame, as the config says) whose input already sits in HBM; the final stream stays
in HBM.  With --gpus N every rank encodes its own copy of the workload per step (frames are independent; no
collective on the data path) -> weak scaling; value = all ranks' pixels / max-over-ranks time.  EVERY rank checks
its streams against the reference goldens before anything is timed.

The same line also carries, as secondary objects, the two batch configurations BASELINE.json names for 8 GPUs, run
as this GPU's share of them: "C4" = 256 x 2048x2048 (32 frames per GPU, frames 32*rank ..), "C5" = 64 x 8192x8192
(8 frames per GPU); every frame of every rank is checked against tests/golden/batch_golden.json (reference CPU
encoder).  --config C4|C5 makes one of them the timed workload instead.

    python bench.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FILT = 0
CONFIGS = {
    # name: geometry, frames per GPU per step, frames of the whole (8-GPU) batch
    "C2": dict(w=4096, h=4096, stages=5, segments=10, per_gpu=1, total=1, what="BASELINE configs[1]"),
    "C4": dict(w=2048, h=2048, stages=4, segments=16, per_gpu=32, total=256, what="BASELINE configs[3]"),
    "C5": dict(w=8192, h=8192, stages=6, segments=32, per_gpu=8, total=64, what="BASELINE configs[4]"),
}
HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
CUS, SIMDS_PER_CU = 256, 4      # MI355X; one wave64 VALU instruction occupies a SIMD for 4 cycles


def cpu_baseline(frame: np.ndarray, expect_crc: str, cfg):
    """Reference CPU encoder (oracle/_ref, built from the untouched sources) on the host cores of this
    box, single thread (the library is non-reentrant), on the SAME 4096x4096 frame; falls back to our
    plain-C restatement (kind "port") when the reference build is absent."""
    from oracle import binding
    binding.build()
    if binding.have_reference():
        impl, kind = binding.Reference(), "reference"
    else:
        impl, kind = binding.Oracle(), "port"
    w, h = cfg["w"], cfg["h"]
    t = time.perf_counter()
    rc, stream, _ = impl.compress([frame], cfg["stages"], FILT, cfg["segments"], 2 * w * h)
    dt = time.perf_counter() - t
    ok = rc == 0 and ("%08x" % zlib.crc32(stream)) == expect_crc
    return {"value": round(w * h / dt / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": kind,
            "sample": f"1 full {w}x{h} frame (same input as the GPU run), {dt:.2f} s, stream matches golden: {ok}"}


def cpu_all_cores(cfg):
    """the CPU encoder on ALL host cores of this box -- P worker processes (oracle/cpu_worker.py: the
    library is not re-entrant), one frame each (frame k uses seed 12345 + k), started together; value = P frames over the
    time of the slowest encode.  Bounded: every worker is killed after 120 s."""
    from oracle import binding
    binding.build()
    w, h = cfg["w"], cfg["h"]
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = max(1, min(cores, 32))
    cmd = [sys.executable, "-m", "oracle.cpu_worker"]
    ps = [subprocess.Popen(cmd + [str(12345 + k), str(w), str(h), str(cfg["stages"]), str(FILT), str(cfg["segments"])], cwd=ROOT,
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
    return {"value": round(len(times) * w * h / slowest / 1e6, 3), "unit": "Mpixels/s", "cores": procs,
            "kind": "reference" if binding.have_reference() else "port",
            "sample": f"{len(times)} of {procs} processes x 1 full {w}x{h} frame in parallel, slowest encode {slowest:.2f} s, all rc 0: {ok}"}


def source_digest():
    """content hash of the kernel sources: a profile taken from other sources is stale"""
    hsh = hashlib.sha256()
    csrc = os.path.join(ROOT, "icer_compression_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if not os.path.isfile(os.path.join(csrc, f)):
            continue
        with open(os.path.join(csrc, f), "rb") as fh:
            hsh.update(f.encode() + b"\0" + fh.read())
    return hsh.hexdigest()[:16]


def frame_goldens(name, rank):
    """(size, crc32) of the reference streams of this rank's frames of configuration `name`"""
    gdir = os.path.join(ROOT, "tests", "golden")
    if name == "C2":
        with open(os.path.join(gdir, "golden.json")) as fh:
            g = json.load(fh)["C2_4096_gray_5st_10seg"]
        return [(g["size"], g["crc32"])]
    with open(os.path.join(gdir, "batch_golden.json")) as fh:
        fr = json.load(fh)[name]["frames"]
    c = CONFIGS[name]
    lo = (rank * c["per_gpu"]) % c["total"]
    return [tuple(fr[(lo + k) % c["total"]]) for k in range(c["per_gpu"])]


class Workload:
    """this rank's frames of one configuration, resident in HBM, with an encoder and output buffers"""

    def __init__(self, name, rank, dev, local_rank):
        import torch
        from icer_compression_amd import api, synth
        self.name, self.cfg = name, CONFIGS[name]
        c = self.cfg
        self.B, self.w, self.h = c["per_gpu"], c["w"], c["h"]
        self.quota = 2 * self.w * self.h
        # frame k of the whole batch uses seed 12345 + k; C2 is the one golden frame on every rank
        first = 0 if name == "C2" else (rank * self.B) % c["total"]
        self.frames = synth.gray_frames_torch(self.B, self.w, self.h, synth.DEFAULT_SEED + first, dev, 1)
        self.out = torch.empty((self.B, self.quota), dtype=torch.uint8, device=dev)
        self.sizes = torch.zeros(self.B, dtype=torch.int64, device=dev)
        self.rcs = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self.enc = api.Encoder(self.w, self.h, 1, c["stages"], FILT, c["segments"], max_frames=self.B, device=local_rank)
        self.gold = frame_goldens(name, rank)

    def step(self):
        self.enc.encode_torch(self.frames, self.quota, self.out, self.sizes, self.rcs)

    def verify(self):
        """every frame of this rank: return code, stream length and CRC-32 equal the reference's"""
        import torch
        torch.cuda.synchronize()
        sizes, rcs = self.sizes.cpu().numpy(), self.rcs.cpu().numpy()
        bad = []
        for k in range(self.B):
            size, crc = self.gold[k]
            ok = int(rcs[k]) == 0 and int(sizes[k]) == size
            if ok:
                ok = ("%08x" % zlib.crc32(self.out[k, :size].cpu().numpy().tobytes())) == crc
            if not ok:
                bad.append(k)
        return bad, int(sizes.sum())

    def close(self):
        self.enc.close()
        del self.frames, self.out


def measure_traffic(args):
    """HBM traffic and issue counters of the dominant kernel, measured on THIS build in child runs under rocprofv3:
    --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, kernel trace only; gfx950 FETCH_SIZE counts half the bytes of a
    streaming read, calibrated on finalize_kernel, see tools/rocprof_summary.py) and the SQ instruction counters."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    import sqlite3
    import tempfile
    res = {}
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--batched-probe", "0",
             "--no-batch-configs", "--no-traffic", "--config", args.config]
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for tag, ctrs in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("sq", ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES"])):
            d = os.path.join(td, tag)
            try:
                subprocess.run([exe, "--kernel-trace", "--pmc"] + ctrs + ["-d", d, "-o", "r", "--"] + child, cwd="/tmp", env=env,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=100)
                db = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
                cur = sqlite3.connect(db[0]).cursor()
                q = ("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%code_units%' "
                     "group by counter_name")
                for n, v, _ in cur.execute(q):
                    res[n] = v
            except Exception as exc:                                   # noqa: BLE001 -- the bench line survives a failed pass
                return None, f"rocprofv3 pass {tag} failed: {exc!r}"
    return res, "measured in this run: rocprofv3 --kernel-trace --pmc passes on a child bench.py (3 steps)"


def run_timed(wl, steps, warmup, barrier, dev):
    import torch
    from icer_compression_amd import shard
    for _ in range(warmup):
        wl.step()
    torch.cuda.synchronize(dev)
    wl.enc.timing_enable(True)
    wl.enc.timing_read(reset=True)
    barrier()
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(steps):
        ts = time.perf_counter()
        wl.step()
        step_ms.append(round((time.perf_counter() - ts) * 1e3, 3))
    barrier()
    elapsed = time.perf_counter() - t0
    stage_ms, calls = wl.enc.timing_read(reset=True)
    wl.enc.timing_enable(False)
    return shard.max_over_ranks(elapsed, dev), step_ms, stage_ms, calls


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=list(CONFIGS), default="C2", help="timed workload (default C2 = BASELINE configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch-configs", action="store_true", help="skip the secondary C4 / C5 figures")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 counter passes (roofline.traffic = null)")
    ap.add_argument("--batched-probe", type=int, default=8,
                    help="also report C2 throughput with this many frames per launch (secondary figure, 0 = skip)")
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

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def all_ranks_ok(ok: bool) -> bool:
        """parity is a property of the whole job: every rank's frames, MIN-reduced"""
        if world == 1:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    cfg = CONFIGS[args.config]
    W, H, B = cfg["w"], cfg["h"], cfg["per_gpu"]
    wl = Workload(args.config, rank, dev, local_rank)

    # parity gate first (one extra untimed encode): EVERY rank checks EVERY one of its frames against the reference
    wl.step()
    bad, bytes_out = wl.verify()
    if not all_ranks_ok(not bad):
        raise SystemExit(f"rank {rank}: frames {bad} are not bit-exact with the reference golden; no number reported")

    elapsed_max, step_ms, stage_ms, calls = run_timed(wl, args.steps, args.warmup, barrier, dev)
    h_sizes_sum = bytes_out
    units_per_frame = wl.enc.info()["units_per_frame"]
    stats = wl.enc.stats()

    # secondary figure: the C2 geometry with several frames per launch; one frame alone cannot fill 256 CUs because its
    # largest coding units form a serial chain
    batched = None
    if args.batched_probe > 1 and args.config == "C2":
        PB = args.batched_probe
        bf = synth.gray_frames_torch(1, W, H, synth.DEFAULT_SEED, dev, 1).repeat(PB, 1, 1).contiguous()
        bout = torch.empty((PB, wl.quota), dtype=torch.uint8, device=dev)
        bsizes = torch.zeros(PB, dtype=torch.int64, device=dev)
        brcs = torch.zeros(PB, dtype=torch.int32, device=dev)
        benc = api.Encoder(W, H, 1, cfg["stages"], FILT, cfg["segments"], max_frames=PB, device=local_rank)
        benc.encode_torch(bf, wl.quota, bout, bsizes, brcs)
        barrier()
        tb = time.perf_counter()
        nb = max(2, args.steps // 2)
        for _ in range(nb):
            benc.encode_torch(bf, wl.quota, bout, bsizes, brcs)
        barrier()
        tb = shard.max_over_ranks(time.perf_counter() - tb, dev)
        g = wl.gold[0]
        ok = bool((brcs.cpu().numpy() == 0).all()) and all(
            int(bsizes[k]) == g[0] and ("%08x" % zlib.crc32(bout[k, : g[0]].cpu().numpy().tobytes())) == g[1] for k in range(PB))
        batched = {"frames_per_gpu_per_launch": PB, "value": round(world * PB * W * H * nb / tb / 1e6, 3), "unit": "Mpixels/s",
                   "ms_per_launch": round(tb / nb * 1e3, 3), "parity": all_ranks_ok(ok), "frames_checked_per_rank": PB}
        benc.close()
        del bf, bout

    # the batch configurations BASELINE.json names for 8 GPUs, as this GPU's share of them
    batch_cfgs = {}
    if not args.no_batch_configs:
        have = os.path.exists(os.path.join(ROOT, "tests", "golden", "batch_golden.json"))
        for name in ("C4", "C5"):
            if name == args.config or not have:
                continue
            try:
                bw = Workload(name, rank, dev, local_rank)
                bw.step()
                badf, out_bytes = bw.verify()
                ok = all_ranks_ok(not badf)
                nst = 3
                t_el, _, st_ms, cl = run_timed(bw, nst, 1, barrier, dev)
                c = bw.cfg
                pix = world * c["per_gpu"] * c["w"] * c["h"] * nst
                batch_cfgs[name] = {
                    "workload": f"{c['what']}: {c['total']} x {c['w']}x{c['h']}, {c['stages']} stages, {c['segments']} segments, lossless; "
                                f"{c['per_gpu']} frames per GPU per launch (frames {c['per_gpu']}*rank ..)",
                    "value": round(pix / t_el / 1e6, 3), "unit": "Mpixels/s", "ms_per_launch": round(t_el / nst * 1e3, 3), "launches": nst,
                    "n_gpus": world, "parity": ok, "frames_checked_per_rank": c["per_gpu"],
                    "parity_note": "every frame of every rank: rc, length and CRC-32 equal the reference CPU encoder's (tests/golden/batch_golden.json)",
                    "bytes_out_rank0": out_bytes, "code_units_ms": round(st_ms["code_units"] / max(cl, 1), 3),
                    "coder": "code_units_kernel<8> for the dense coding units; the all-but-blank ones (>= 95 % blank chunks, listed on the "
                             "device by route_units_kernel) by code_units_wgs_list_kernel on a second stream beside it"}
                if rank == 0 and world == 1:
                    ab = float(c["per_gpu"] * c["w"] * c["h"] * 2 + out_bytes)
                    kms = st_ms["code_units"] / max(cl, 1)
                    batch_cfgs[name]["roofline_frac"] = round(ab / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6)
                bw.close()
                del bw
                torch.cuda.empty_cache()
            except Exception as exc:                                   # noqa: BLE001 -- secondary figure
                batch_cfgs[name] = {"error": repr(exc)}

    if rank == 0:
        n_pix = world * B * W * H * args.steps
        value = n_pix / elapsed_max / 1e6
        # roofline of the dominant kernel (the coding-unit kernel): algorithmic bytes per launch =
        # SURVEY 8(d) per-frame figure (input planes read once + final stream written once) x frames per launch
        k_ms = stage_ms["code_units"] / max(calls, 1)
        alg_bytes = float(B * W * H * 2 + h_sizes_sum)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        pipe = "code_units_kernel<11> (wave pipeline, 11-wave workgroups)" if B == 1 else "code_units_kernel<8> (wave pipeline, 8-wave workgroups)"
        kernel = {0: pipe + "; code_units_wg_kernel in progressive mode", 1: pipe, 2: "code_units_wg_kernel"}[stats["coder_mode"]]
        line = {
            "metric": "Mpixels/s encode (bit-exact), 4096x4096 gray", "value": round(value, 3), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": f"{cfg['what']}: {B} x {W}x{H} 8-bit gray (uint16 API) per GPU per step, "
                                   f"{cfg['stages']} DWT stages, filter A, {cfg['segments']} segments, lossless quota 2*W*H; input and "
                                   "output stream resident in HBM", "frames_per_gpu_per_step": B,
                       "units_per_frame": units_per_frame,
                       "parity": "every frame of every rank: rc, stream length and CRC-32 equal the reference golden (checked before timing)"},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(k_ms, 4)},
            "stage_ms_per_step": {k: round(v / max(calls, 1), 4) for k, v in stage_ms.items()},
            "step_ms": step_ms,
            "coder_events": {k: stats[k] for k in ("unit_timeouts", "fallback_batches", "slot_retries")},
        }
        if batched:
            line["batched"] = batched
        if batch_cfgs:
            line["batch_configs"] = batch_cfgs
        if world == 1 and not args.no_traffic:
            ctr, src = measure_traffic(args)
            line["roofline"]["traffic_source"] = src + f"; sources {source_digest()}"
            if ctr and "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
                line["roofline"]["traffic"] = int(2 * ctr["FETCH_SIZE"] * 1024 + ctr["WRITE_SIZE"] * 1024)
                line["roofline"]["traffic_counters_KiB"] = {"FETCH_SIZE": round(ctr["FETCH_SIZE"], 1), "WRITE_SIZE": round(ctr["WRITE_SIZE"], 1)}
            if ctr and "SQ_INSTS_VALU" in ctr:
                # issue roofline: a wave64 VALU instruction occupies one of the chip's 1024 SIMDs for 4 cycles
                cyc = k_ms * 1e-3 * 2.4e9
                line["roofline"]["issue"] = {
                    "valu_wave_insts_per_launch": int(ctr["SQ_INSTS_VALU"]), "salu_insts_per_launch": int(ctr.get("SQ_INSTS_SALU", 0)),
                    "lds_insts_per_launch": int(ctr.get("SQ_INSTS_LDS", 0)),
                    "valu_busy_frac": round(ctr["SQ_INSTS_VALU"] * 4 / (CUS * SIMDS_PER_CU * cyc), 4),
                    "note": "valu_busy_frac = VALU wave-instructions x 4 cycles / (1024 SIMDs x kernel cycles at 2.4 GHz)"}
        if world == 1 and args.config == "C2":
            # PCIe-inclusive figures of the host-buffer entry point (never `value`): H2D frame + kernels + D2H stream,
            # caller buffers pageable (the runtime stages them) and page-locked (icerx_pin_host: DMA at link speed)
            host_frame = np.ascontiguousarray(wl.frames[:1].cpu().numpy().view(np.uint16))
            h_out = np.zeros((1, wl.quota), np.uint8)
            h_sizes, h_rcs = np.zeros(1, np.uint64), np.zeros(1, np.int32)
            hb = {}
            for mode in ("pageable", "pinned"):
                pinned = mode == "pinned" and api.pin_host(host_frame) and api.pin_host(h_out)
                if mode == "pinned" and not pinned:
                    continue
                wl.enc.encode_host_into(host_frame, wl.quota, h_out, h_sizes, h_rcs)
                t0h = time.perf_counter()
                reps = 5
                for _ in range(reps):
                    wl.enc.encode_host_into(host_frame, wl.quota, h_out, h_sizes, h_rcs)
                th = (time.perf_counter() - t0h) / reps
                okh = int(h_rcs[0]) == 0 and ("%08x" % zlib.crc32(h_out[0, : int(h_sizes[0])].tobytes())) == wl.gold[0][1]
                hb[mode] = {"ms_per_frame": round(th * 1e3, 3), "value": round(W * H / th / 1e6, 3), "parity": okh}
                if pinned:
                    api.unpin_host(host_frame); api.unpin_host(h_out)
            best = hb.get("pinned", hb["pageable"])
            line["host_buffers"] = {"ms_per_frame": best["ms_per_frame"], "value": best["value"], "unit": "Mpixels/s", "parity": all(v["parity"] for v in hb.values()),
                                    "note": "icerx_encode_host: H2D of the frame, all kernels, D2H of size/rc/stream; caller buffers page-locked with "
                                            "icerx_pin_host when available", "by_caller_memory": hb}
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(host_frame[0], wl.gold[0][1], cfg)
                line["speedup_vs_cpu_1thread"] = round(value / line["cpu_baseline"]["value"], 2)
                try:
                    line["cpu_all_cores"] = cpu_all_cores(cfg)
                except Exception as exc:                                   # reported, never fatal for the GPU numbers
                    line["cpu_all_cores"] = {"error": repr(exc)}
        print(json.dumps(line), flush=True)
    wl.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
