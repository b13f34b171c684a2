#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ICER encoder.

Metric (BASELINE.json): Mpixels/s encode, bit-exact, 4096x4096 gray (+ achieved HBM GB/s of the
dominant kernel against the 8 TB/s roofline).  Workload = BASELINE.json configs[1] ("C2"): one 4096x4096
8-bit synthetic frame (widened to uint16 like the reference's own callers), 5 DWT stages, filter A,
10 error-containment segments, lossless byte quota (2*W*H).  A "step" is one pass of the whole encode
hot path (DWT -> LL mean -> sign-magnitude -> all coding units -> quota scan -> final stream) over one
batch (default: ONE frame, as the config says) whose input already sits in HBM; the final stream stays
in HBM.  EVERY rank checks its streams against the reference goldens before anything is timed AND again after the timed loop.

--gpus N > 1 (started plainly, bench.py launches its own N ranks; under torch.distributed.run it is a rank): the timed workload is
BASELINE configs[3] ("C4"), the WHOLE batch of 256 x 2048x2048 frames split over the ranks (strong scaling; no collective on the
data path), a rank's launches of 32 frames coded with two launches in flight (within a step when a rank has several, across steps
when it has one: step k is submitted before step k - 1 is waited for, everything complete before the closing barrier); value = 256
frames' pixels x steps / max-over-ranks time.  Its N = 1 point is `scaling_reference` of the --gpus 1 line (whose own `value` is the lone C2 frame).  --config C2 --gpus N:
every rank its own copy of the C2 frame (weak).

Secondary objects of the same line (none of them is `value`):
  batched        C2 geometry, 8 frames per launch
  batch_configs  this GPU's share of the two batch configurations BASELINE.json names for 8 GPUs: "C4" = 256 x 2048x2048
                 (32 frames per GPU, frames 32*rank ..), "C5" = 64 x 8192x8192 (8 per GPU); every frame of every rank is
                 checked against tests/golden/batch_golden.json (reference CPU encoder)
  batch_host     the same shares fed from (page-locked) HOST memory through icerx_compress_batch_uint16_devices: upload,
                 kernels and download of the streams overlapped in sub-batches -- the one resource N GPUs of a node share
  C3             BASELINE configs[2]: 4096x4096 YUV, byte quota 70 000 (progressive early stop)
  decode         the C2 stream back through libicer_hip_dec.so, with the reference decoder on one core;
                 batch_configs[*].decode: this rank's C4 / C5 streams back through it in one call, every frame compared
  dropin         the lib_icer entry points themselves as a reference user calls them (example/src/example_encode.c:36-77):
                 icer_compress_image_uint16 on the C2 frame and icer_compress_image_yuv_uint16 on C3, pageable caller
                 memory, coefficient write-back included
  host_buffers, cpu_baseline, cpu_all_cores, roofline.traffic / roofline.issue (child rocprofv3 passes)
  scaling_reference  (N = 1) all 256 C4 frames on this GPU per step: the N = 1 point of the N > 1 headline
  c2_per_rank        (N > 1) the lone C2 frame on every rank at once
  one_process        icerx_compress_batch_uint16(n_gpus = 0): all 256 C4 frames from page-locked host memory over every visible
                     device from ONE process (run in a child process)
  batch_host[*].crowded_process / .runtime_default_hw_queues   the host-fed call beside live torch streams and a decoder; and in
                     a child process that has not asked for more hardware queues

--config C4|C5 makes one of the batch configurations the timed workload; with --sweep it is run as every rank of an
8-GPU job in turn on this one GPU (all 256 / 64 frames, each against its reference golden); with --scaling strong the
whole batch is split over the ranks (N = 1: all of it on one GPU); --source host times the host-fed path instead.

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
# (before the HIP runtime starts: the host-fed batch runs three encoder streams + their side streams; the runtime's default
# of 4 hardware queues makes some of them take turns.  A process-wide choice, so the benchmark makes it, not the library.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

FILT = 0
CONFIGS = {
    # name: geometry, frames per GPU per step, frames of the whole (8-GPU) batch
    "C2": dict(w=4096, h=4096, stages=5, segments=10, per_gpu=1, total=1, what="BASELINE configs[1]"),
    "C4": dict(w=2048, h=2048, stages=4, segments=16, per_gpu=32, total=256, what="BASELINE configs[3]"),
    "C5": dict(w=8192, h=8192, stages=6, segments=32, per_gpu=8, total=64, what="BASELINE configs[4]"),
}
HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
CUS, SIMDS_PER_CU = 256, 4
CLOCK_HZ = 2.4e9
# Issue cost of a wave64 instruction on one SIMD, MEASURED on this chip by tools/valu_rate_ubench.py
# (profiles/r03_valu_rate_ubench.md): cycles a SIMD is occupied per wave-instruction of the class when enough waves are
# resident to saturate it.  The coder's VALU mix is 32-bit integer ALU (v_and / v_lshl / v_cndmask / v_bfe / v_mbcnt / v_add).
ISSUE_CYCLES = {"valu_int32": None, "source": "profiles/r03_valu_rate_ubench.md"}
try:
    with open(os.path.join(ROOT, "profiles", "r03_valu_rate_ubench.json")) as _fh:
        ISSUE_CYCLES.update(json.load(_fh).get("bench_constants", {}))
except Exception:                                                          # noqa: BLE001 -- not measured yet
    pass


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


def frame_goldens(name, rank, first=None, count=None):
    """(size, crc32) of the reference streams of this rank's frames of configuration `name` (weak scaling: frames
    per_gpu * rank ..; `first` / `count` name another block of the batch)"""
    gdir = os.path.join(ROOT, "tests", "golden")
    if name == "C2":
        with open(os.path.join(gdir, "golden.json")) as fh:
            g = json.load(fh)["C2_4096_gray_5st_10seg"]
        return [(g["size"], g["crc32"])]
    with open(os.path.join(gdir, "batch_golden.json")) as fh:
        fr = json.load(fh)[name]["frames"]
    c = CONFIGS[name]
    lo = (rank * c["per_gpu"]) % c["total"] if first is None else first
    n = c["per_gpu"] if count is None else count
    return [tuple(fr[(lo + k) % c["total"]]) for k in range(n)]


class Workload:
    """a block of frames of one configuration, resident in HBM, with an encoder and output buffers; coded in launches of
    at most `per_gpu` frames"""

    def __init__(self, name, rank, dev, local_rank, first=None, count=None, in_flight=1):
        import torch
        from icer_compression_amd import api, synth
        self.name, self.cfg = name, CONFIGS[name]
        c = self.cfg
        self.w, self.h = c["w"], c["h"]
        self.quota = 2 * self.w * self.h
        # frame k of the whole batch uses seed 12345 + k; C2 is the one golden frame on every rank
        self.first = (0 if name == "C2" else (rank * c["per_gpu"]) % c["total"]) if first is None else first
        self.B = c["per_gpu"] if count is None else count
        self.launch = min(self.B, c["per_gpu"])
        self.frames = synth.gray_frames_torch(self.B, self.w, self.h, synth.DEFAULT_SEED + self.first, dev, 1)
        self.out = torch.empty((self.B, self.quota), dtype=torch.uint8, device=dev)
        self.sizes = torch.zeros(self.B, dtype=torch.int64, device=dev)
        self.rcs = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self.enc = api.Encoder(self.w, self.h, 1, c["stages"], FILT, c["segments"], max_frames=self.launch, device=local_rank)
        self.gold = frame_goldens(name, rank, self.first, self.B)
        # A block of several launches (strong scaling below 8 ranks, the N = 1 reference of that curve) is coded the way a caller
        # with a queue of batches does it: two encoders taking the launches in turns through the asynchronous half of the API, a
        # stream each, two launches in flight -- the transform and the tail of one launch hide behind the other's coding units.
        self.pipelined = self.B > self.launch
        # ... and a block that is ONE launch per step (the strong-scaling headline at 8 ranks) the same way ACROSS steps when asked to
        # (in_flight = 2): step k is submitted before step k - 1 is waited for, into a second set of output buffers; finish() waits for
        # the last one -- run_timed calls it before the closing barrier, verify() before it looks at anything
        self.cross_step = in_flight == 2 and not self.pipelined and name != "C2"
        self._prev, self._k, self.out2 = None, 0, None
        if self.cross_step:
            self.out2 = torch.empty((self.B, self.quota), dtype=torch.uint8, device=dev)
            self.sizes2 = torch.zeros(self.B, dtype=torch.int64, device=dev)
            self.rcs2 = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self.enc2, self._hip, self._streams = None, None, []
        if self.pipelined or self.cross_step:
            import ctypes
            self.enc2 = api.Encoder(self.w, self.h, 1, c["stages"], FILT, c["segments"], max_frames=self.launch, device=local_rank)
            self._hip = ctypes.CDLL("libamdhip64.so")
            self._hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
            self._hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
            for _ in range(2):                                          # (plain HIP streams: torch.cuda.Stream() would take 64 from torch's pool)
                st_ = ctypes.c_void_p()
                if self._hip.hipStreamCreateWithFlags(ctypes.byref(st_), 1) != 0:      # hipStreamNonBlocking
                    raise RuntimeError("hipStreamCreateWithFlags failed")
                self._streams.append(st_)
        torch.cuda.synchronize(dev)                                     # (the frames were made on torch's stream)

    def finish(self):
        if self._prev is not None:
            self._prev.wait()
            self._prev = None

    def step(self):
        if self.cross_step:
            k = self._k & 1
            e = (self.enc, self.enc2)[k]
            out, sizes, rcs = ((self.out, self.sizes, self.rcs), (self.out2, self.sizes2, self.rcs2))[k]
            e.encode_device_async_ptrs(self.frames.data_ptr(), self.B, self.quota, out.data_ptr(), out.stride(0), sizes.data_ptr(), rcs.data_ptr(),
                                       self._streams[k].value)
            if self._prev is not None:
                self._prev.wait()
            self._prev = e
            self._k += 1
            return
        if self.pipelined:
            encs, prev = (self.enc, self.enc2), None
            for k, lo in enumerate(range(0, self.B, self.launch)):
                hi = min(lo + self.launch, self.B)
                e = encs[k & 1]
                e.encode_device_async_ptrs(self.frames[lo:hi].data_ptr(), hi - lo, self.quota, self.out[lo:hi].data_ptr(), self.out.stride(0),
                                           self.sizes[lo:hi].data_ptr(), self.rcs[lo:hi].data_ptr(), self._streams[k & 1].value)
                if prev is not None:
                    prev.wait()                                         # launch k - 1, after launch k has been submitted
                prev = e
            prev.wait()
            return
        for lo in range(0, self.B, self.launch):
            hi = min(lo + self.launch, self.B)
            self.enc.encode_torch(self.frames[lo:hi], self.quota, self.out[lo:hi], self.sizes[lo:hi], self.rcs[lo:hi])

    def verify(self):
        """every frame of this block: return code, stream length and CRC-32 equal the reference's"""
        import torch
        self.finish()
        torch.cuda.synchronize()
        sizes, rcs = self.sizes.cpu().numpy(), self.rcs.cpu().numpy()
        bad = []
        sets = [(self.out, sizes, rcs)]
        if self.cross_step and self._k >= 2:                        # (the second set of buffers has been written, too)
            sets.append((self.out2, self.sizes2.cpu().numpy(), self.rcs2.cpu().numpy()))
        for out, sz, rc in sets:
            for k in range(self.B):
                size, crc = self.gold[k]
                ok = int(rc[k]) == 0 and int(sz[k]) == size
                if ok:
                    ok = ("%08x" % zlib.crc32(out[k, :size].cpu().numpy().tobytes())) == crc
                if not ok and self.first + k not in bad:
                    bad.append(self.first + k)
        return bad, int(sizes.sum())

    def close(self):
        self.enc.close()
        if self.enc2 is not None:
            self.enc2.close()
        for st_ in self._streams:
            self._hip.hipStreamDestroy(st_)
        self._streams = []
        self.out2 = None
        del self.frames, self.out


class HostWorkload:
    """the same block of frames in page-locked HOST memory, coded by icerx_compress_batch_uint16_devices on this rank's
    device: copy-in, kernels and copy-out of sub-batches overlap on three streams (csrc/api.hip)"""

    def __init__(self, name, rank, dev, local_rank, first=None, count=None, devices="own"):
        import torch
        from icer_compression_amd import api, synth
        self.api = api
        self.name, self.cfg = name, CONFIGS[name]
        c = self.cfg
        self.w, self.h, self.local_rank = c["w"], c["h"], local_rank
        self.devices = [local_rank] if devices == "own" else devices      # None = icerx_compress_batch_uint16(n_gpus = 0): every visible device
        self.quota = 2 * self.w * self.h
        self.first = (0 if name == "C2" else (rank * c["per_gpu"]) % c["total"]) if first is None else first
        self.B = c["per_gpu"] if count is None else count
        self.frames = np.empty((self.B, self.h, self.w), np.uint16)
        for lo in range(0, self.B, 8):                                          # generated on the device, kept on the host
            hi = min(lo + 8, self.B)
            d = synth.gray_frames_torch(hi - lo, self.w, self.h, synth.DEFAULT_SEED + self.first + lo, dev, 1)
            self.frames[lo:hi] = d.cpu().numpy().view(np.uint16)
            del d
        torch.cuda.empty_cache()
        # rows of a lossless stream's worst case (bytes <= samples here: 8-bit content costs ~0.6 B per pixel)
        self.stride = self.w * self.h
        self.out = np.zeros((self.B, self.stride), np.uint8)
        self.sizes, self.rcs = np.zeros(self.B, np.uint64), np.zeros(self.B, np.int32)
        self.pinned = bool(api.pin_host(self.frames)) and bool(api.pin_host(self.out))
        self.gold = frame_goldens(name, rank, self.first, self.B)
        self.enc = None

    def step(self):
        c = self.cfg
        rc = self.api.compress_batch(self.frames, c["stages"], FILT, c["segments"], self.quota, self.out, self.sizes, self.rcs, devices=self.devices)
        if rc != 0:
            raise RuntimeError(f"icerx_compress_batch_uint16_devices rc={rc}: {self.api.load_library().icerx_last_error().decode()}")

    def verify(self):
        bad = []
        for k in range(self.B):
            size, crc = self.gold[k]
            ok = int(self.rcs[k]) == 0 and int(self.sizes[k]) == size and ("%08x" % zlib.crc32(self.out[k, :size].tobytes())) == crc
            if not ok:
                bad.append(self.first + k)
        return bad, int(self.sizes.sum())

    def close(self):
        if self.pinned:
            self.api.unpin_host(self.frames); self.api.unpin_host(self.out)
        self.api.load_library().icerx_batch_release()
        del self.frames, self.out


PMC_PASSES = (("stats", None), ("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
              ("sq", ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES"]),
              # hardware's own account of the waves' time (quad-cycles, MI355X_MICROARCH.md "rocprofv3 PMC slots"): no
              # cycles-per-instruction assumption needed
              ("sqw", ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INST_CYCLES_SALU"]),
              ("lds", ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVES", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]))


def measure_traffic(args, config=None, passes=None):
    """HBM traffic and issue counters of the dominant kernel, measured on THIS build in child runs under rocprofv3:
    --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, kernel trace only; gfx950 FETCH_SIZE counts half the bytes of a
    streaming read, calibrated on finalize_kernel, see tools/rocprof_summary.py) and the SQ counters.  A pass that fails
    (a counter this rocprofv3 does not know) is reported and skipped; the others still count."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    import sqlite3
    import tempfile
    res, failed, per_kernel = {}, [], {}
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--batched-probe", "0",
             "--no-batch-configs", "--no-traffic", "--no-extras", "--no-one-process", "--config", config or args.config]
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for tag, ctrs in PMC_PASSES:
            if passes and tag not in passes:
                continue
            d = os.path.join(td, tag)
            try:
                # (counter passes: --kernel-trace + --pmc only; the duration pass: --kernel-trace --stats, no counters)
                what = ["--kernel-trace", "--stats"] if ctrs is None else ["--kernel-trace", "--pmc"] + ctrs
                subprocess.run([exe] + what + ["-d", d, "-o", "r", "--"] + child, cwd="/tmp", env=env,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=150)
                db = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
                cur = sqlite3.connect(db[0]).cursor()
                if ctrs is None:
                    # kernel durations without counters attached (what `rocprofv3 --kernel-trace --stats` of the same command reports)
                    for k, calls, avg in cur.execute("select name, total_calls, average from top_kernels"):
                        per_kernel.setdefault(short_kernel(k), {})["avg_us"] = avg
                    continue
                q = ("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%code_units_kernel%' "
                     "group by counter_name")
                for n, v, _ in cur.execute(q):
                    res[n] = v
                if tag in ("fetch", "write"):
                    # every kernel of the launch: the counter summed over its dispatches (a DWT stage, a part of a batch: several per call)
                    q = "select kernel_name, sum(value), count(*) from counters_collection where counter_name=? group by kernel_name"
                    for k, v, cnt in cur.execute(q, (ctrs[0],)):
                        e = per_kernel.setdefault(short_kernel(k), {})
                        e[ctrs[0] + "_KiB_sum"] = v
                        e["dispatches"] = cnt
            except Exception as exc:                                   # noqa: BLE001 -- the bench line survives a failed pass
                failed.append(f"{tag}: {exc!r}")
    note = "measured in this run: rocprofv3 --kernel-trace --pmc passes on a child bench.py (3 steps)"
    if failed:
        note += "; failed passes: " + "; ".join(failed)
    res["per_kernel"] = per_kernel
    return res, note


def short_kernel(name: str) -> str:
    n = name.replace("(anonymous namespace)::", "").split("(")[0]
    head = n.split("<")[0]
    return head.split("::")[-1].replace("void ", "") + n[len(head):].replace("icer::", "")


# FETCH_SIZE on gfx950 counts a 128-byte request as 64 bytes (MI355X_MICROARCH.md, HBM): x 2 for reads that arrive as 128-byte requests,
# x 1 for reads that arrive as 64-byte ones.  Per kernel, calibrated on a known byte count in that kernel's own access pattern:
#   finalize_ll_kernel   2.0   reads and writes exactly one LL rectangle: WRITE_SIZE exact, FETCH_SIZE 0.50 x (round 1)
#   dwt_tile_kernel      1.0   stage 0 of a 4096^2 frame reads 33.5 MB + halo once: FETCH_SIZE uncorrected = 43 MiB over the five stages
#                              = the compulsory 42.7 MiB (the blanket x 2 over-counted it)
#   family_events_kernel 1.0   2-byte gathers of a 3x3 window, one 64-byte line per wave and row
#   everything else      2.0   the guide's figure (an upper bound where the pattern is not a streaming one)
FETCH_FACTOR = {"dwt_tile_kernel": 1.0, "family_events_kernel": 1.0}


def launch_traffic(per_kernel, parts_per_call):
    """HBM bytes of ONE call summed over every kernel it launches (ours only: the bench's own torch kernels are not the call's)"""
    ours = ("dwt_tile", "ll_sum", "ll_mean", "frame_status", "finalize_ll", "family_events", "route_units", "code_units", "splice_units", "scan_kernel", "gather_kernel")
    scan = per_kernel.get("scan_kernel", {}).get("dispatches", 0)
    calls = scan / max(parts_per_call, 1)
    if calls <= 0:
        return None, {}
    table, total = {}, 0.0
    for k, e in sorted(per_kernel.items()):
        if not k.startswith(ours) or "FETCH_SIZE_KiB_sum" not in e or "WRITE_SIZE_KiB_sum" not in e:
            continue
        f = FETCH_FACTOR.get(k.split("<")[0], 2.0)
        b = (f * e["FETCH_SIZE_KiB_sum"] + e["WRITE_SIZE_KiB_sum"]) * 1024.0 / calls
        total += b
        table[k] = {"dispatches_per_call": round(e["dispatches"] / calls, 2), "FETCH_SIZE_KiB_per_call": round(e["FETCH_SIZE_KiB_sum"] / calls, 1),
                    "WRITE_SIZE_KiB_per_call": round(e["WRITE_SIZE_KiB_sum"] / calls, 1), "fetch_factor": f, "bytes_per_call": int(b)}
        if "avg_us" in e:
            table[k]["avg_us_under_rocprof"] = round(e["avg_us"], 2)
    return int(total), table


def run_timed(wl, steps, warmup, barrier, dev, red_dev=None):
    import torch
    from icer_compression_amd import shard
    for _ in range(warmup):
        wl.step()
    if hasattr(wl, "finish"):
        wl.finish()
    torch.cuda.synchronize(dev)
    if wl.enc is not None:
        wl.enc.timing_enable(True)
        wl.enc.timing_read(reset=True)
    import gc
    gc.collect()
    gc.disable()                                                      # (no collector pause inside the timed region)
    try:
        barrier()
        t0 = time.perf_counter()
        step_ms = []
        for _ in range(steps):
            ts = time.perf_counter()
            wl.step()
            step_ms.append(round((time.perf_counter() - ts) * 1e3, 3))
        if hasattr(wl, "finish"):
            wl.finish()                                               # (a step still in flight belongs to the timed region)
        barrier()
        elapsed = time.perf_counter() - t0
    finally:
        gc.enable()                                                   # (also when a step raises: the secondary objects catch and carry on)
    stage_ms, calls = ({}, 0)
    if wl.enc is not None:
        stage_ms, calls = wl.enc.timing_read(reset=True)
        wl.enc.timing_enable(False)
    return shard.max_over_ranks(elapsed, red_dev if red_dev is not None else dev), step_ms, stage_ms, calls


def pingpong_object(bw, name, rank, dev, local_rank, barrier, all_ranks_ok, world, red_dev=None, n=6):
    """The rank's share of a batch configuration and the NEXT share of it, coded by two encoders in turns through the
    asynchronous half of the API (icerx_encode_device_async on a stream each; the host waits for launch k - 1 after it has
    submitted launch k): the tail of a launch -- its last long coding units, with most of the chip idle -- hides behind the
    start of the next.  Same kernels, same launches; what a caller with a queue of batches does."""
    import torch
    from icer_compression_amd import shard
    c = bw.cfg
    other = Workload(name, rank, dev, local_rank, first=(bw.first + c["per_gpu"]) % c["total"])
    wls = [bw, other]
    # two plain HIP streams, made here and destroyed below: torch.cuda.Stream() would take them from torch's pool, whose first use
    # creates 64 streams that live as long as the process -- and the hardware queues (GPU_MAX_HW_QUEUES) are shared out over all
    # live streams: legs that follow (the host-fed pipeline overlaps five) then find theirs doubled up (measured: 0.66 x)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    streams = [ctypes.c_void_p() for _ in range(2)]
    for st_ in streams:
        if hip.hipStreamCreateWithFlags(ctypes.byref(st_), 1) != 0:             # hipStreamNonBlocking
            raise RuntimeError("hipStreamCreateWithFlags failed")
    other.step()
    bad0, _ = other.verify()
    for w in wls:
        w.sizes.zero_()

    def launch(i):
        w = wls[i]
        w.enc.encode_device_async_ptrs(w.frames.data_ptr(), w.B, w.quota, w.out.data_ptr(), w.out.stride(0), w.sizes.data_ptr(), w.rcs.data_ptr(),
                                       streams[i].value)

    def run(n):
        launch(0)
        for k in range(1, n):
            launch(k & 1)
            wls[(k - 1) & 1].enc.wait()
        wls[(n - 1) & 1].enc.wait()
    run(2)
    torch.cuda.synchronize(dev)
    barrier()
    t0 = time.perf_counter()
    run(n)
    torch.cuda.synchronize(dev)
    barrier()
    t_el = shard.max_over_ranks(time.perf_counter() - t0, red_dev if red_dev is not None else dev)
    bad = [w.verify()[0] for w in wls]
    ok = all_ranks_ok(not bad0 and not bad[0] and not bad[1])
    other.close()
    for st_ in streams:
        hip.hipStreamDestroy(st_)
    return {"value": round(world * n * c["per_gpu"] * c["w"] * c["h"] / t_el / 1e6, 3), "unit": "Mpixels/s", "ms_per_launch": round(t_el / n * 1e3, 3), "launches": n,
            "parity": ok, "frames_checked_per_rank": 2 * c["per_gpu"],
            "note": "two encoders, two streams, icerx_encode_device_async / icerx_encoder_wait; every frame of both shares equals the reference's after the timed launches"}


def c3_object(dev, local_rank, barrier, all_ranks_ok, world, red_dev=None):
    """BASELINE configs[2]: one 4096x4096 YUV frame, 5 stages, 10 segments, byte quota 70 000 (the stream keeps the
    highest-priority packets up to the quota: icer_color.c:343-530), device-resident, against the reference golden"""
    import torch
    from icer_compression_amd import api, shard, synth
    w = h = 4096
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as fh:
        g = json.load(fh)["C3_4096_yuv_quota70000"]
    planes = np.stack(synth.color_frame_yuv(w, h, synth.DEFAULT_SEED))[None]
    d = torch.from_numpy(planes.view(np.int16)).to(dev)
    out = torch.empty((1, g["quota"] + 64), dtype=torch.uint8, device=dev)
    sizes = torch.zeros(1, dtype=torch.int64, device=dev)
    rcs = torch.zeros(1, dtype=torch.int32, device=dev)
    enc = api.Encoder(w, h, 3, g["stages"], FILT, g["segments"], max_frames=1, device=local_rank)

    def check():
        torch.cuda.synchronize()
        s = out[0, : int(sizes[0])].cpu().numpy().tobytes()
        return int(rcs[0]) == g["rc"] and len(s) == g["size"] and "%08x" % zlib.crc32(s) == g["crc32"]
    enc.encode_torch(d, g["quota"], out, sizes, rcs)
    ok = check()
    enc.timing_enable(True)
    enc.timing_read(reset=True)
    barrier()
    t = time.perf_counter()
    n = 10
    for _ in range(n):
        enc.encode_torch(d, g["quota"], out, sizes, rcs)
    barrier()
    el = shard.max_over_ranks(time.perf_counter() - t, red_dev if red_dev is not None else dev)
    st, calls = enc.timing_read(reset=True)
    ok = ok and check()
    mode = enc.stats()["coder_mode"]
    enc.close()
    return {"workload": "BASELINE configs[2]: 1 x 4096x4096 YUV (3 planes, uint16 API), 5 stages, filter A, 10 segments, byte quota 70 000 "
                        "(rc = ICER_BYTE_QUOTA_EXCEEDED, 69 982 bytes); input and stream resident in HBM",
            "value": round(world * w * h * n / el / 1e6, 3), "unit": "Mpixels/s (pixels = W*H, not x channels)", "ms_per_step": round(el / n * 1e3, 4), "steps": n,
            "parity": all_ranks_ok(ok), "parity_note": "rc, length and CRC-32 equal the reference golden, before and after the timed loop",
            "stage_ms_per_step": {k: round(v / max(calls, 1), 4) for k, v in st.items()},
            "coder": "code_units_wg_kernel<WgFour> (progressive mode: four-wave workgroups, units in priority order, early stop, blank runs in closed form)" if mode == 0 else f"coder_mode {mode}",
            "roofline_frac": round((w * h * 2 * 3 + g["size"]) / (st["code_units"] / max(calls, 1) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6) if st.get("code_units") else None}


def decode_object(stream_dev, size, frame_dev, cfg, with_cpu):
    """the C2 stream (already in HBM: the encoder's output) -> uint16 planes in HBM through
    libicer_hip_dec.so (icerx_decode_device, whole call); 1, 8, 16 and 64 streams per call; every decoded frame equals
    the encoder's input; the reference decoder (oracle/_ref) on one host core on the same stream beside it"""
    import torch
    from icer_compression_amd import decoder
    W, H = cfg["w"], cfg["h"]
    dev = stream_dev.device

    def run(n, reps):
        d_data = stream_dev[:size].repeat(n).contiguous()
        d_out = torch.zeros((n, H * W), dtype=torch.int16, device=dev)
        dec = decoder.Decoder(1, cfg["stages"], FILT, cfg["segments"])
        offs, lens = [k * size for k in range(n)], [size] * n
        times = []
        for _ in range(reps + 1):                                # (first call: allocations)
            torch.cuda.synchronize()
            t = time.perf_counter()
            rc2, rcs, _, _ = dec.decode_device(n, d_data.data_ptr(), offs, lens, d_out.data_ptr(), W * H)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t)
        ok = rc2 == 0 and all(r == 0 for r in rcs) and all(bool(torch.equal(d_out[k].view(H, W), frame_dev)) for k in range(n))
        dec.close()
        return min(times[1:]), ok
    t1, ok1 = run(1, 2)
    tb, okb = run(16, 2)
    t8, ok8 = run(8, 1)
    t64, ok64 = run(64, 1)
    okb = okb and ok8 and ok64
    alg = float(size + W * H * 2)
    obj = {"metric": "Mpixels/s decode (bit-exact)", "workload": f"the stream of the timed workload ({size} bytes = the reference golden) resident in HBM -> uint16 planes "
           "in HBM; icerx_decode_device, whole call incl. the packet walk", "value": round(W * H / t1 / 1e6, 2), "unit": "Mpixels/s",
           "ms_per_frame": round(t1 * 1e3, 2), "parity": bool(ok1 and okb), "parity_note": "every decoded frame equals the encoder's input",
           "streams_8_per_call": {"value": round(8 * W * H / t8 / 1e6, 2), "ms_per_call": round(t8 * 1e3, 2)},
           "streams_16_per_call": {"value": round(16 * W * H / tb / 1e6, 2), "ms_per_call": round(tb * 1e3, 2)},
           "streams_64_per_call": {"value": round(64 * W * H / t64 / 1e6, 2), "ms_per_call": round(t64 * 1e3, 2),
                                   "note": "chains of all streams launched longest first; from 20 streams per call on the lane-per-plane kernel, once per size class of row ring"},
           "roofline_frac": round(alg / t1 / 1e9 / HBM_PEAK_GBPS, 7)}
    if with_cpu:
        try:
            from oracle import binding
            binding.build()
            stream = stream_dev[:size].cpu().numpy().tobytes()
            img = frame_dev.cpu().numpy().view(np.uint16)
            if binding.have_reference():
                ref, kind = binding.Reference(), "reference"
                t = time.perf_counter()
                out = ref.decompress(stream, 1, cfg["stages"], FILT, cfg["segments"])
                dt = time.perf_counter() - t
                good = out[0] == 0 and np.array_equal(np.asarray(out[1][0]).reshape(H, W), img)
            else:
                ref, kind = binding.Oracle(), "port"
                t = time.perf_counter()
                out = ref.decompress(stream, 1, cfg["stages"], FILT, cfg["segments"], bufsize=W * H)
                dt = time.perf_counter() - t
                good = out[0] == 0 and np.array_equal(np.asarray(out[3][0]).reshape(H, W), img)
            obj["cpu_baseline"] = {"value": round(W * H / dt / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": kind,
                                   "sample": f"the same stream, one full frame, {dt:.2f} s, decoded image equals the input: {bool(good)}"}
        except Exception as exc:                                   # noqa: BLE001
            obj["cpu_baseline"] = {"error": repr(exc)}
    return obj


def dropin_object(host_frame, gold, cfg, coef_expect, with_c3=True):
    """The drop-in boundary as a reference user sees it: icer_init / icer_init_output_struct / icer_compress_image_uint16 with
    PAGEABLE caller memory (example/src/example_encode.c:36-77, example/src/icer_util.c:186-206), whole call: upload, all
    kernels, stream download and the coefficient planes written back over the caller's image (icer_wavelet.c:871-877).
    `coef_expect`: the coefficient plane of the same frame read from the device-resident encoder (icerx_get_coefficients)."""
    from icer_compression_amd import api, synth
    W, H = cfg["w"], cfg["h"]
    quota = 2 * W * H
    api.icer_init()
    buf = np.zeros(2 * quota + 64, np.uint8)
    od = api.icer_output_data_buf_typedef()

    img = np.empty_like(host_frame)                               # the caller's image buffer, reused call after call

    def call():
        np.copyto(img, host_frame)                                # (the call overwrites its input, as the reference does)
        assert api.icer_init_output_struct(od, buf, buf.size, quota) == 0
        t = time.perf_counter()
        rc = api.icer_compress_image_uint16(img, W, H, cfg["stages"], FILT, cfg["segments"], od)
        return time.perf_counter() - t, rc, img
    call()
    times, ok = [], True
    for _ in range(6):
        dt, rc, img = call()
        times.append(dt)
        stream = buf[quota: quota + od.size_used].tobytes()
        ok = ok and rc == 0 and len(stream) == gold[0] and "%08x" % zlib.crc32(stream) == gold[1]
    ok = ok and bool(np.array_equal(img, coef_expect))
    t = min(times)
    obj = {"workload": "icer_compress_image_uint16 on the timed workload's frame: pageable caller memory -> stream in the caller's buffer, coefficient planes "
                       "written back over the caller's image; whole call", "ms_per_frame": round(t * 1e3, 3), "ms_per_frame_median": round(sorted(times)[len(times) // 2] * 1e3, 3),
           "value": round(W * H / t / 1e6, 3), "unit": "Mpixels/s", "parity": bool(ok),
           "parity_note": "rc, length and CRC-32 equal the reference golden in every call; the image left behind equals the device-resident encoder's coefficient plane",
           "pcie_bytes_per_call": int(2 * W * H * 2 + gold[0])}
    if with_c3:
        try:
            with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as fh:
                g = json.load(fh)["C3_4096_yuv_quota70000"]
            planes = [np.ascontiguousarray(p) for p in synth.color_frame_yuv(g["w"], g["h"], synth.DEFAULT_SEED)]
            cbuf = np.zeros(2 * g["quota"] + 64, np.uint8)
            ts, okc = [], True
            for k in range(4):
                work = [p.copy() for p in planes]
                assert api.icer_init_output_struct(od, cbuf, cbuf.size, g["quota"]) == 0
                t0 = time.perf_counter()
                rc = api.icer_compress_image_yuv_uint16(work[0], work[1], work[2], g["w"], g["h"], g["stages"], FILT, g["segments"], od)
                if k:
                    ts.append(time.perf_counter() - t0)
                st = cbuf[g["quota"]: g["quota"] + od.size_used].tobytes()
                okc = okc and rc == g["rc"] and len(st) == g["size"] and "%08x" % zlib.crc32(st) == g["crc32"]
            obj["C3"] = {"workload": "icer_compress_image_yuv_uint16, BASELINE configs[2] (4096x4096 YUV, quota 70 000), pageable planes, three coefficient planes written back",
                         "ms_per_frame": round(min(ts) * 1e3, 3), "value": round(g["w"] * g["h"] / min(ts) / 1e6, 3), "unit": "Mpixels/s", "parity": bool(okc)}
        except Exception as exc:                                   # noqa: BLE001 -- secondary figure
            obj["C3"] = {"error": repr(exc)}
    return obj


def batch_decode_object(bw):
    """SURVEY 8f next-1 'batched decode': the streams a device-resident batch workload has just produced (this rank's share of C4
    or C5, still in HBM) back through libicer_hip_dec.so in ONE icerx_decode_device call; every decoded frame against the input"""
    import torch
    from icer_compression_amd import decoder
    c, W, H, B = bw.cfg, bw.w, bw.h, bw.B
    sizes = [int(x) for x in bw.sizes.cpu().numpy()]
    offs = [k * bw.out.stride(0) for k in range(B)]
    d_out = torch.zeros((B, H * W), dtype=torch.int16, device=bw.out.device)
    dec = decoder.Decoder(1, c["stages"], FILT, c["segments"])
    times = []
    for _ in range(2):                                            # (first call: allocations)
        torch.cuda.synchronize()
        t = time.perf_counter()
        rc, rcs, _, _ = dec.decode_device(B, bw.out.data_ptr(), offs, sizes, d_out.data_ptr(), W * H)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t)
    ok = rc == 0 and all(r == 0 for r in rcs) and all(bool(torch.equal(d_out[k].view(H, W), bw.frames[k])) for k in range(B))
    dec.close()
    del d_out
    return {"value": round(B * W * H / times[-1] / 1e6, 2), "unit": "Mpixels/s", "ms_per_call": round(times[-1] * 1e3, 2), "streams_per_call": B,
            "parity": bool(ok), "parity_note": "every decoded frame equals the encoder's input (lossless streams)"}


def c2_per_rank_object(rank, dev, local_rank, barrier, all_ranks_ok, world, red_dev, steps=10):
    """secondary object of an N > 1 line: the lone C2 frame (BASELINE configs[1], the N = 1 headline) on every rank at once"""
    from icer_compression_amd import shard
    w2 = Workload("C2", rank, dev, local_rank)
    w2.step()
    bad, _ = w2.verify()
    t_el, _, st_ms, calls = run_timed(w2, steps, 2, barrier, dev, red_dev)
    bad2, _ = w2.verify()
    ok = all_ranks_ok(not bad and not bad2)
    w2.close()
    return {"workload": "BASELINE configs[1]: one 4096x4096 frame per rank per step (weak: every rank the same frame), device-resident",
            "value": round(world * w2.w * w2.h * steps / t_el / 1e6, 3), "unit": "Mpixels/s", "ms_per_step": round(t_el / steps * 1e3, 4), "steps": steps,
            "n_gpus": world, "parity": ok, "code_units_ms": round(st_ms.get("code_units", 0.0) / max(calls, 1), 4)}


def one_process_object(name, dev, local_rank, calls=2):
    """SURVEY 8(b) "our additions": ONE process, every visible device -- icerx_compress_batch_uint16(..., n_gpus = 0) on the WHOLE
    batch of configuration `name` in page-locked host memory (a worker thread and a pool of sub-batch encoders per device inside
    the library, no torch.distributed); every frame against its reference golden"""
    import torch
    hw = HostWorkload(name, 0, dev, local_rank, first=0, count=CONFIGS[name]["total"], devices=None)
    try:
        hw.step()
        bad, nbytes = hw.verify()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            hw.step()
        t_el = time.perf_counter() - t0
        bad2, _ = hw.verify()
        c = hw.cfg
        return {"workload": f"{c['what']}: all {c['total']} x {c['w']}x{c['h']} frames in page-locked host memory -> streams in page-locked host memory, "
                            "icerx_compress_batch_uint16(n_gpus = 0) from one process over every visible device",
                "value": round(c["total"] * c["w"] * c["h"] * calls / t_el / 1e6, 3), "unit": "Mpixels/s", "ms_per_call": round(t_el / calls * 1e3, 3), "calls": calls,
                "devices": torch.cuda.device_count(), "logical_devices": int(os.environ.get("ICER_HIP_VIRTUAL_DEVICES", "0")) or torch.cuda.device_count(),
                "frames_checked": c["total"], "frames_not_bit_exact": bad + [b for b in bad2 if b not in bad], "parity": not bad and not bad2,
                "pinned": hw.pinned, "bytes_out": nbytes}
    finally:
        hw.close()


def crowded_host_object(name, rank, dev, local_rank, device_value):
    """The host-fed batch in a process that is NOT the library's alone: two torch streams that have done work and stay alive,
    and a live decoder with its side streams (the runtime shares its hardware queues out over every live stream of the process,
    INTEGRATION.md 4).  Same call, same frames, every frame checked; the rate beside the quiet-process one."""
    import torch
    from icer_compression_amd import api, decoder
    others = [torch.cuda.Stream(device=dev) for _ in range(2)]
    junk = []
    for st in others:
        with torch.cuda.stream(st):
            junk.append(torch.ones(1 << 20, device=dev) * 2)
    torch.cuda.synchronize(dev)
    c = CONFIGS[name]
    dec = decoder.Decoder(1, c["stages"], FILT, c["segments"])
    hw = HostWorkload(name, rank, dev, local_rank)
    try:
        hw.step()
        bad, _ = hw.verify()
        n = 6 if name == "C4" else 3
        t0 = time.perf_counter()
        for _ in range(n):
            hw.step()
        t_el = time.perf_counter() - t0
        bad2, _ = hw.verify()
        val = n * c["per_gpu"] * c["w"] * c["h"] / t_el / 1e6
        st = api.process_stats()
        return {"value": round(val, 3), "unit": "Mpixels/s", "ms_per_call": round(t_el / n * 1e3, 3), "calls": n, "parity": not bad and not bad2,
                "frames_checked": c["per_gpu"], "live_streams_besides_the_library": "2 torch streams + a decoder's", "unit_timeouts": st["unit_timeouts"],
                "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "vs_device_resident": round(val / device_value, 3) if device_value else None}
    finally:
        hw.close()
        dec.close()
        del junk, others


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher around it: re-run this very command line as N ranks through
    torch.distributed.run on 127.0.0.1 (a free port), one rank per GPU; with fewer visible GPUs than ranks the ranks share
    devices (the dry-run path of main(): logical devices + gloo).  Returns the launcher's exit code."""
    import socket
    rc = 1
    for attempt in range(3):                  # (a port found free by bind-then-close can be taken before the ranks bind it: try another one)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        p = subprocess.run(cmd, env=env, stderr=subprocess.PIPE, text=True)
        sys.stderr.write(p.stderr)
        rc = p.returncode
        if rc == 0 or not any(m in p.stderr for m in ("EADDRINUSE", "Address already in use", "address already in use")):
            break
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=list(CONFIGS), default=None, help="timed workload (default: C2 = BASELINE configs[1] on one GPU; "
                    "C4, the 256-frame batch split over the ranks, with --gpus N > 1)")
    ap.add_argument("--sweep", action="store_true", help="with --config C4|C5 on one GPU: run the share of every rank of an 8-GPU job in turn "
                    "(all 256 / 64 frames), every frame against its reference golden")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None, help="strong (--config C4|C5): the WHOLE batch split over the ranks, "
                    "N = 1 codes all of it; weak: every rank its 1/8 share (default: weak on one GPU and for C2, strong for a batch with --gpus N > 1)")
    ap.add_argument("--source", choices=["device", "host"], default="device", help="host: frames and streams in page-locked host memory, "
                    "icerx_compress_batch_uint16_devices (PCIe overlapped with the kernels)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch-configs", action="store_true", help="skip the secondary C4 / C5 figures")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 counter passes (roofline.traffic = null)")
    ap.add_argument("--no-batch-traffic", action="store_true", help="skip the rocprofv3 counter passes of the C4 / C5 batch configurations (batch_configs.*.roofline)")
    ap.add_argument("--no-extras", action="store_true", help="skip C3, decode, batch_host and host_buffers (the child runs under rocprofv3)")
    ap.add_argument("--in-flight", type=int, choices=[1, 2], default=None, help="launches in flight for a batch workload whose step is ONE launch (--config C4|C5): "
                    "2 = step k is submitted before step k - 1 is waited for (default: 2 with --scaling strong, else 1)")
    ap.add_argument("--launch-probe", action="store_true", help="launcher check only (runs without a GPU): the ranks rendezvous over gloo, reduce their "
                    "shard sizes of the timed configuration and rank 0 prints one JSON line; no encode")
    ap.add_argument("--all-extras", action="store_true", help="with --gpus N > 1: also the host-fed, decode and in-flight legs (default: N = 1 only)")
    ap.add_argument("--one-process-child", action="store_true", help=argparse.SUPPRESS)     # (internal: the one_process object in a process of its own)
    ap.add_argument("--no-one-process", action="store_true", help="skip the one_process object (icerx_compress_batch_uint16 over every visible device)")
    ap.add_argument("--batched-probe", type=int, default=8,
                    help="also report C2 throughput with this many frames per launch (secondary figure, 0 = skip)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    # N > 1: the whole C4 batch (256 frames, BASELINE configs[3]: what north_star says scales) split over the ranks, every frame
    # checked; the lone C2 frame per rank is a secondary object of that line.  N = 1: C2, the configuration the metric is quoted on.
    if args.config is None:
        args.config = "C2" if args.gpus == 1 else "C4"
    if args.scaling is None:
        args.scaling = "strong" if (args.gpus > 1 and args.config != "C2") else "weak"

    if args.one_process_child:
        # the one_process object, in a process of its own (started by rank 0 with the launcher's variables removed): a hang or a
        # crash of the multi-device host batch -- a path no box with more than one GPU has run yet -- costs the line this object only
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("no HIP device")
        torch.zeros(1, device="cuda")
        print(json.dumps({"one_process": one_process_object(args.config or "C4", torch.device("cuda", 0), 0)}), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started plainly (`python bench.py --gpus N`): this process becomes the launcher of one rank per GPU and
        # passes the ranks' output through; under torch.distributed.run (the driver's form) WORLD_SIZE is set and we are a rank.
        raise SystemExit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    from icer_compression_amd import api, shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start `python bench.py --gpus N` plainly (it launches its own ranks) "
                         f"or under torch.distributed.run --nproc-per-node N")
    if args.launch_probe:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo")
        lo, hi = shard.shard_range(CONFIGS[args.config]["total"], rank, world)
        t = torch.tensor([hi - lo, rank], dtype=torch.int64)
        if world > 1:
            dist.all_reduce(t)
            dist.barrier()
        if rank == 0:
            print(json.dumps({"launch_probe": True, "n_gpus": world, "config": args.config, "scaling": args.scaling,
                              "frames_over_ranks": int(t[0]), "rank_sum": int(t[1])}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if world > 1 and not args.all_extras:
        args.no_extras = True                                     # an N > 1 line: the headline, batch_configs, c2_per_rank, one_process
    # One process per GPU.  A dry run of the multi-rank path on a box with FEWER GPUs than ranks (gpurun exposes one) maps
    # rank r onto physical device r % devices: the library's logical devices (ICER_HIP_VIRTUAL_DEVICES) and the gloo backend
    # for the three collectives of this file (RCCL refuses two ranks on one device).  Nothing else changes.
    ndev = torch.cuda.device_count()
    oversubscribed = world > ndev
    if oversubscribed:
        os.environ["ICER_HIP_VIRTUAL_DEVICES"] = str(world)
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    red_dev = dev                                                 # where the reduction tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if oversubscribed:
            dist.init_process_group("gloo")
            red_dev = torch.device("cpu")
        else:
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
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    cfg = CONFIGS[args.config]
    W, H = cfg["w"], cfg["h"]
    if args.scaling == "strong" and args.config == "C2":
        raise SystemExit("--scaling strong needs a batch configuration (--config C4|C5): C2 is one frame")
    if args.sweep and (args.config == "C2" or world != 1):
        raise SystemExit("--sweep needs --config C4|C5 on one GPU")
    WL = HostWorkload if args.source == "host" else Workload
    device_mode = args.source != "host"

    # ---- sweep: this GPU plays every rank of the 8-GPU job in turn ----------------------------------------------------
    if args.sweep:
        shares = cfg["total"] // cfg["per_gpu"]
        bad_all, ms_all, checked, bytes_all = [], [], 0, 0
        for r in range(shares):
            wl = WL(args.config, r, dev, local_rank)
            wl.step()
            bad, nbytes = wl.verify()
            t_el, _, _, _ = run_timed(wl, 2, 0, barrier, dev, red_dev)
            bad2, _ = wl.verify()
            bad_all += bad + [b for b in bad2 if b not in bad]
            ms_all.append(round(t_el / 2 * 1e3, 3))
            checked += wl.B
            bytes_all += nbytes
            wl.close()
            del wl
            torch.cuda.empty_cache()
        tot_ms = sum(ms_all)
        print(json.dumps({"sweep": args.config, "source": args.source, "workload": f"{cfg['what']}: all {cfg['total']} x {W}x{H} frames as {shares} launches of {cfg['per_gpu']} on ONE GPU",
                          "frames_checked": checked, "frames_not_bit_exact": bad_all, "parity": not bad_all, "bytes_out": bytes_all,
                          "parity_note": "every frame: rc, length and CRC-32 equal the reference CPU encoder's (tests/golden/batch_golden.json), after the first "
                                         "encode and again after two timed ones", "ms_per_launch": ms_all,
                          "value": round(cfg["total"] * W * H / (tot_ms * 1e-3) / 1e6, 3), "unit": "Mpixels/s", "n_gpus": 1}), flush=True)
        raise SystemExit(0 if not bad_all else 1)

    # ---- the timed workload ----------------------------------------------------------------------------------------------
    if args.scaling == "strong":
        lo, hi = shard.shard_range(cfg["total"], rank, world)
        wl = WL(args.config, rank, dev, local_rank, first=lo, count=hi - lo, **({"in_flight": args.in_flight or 2} if device_mode else {}))
    else:
        wl = WL(args.config, rank, dev, local_rank, **({"in_flight": args.in_flight or 1} if device_mode else {}))
    B = wl.B

    # parity gate first (one extra untimed encode): EVERY rank checks EVERY one of its frames against the reference
    wl.step()
    bad, bytes_out = wl.verify()
    if not all_ranks_ok(not bad):
        raise SystemExit(f"rank {rank}: frames {bad} are not bit-exact with the reference golden; no number reported")

    elapsed_max, step_ms, stage_ms, calls = run_timed(wl, args.steps, args.warmup, barrier, dev, red_dev)
    # ... and again on what the LAST timed step left in the output buffers (the timed calls run with event timing on)
    bad_after, _ = wl.verify()
    parity_after = all_ranks_ok(not bad_after)
    if not parity_after:
        raise SystemExit(f"rank {rank}: frames {bad_after} differ from the reference golden after the timed loop; no number reported")
    h_sizes_sum = bytes_out
    device_wl = args.source == "device"
    units_per_frame = wl.enc.info()["units_per_frame"] if device_wl else None
    stats = wl.enc.stats() if device_wl else api.process_stats()
    launches_per_step = (B + wl.launch - 1) // wl.launch if device_wl else 1

    # the batch configurations BASELINE.json names for 8 GPUs, as this GPU's share of them: device-resident, and fed from
    # page-locked host memory through the overlapped batch call
    batch_cfgs, batch_host, kept, batch_roof = {}, {}, {}, {}
    have_bg = os.path.exists(os.path.join(ROOT, "tests", "golden", "batch_golden.json"))
    if not args.no_batch_configs and have_bg:
        for name in ("C4", "C5"):
            if name == args.config:
                continue
            try:
                bw = Workload(name, rank, dev, local_rank)
                bw.step()
                badf, out_bytes = bw.verify()
                nst = 3
                t_el, _, st_ms, cl = run_timed(bw, nst, 1, barrier, dev, red_dev)
                badf2, _ = bw.verify()
                ok = all_ranks_ok(not badf and not badf2)
                c = bw.cfg
                pix = world * c["per_gpu"] * c["w"] * c["h"] * nst
                batch_cfgs[name] = {
                    "workload": f"{c['what']}: {c['total']} x {c['w']}x{c['h']}, {c['stages']} stages, {c['segments']} segments, lossless; "
                                f"{c['per_gpu']} frames per GPU per launch (frames {c['per_gpu']}*rank ..)",
                    "value": round(pix / t_el / 1e6, 3), "unit": "Mpixels/s", "ms_per_launch": round(t_el / nst * 1e3, 3), "launches": nst,
                    "n_gpus": world, "parity": ok, "frames_checked_per_rank": c["per_gpu"],
                    "parity_note": "every frame of every rank: rc, length and CRC-32 equal the reference CPU encoder's (tests/golden/batch_golden.json), "
                                   "before and after the timed launches",
                    "bytes_out_rank0": out_bytes, "code_units_ms": round(st_ms["code_units"] / max(cl, 1), 3), "dwt_ms": round(st_ms["dwt"] / max(cl, 1), 3),
                    "coder": "code_units_kernel<8> for the dense coding units; the all-but-blank ones (>= 95 % blank chunks, listed on the "
                             "device by route_units_kernel) by code_units_list_kernel<WgOne> (one wavefront per staying workgroup) on a second stream beside it"}
                if rank == 0 and world == 1:
                    ab = float(c["per_gpu"] * c["w"] * c["h"] * 2 + out_bytes)
                    kms = st_ms["code_units"] / max(cl, 1)
                    parts = bw.enc.parts()
                    batch_cfgs[name]["roofline_frac"] = round(ab / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6)
                    batch_cfgs[name]["parts_per_call"] = parts
                    if parts > 1:
                        batch_cfgs[name]["stage_note"] = (f"the call enqueues its frames in {parts} parts on two streams (their stages overlap): code_units_ms is the call's whole "
                                                          "span, dwt_ms is booked inside it")
                    batch_roof[name] = (ab, kms, parts, c["per_gpu"] * c["w"] * c["h"])
                bw.close()
                del bw
                torch.cuda.empty_cache()
                if not args.no_extras:
                    kept[name] = True                                  # (its decode and two-in-flight legs come after the host-fed legs, below)
            except Exception as exc:                                   # noqa: BLE001 -- secondary figure
                batch_cfgs[name] = {"error": repr(exc)}
            if args.no_extras:
                continue
            try:
                hw = HostWorkload(name, rank, dev, local_rank)
                hw.step()
                badh, _ = hw.verify()
                nst = 8 if name == "C4" else 4
                t_el, call_ms, _, _ = run_timed(hw, nst, 1, barrier, dev, red_dev)
                badh2, _ = hw.verify()
                c = hw.cfg
                pix = world * c["per_gpu"] * c["w"] * c["h"] * nst
                val = pix / t_el / 1e6
                batch_host[name] = {
                    "workload": f"{c['what']}: this rank's {c['per_gpu']} frames in page-locked host memory -> streams in page-locked host memory; "
                                "icerx_compress_batch_uint16_devices on this rank's GPU (upload, kernels, download of sub-batches on three streams)",
                    "value": round(val, 3), "unit": "Mpixels/s", "ms_per_call": round(t_el / nst * 1e3, 3), "calls": nst, "call_ms": call_ms, "n_gpus": world,
                    "parity": all_ranks_ok(not badh and not badh2), "frames_checked_per_rank": c["per_gpu"], "pinned": hw.pinned,
                    "pcie_bytes_per_call": int(c["per_gpu"] * c["w"] * c["h"] * 2 + hw.sizes.sum()),
                    "vs_device_resident": round(val / batch_cfgs[name]["value"], 3) if "value" in batch_cfgs.get(name, {}) else None}
                hw.close()
                del hw
            except Exception as exc:                                   # noqa: BLE001 -- secondary figure
                batch_host[name] = {"error": repr(exc)}
        # The legs that create streams of their own (the decoder's side streams, the two encoders in turns) come after the
        # host-fed legs, and nothing but the timed workload's encoder is alive while those run: the host-fed pipeline overlaps
        # five streams per device on the runtime's eight hardware queues, and every other live stream of the process -- idle or
        # not -- takes part in how the runtime shares the queues out.  Alone in a process the host-fed legs reach 0.94 x of the
        # device-resident rate; after other legs of this script they measured 0.66-0.90 x (profiles/r04_logs/r04_z*_bench.json).
        for name in list(kept):
            try:
                bw = Workload(name, rank, dev, local_rank)
                bw.step()
            except Exception as exc:                                   # noqa: BLE001 -- secondary figure
                batch_cfgs[name]["two_launches_in_flight"] = {"error": repr(exc)}
                continue
            if rank == 0:
                try:
                    batch_cfgs[name]["decode"] = batch_decode_object(bw)
                except Exception as exc:                               # noqa: BLE001 -- secondary figure
                    batch_cfgs[name]["decode"] = {"error": repr(exc)}
            try:
                batch_cfgs[name]["two_launches_in_flight"] = pingpong_object(bw, name, rank, dev, local_rank, barrier, all_ranks_ok, world, red_dev)
            except Exception as exc:                                   # noqa: BLE001 -- secondary figure
                batch_cfgs[name]["two_launches_in_flight"] = {"error": repr(exc)}
            bw.close()
        kept.clear()
        torch.cuda.empty_cache()
        # ... the host-fed call in a process that has NOT asked for more hardware queues (a child without GPU_MAX_HW_QUEUES: the
        # library then puts its encoder streams on the low priority level's queue pool, INTEGRATION.md "Hardware queues"), quiet and crowded
        if not args.no_extras and world == 1 and "C4" in batch_host and "value" in batch_host["C4"]:
            try:
                env = dict(os.environ, GPU_MAX_HW_QUEUES="default")
                env.pop("ICER_HIP_STREAM_PRIO", None)
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_batch_probe.py"), "C4"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
                o = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
                dv = batch_cfgs.get("C4", {}).get("value")
                batch_host["C4"]["runtime_default_hw_queues"] = {
                    "value": o["quiet_Mpix_s"], "crowded_process": o["crowded_Mpix_s"], "unit": "Mpixels/s", "parity": o["parity"],
                    "vs_device_resident": round(o["quiet_Mpix_s"] / dv, 3) if dv else None,
                    "note": "a child process without GPU_MAX_HW_QUEUES (4 hardware queues per priority level): the pipeline's encoder streams are on the low priority level's own queue pool"}
            except Exception as exc:                                   # noqa: BLE001 -- secondary figure
                batch_host["C4"]["runtime_default_hw_queues"] = {"error": repr(exc)}

    # secondary figure: a STREAM of single frames (not `value`, which times one frame at a time): two encoders taking the frames in
    # turns, two launches in flight.  (First among the secondary legs: the two launches overlap only while their streams sit on
    # hardware queues of their own, and which queue a stream gets goes by how many the process has created before -- measured
    # with tools/batch_pingpong_probe.py: 4.15 ms per frame with GPU_MAX_HW_QUEUES = 8, 6.46 with 4, 8.7 with 2.)
    streaming = None
    if not args.no_extras and args.config == "C2" and device_wl:
        try:
            streaming = pingpong_object(wl, "C2", rank, dev, local_rank, barrier, all_ranks_ok, world, red_dev, n=32)
            streaming["workload"] = "a stream of 4096x4096 frames (the timed workload's frame), two single-frame launches in flight: frames per second, not the latency of one"
            streaming["ms_per_frame"] = streaming.pop("ms_per_launch")
            streaming["frames_in_flight"] = 2
        except Exception as exc:                                       # noqa: BLE001 -- secondary figure
            streaming = {"error": repr(exc)}

    # secondary figure: the C2 geometry with several frames per launch; one frame alone cannot fill 256 CUs because its
    # largest coding units form a serial chain
    batched = None
    if args.batched_probe > 1 and args.config == "C2" and device_wl:
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
        tb = shard.max_over_ranks(time.perf_counter() - tb, red_dev)
        g = wl.gold[0]
        ok = bool((brcs.cpu().numpy() == 0).all()) and all(
            int(bsizes[k]) == g[0] and ("%08x" % zlib.crc32(bout[k, : g[0]].cpu().numpy().tobytes())) == g[1] for k in range(PB))
        batched = {"frames_per_gpu_per_launch": PB, "value": round(world * PB * W * H * nb / tb / 1e6, 3), "unit": "Mpixels/s",
                   "ms_per_launch": round(tb / nb * 1e3, 3), "parity": all_ranks_ok(ok), "frames_checked_per_rank": PB}
        benc.close()
        del bf, bout

    extras = {}
    if not args.no_extras and args.config == "C2" and device_wl:
        try:
            extras["C3"] = c3_object(dev, local_rank, barrier, all_ranks_ok, world, red_dev)
        except Exception as exc:                                       # noqa: BLE001 -- secondary figure
            extras["C3"] = {"error": repr(exc)}
        if rank == 0:
            try:
                extras["decode"] = decode_object(wl.out[0], wl.gold[0][0], wl.frames[0], cfg, with_cpu=not args.no_cpu_baseline)
            except Exception as exc:                                   # noqa: BLE001 -- secondary figure
                extras["decode"] = {"error": repr(exc)}

    # secondary objects of an N > 1 line whose headline is a batch: the lone C2 frame on every rank; at N = 1 (headline C2): the
    # WHOLE C4 batch on this one GPU -- the N = 1 point of the strong-scaling curve the N > 1 lines continue
    c2_rank = scaling_ref = None
    if device_wl and have_bg:
        if world > 1 and args.config != "C2":
            try:
                c2_rank = c2_per_rank_object(rank, dev, local_rank, barrier, all_ranks_ok, world, red_dev)
            except Exception as exc:                                   # noqa: BLE001 -- secondary figure
                c2_rank = {"error": repr(exc)}
        elif world == 1 and args.config == "C2" and not args.no_batch_configs:
            try:
                sw = Workload("C4", 0, dev, local_rank, first=0, count=CONFIGS["C4"]["total"])
                sw.step()
                bads, _ = sw.verify()
                t_el, _, _, _ = run_timed(sw, 2, 0, barrier, dev, red_dev)
                bads2, _ = sw.verify()
                scaling_ref = {"workload": "BASELINE configs[3], the headline of the N > 1 lines (--gpus N: the 256 frames split over N ranks), at N = 1: "
                                           "all 256 x 2048x2048 frames on this GPU per step, 8 launches of 32, two in flight",
                               "value": round(CONFIGS["C4"]["total"] * sw.w * sw.h * 2 / t_el / 1e6, 3), "unit": "Mpixels/s", "ms_per_step": round(t_el / 2 * 1e3, 3),
                               "steps": 2, "n_gpus": 1, "scaling": "strong", "frames_checked": sw.B, "parity": not bads and not bads2}
                sw.close()
                del sw
                torch.cuda.empty_cache()
            except Exception as exc:                                   # noqa: BLE001 -- secondary figure
                scaling_ref = {"error": repr(exc)}
    # N > 1: the same strong-scaling split FED FROM THE HOST (icerx_compress_batch_uint16_devices on every rank's own device: frames and
    # streams in page-locked host memory) beside the device-resident headline -- at 8 GPUs the ranks share the host's DRAM and PCIe root
    # (about 270 MB in and 80 MB out per rank and step of 32 frames), so this is the curve that can bend first
    host_fed = None
    if world > 1 and args.config != "C2" and args.scaling == "strong" and device_wl and have_bg:
        try:
            lo, hi = shard.shard_range(cfg["total"], rank, world)
            hw = HostWorkload(args.config, rank, dev, local_rank, first=lo, count=hi - lo)
            hw.step()
            badh, _ = hw.verify()
            nst = 3
            t_el, _, _, _ = run_timed(hw, nst, 1, barrier, dev, red_dev)
            badh2, _ = hw.verify()
            host_fed = {"workload": f"{cfg['what']}: the whole batch of {cfg['total']} frames split over {world} ranks, every rank's share in page-locked HOST memory -> "
                                    "streams in page-locked host memory through icerx_compress_batch_uint16_devices on its own device (copies and kernels of sub-batches overlap)",
                        "value": round(cfg["total"] * hw.w * hw.h * nst / t_el / 1e6, 3), "unit": "Mpixels/s", "ms_per_step": round(t_el / nst * 1e3, 3), "steps": nst,
                        "n_gpus": world, "scaling": "strong", "parity": all_ranks_ok(not badh and not badh2), "frames_checked_per_rank": hi - lo,
                        "pinned": hw.pinned}
            hw.close()
            del hw
        except Exception as exc:                                       # noqa: BLE001 -- secondary figure
            host_fed = {"error": repr(exc)}
    # ... and last of the legs of this process: the host-fed call again with other streams alive in the process (torch's stream pool stays
    # alive from here on, which is why nothing that overlaps launches on streams of its own comes after this)
    if not args.no_extras and world == 1 and not args.no_batch_configs and have_bg:
        for name in ("C4", "C5"):
            if name in batch_host and "value" in batch_host[name]:
                try:
                    batch_host[name]["crowded_process"] = crowded_host_object(name, rank, dev, local_rank, batch_cfgs.get(name, {}).get("value"))
                except Exception as exc:                               # noqa: BLE001 -- secondary figure
                    batch_host[name]["crowded_process"] = {"error": repr(exc)}
    one_proc = None
    if device_wl and have_bg and not args.no_one_process and not args.no_batch_configs:
        barrier()                                                  # the other ranks wait here while rank 0's library drives every device
        if rank == 0:
            try:
                env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                                         "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "ICER_HIP_VIRTUAL_DEVICES")}
                if oversubscribed:
                    env["ICER_HIP_VIRTUAL_DEVICES"] = str(world)        # (the dry run: as many logical devices as ranks)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one-process-child", "--config", "C4"], env=env, cwd=ROOT,
                                   capture_output=True, text=True, timeout=420)
                objs = [json.loads(x) for x in r.stdout.splitlines() if x.startswith('{"one_process"')]
                one_proc = objs[-1]["one_process"] if objs else {"error": f"child rc={r.returncode}: {r.stderr[-400:]}"}
            except Exception as exc:                                   # noqa: BLE001 -- secondary figure (a time-out included)
                one_proc = {"error": repr(exc)}
        barrier()

    if rank == 0:
        n_pix = world * B * W * H * args.steps
        if args.scaling == "strong":
            n_pix = cfg["total"] * W * H * args.steps
        value = n_pix / elapsed_max / 1e6
        src = ("input and output stream resident in HBM" if device_wl else
               "frames and streams in page-locked HOST memory (icerx_compress_batch_uint16_devices: PCIe overlapped with the kernels)")
        line = {
            "metric": f"Mpixels/s encode (bit-exact), {W}x{H} gray", "value": round(value, 3), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": f"{cfg['what']}: {B} x {W}x{H} 8-bit gray (uint16 API) on rank 0 per step"
                                   + (f" (the whole batch of {cfg['total']} split over {world} ranks)" if args.scaling == "strong" else " (every rank the same amount)")
                                   + f", {cfg['stages']} DWT stages, filter A, {cfg['segments']} segments, lossless quota 2*W*H; {src}",
                       "frames_per_gpu_per_step": B, "launches_per_step": launches_per_step, "units_per_frame": units_per_frame,
                       "launches_in_flight": 2 if (getattr(wl, "pipelined", False) or getattr(wl, "cross_step", False)) else 1,
                       "parity": "every frame of every rank: rc, stream length and CRC-32 equal the reference golden (checked before timing and again "
                                 "on the output of the last timed step)"},
            "parity_after_timing": parity_after,
            "physical_gpus": min(world, ndev),
            "step_ms": step_ms,
            "coder_events": {k: stats[k] for k in ("unit_timeouts", "fallback_batches", "slot_retries")},
        }
        k_ms = 0.0
        if device_wl:
            # roofline of the dominant kernel (the coding-unit kernel): algorithmic bytes per launch =
            # SURVEY 8(d) per-frame figure (input planes read once + final stream written once) x frames per launch
            k_ms = stage_ms["code_units"] / max(calls, 1)
            if getattr(wl, "pipelined", False) or getattr(wl, "cross_step", False):
                # two launches in flight: a launch's own events span the time it shared the chip with its neighbour; what a launch
                # costs is the step's time over its launches
                k_ms = elapsed_max / args.steps / launches_per_step * 1e3
            alg_bytes = float(B * W * H * 2 + h_sizes_sum) / launches_per_step
            achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
            li = wl.enc.launch_info()
            pipe = f"code_units_kernel<{li['pipeline_waves']}> (wave pipeline, {li['pipeline_waves']}-wave workgroups)"
            if li["split"]:
                pipe += (f"; the dense coding units cut into sub-ranges, {li['sub_range_workgroups']} extra workgroups in the same launch, spliced by "
                         "splice_units_kernel (inside the timed stage)")
            if li["window_coder_beside"]:
                pipe += "; the all-but-blank units by code_units_list_kernel (the small window coder: four-wave workgroups for a lone frame, one-wave ones in a batch) on a second stream beside it (inside the timed stage)"
            kernel = {0: pipe + "; code_units_wg_kernel in progressive mode", 1: pipe, 2: "code_units_wg_kernel"}[stats["coder_mode"]]
            line["roofline"] = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS,
                                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": None,
                                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(k_ms, 4)}
            line["stage_ms_per_step"] = {k: round(v / max(calls, 1) * launches_per_step, 4) for k, v in stage_ms.items()}
            if getattr(wl, "pipelined", False) or getattr(wl, "cross_step", False):
                line["stage_ms_note"] = ("two launches in flight: the stage times are each launch's own event spans (they overlap its neighbour's and add up to "
                                         "more than the step); roofline.avg_launch_ms = step time / launches")
        if batched:
            line["batched"] = batched
        if batch_cfgs:
            line["batch_configs"] = batch_cfgs
        if batch_host:
            line["batch_host"] = batch_host
        if c2_rank:
            line["c2_per_rank"] = c2_rank
        if scaling_ref:
            line["scaling_reference"] = scaling_ref
        if host_fed:
            line["host_fed"] = host_fed
        if one_proc:
            line["one_process"] = one_proc
        if world > 1 and args.config != "C2":
            line["scaling_note"] = ("the N > 1 headline is the C4 batch split over the ranks (strong scaling); its N = 1 point is "
                                    "`scaling_reference` of the --gpus 1 line (same 256 frames on one GPU), not that line's C2 `value`")
        if streaming:
            line["streaming"] = streaming
        line.update(extras)
        if world == 1 and not args.no_traffic and device_wl:
            ctr, src_note = measure_traffic(args)
            line["roofline"]["traffic_source"] = src_note + f"; sources {source_digest()}"
            if ctr and "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
                line["roofline"]["traffic"] = int(2 * ctr["FETCH_SIZE"] * 1024 + ctr["WRITE_SIZE"] * 1024)
                line["roofline"]["traffic_counters_KiB"] = {"FETCH_SIZE": round(ctr["FETCH_SIZE"], 1), "WRITE_SIZE": round(ctr["WRITE_SIZE"], 1)}
                line["roofline"]["traffic_note"] = "the dominant kernel alone, per dispatch (2 x FETCH_SIZE + WRITE_SIZE); the whole call: traffic_launch"
            if ctr and ctr.get("per_kernel"):
                tl, table = launch_traffic(ctr["per_kernel"], wl.enc.parts())
                if tl:
                    line["roofline"]["traffic_launch"] = tl
                    line["roofline"]["traffic_launch_over_algorithmic"] = round(tl / (alg_bytes * launches_per_step), 3)
                    line["roofline"]["traffic_launch_kernels"] = table
                    line["roofline"]["traffic_launch_note"] = ("HBM bytes of one call summed over every kernel it launches; per kernel fetch_factor x FETCH_SIZE + WRITE_SIZE, the "
                                                               "factor calibrated per access pattern (bench.py FETCH_FACTOR); avg_us_under_rocprof = the kernel's duration in that pass")
                    dom = max((k for k in table if k.startswith("code_units_kernel")), key=lambda k: table[k]["bytes_per_call"], default=None)
                    if dom and "avg_us_under_rocprof" in table[dom]:
                        line["roofline"]["kernel_avg_ms_rocprof"] = round(table[dom]["avg_us_under_rocprof"] / 1e3, 4)
                        line["roofline"]["kernel_name_rocprof"] = dom
            if ctr and "SQ_INSTS_VALU" in ctr:
                cyc = k_ms * 1e-3 * CLOCK_HZ
                issue = {"valu_wave_insts_per_launch": int(ctr["SQ_INSTS_VALU"]), "salu_insts_per_launch": int(ctr.get("SQ_INSTS_SALU", 0)),
                         "lds_insts_per_launch": int(ctr.get("SQ_INSTS_LDS", 0))}
                cpi = ISSUE_CYCLES.get("valu_int32")
                if cpi:
                    issue["valu_cycles_per_wave_inst"] = cpi
                    issue["valu_cycles_source"] = ISSUE_CYCLES.get("source")
                    issue["valu_busy_frac"] = round(ctr["SQ_INSTS_VALU"] * cpi / (CUS * SIMDS_PER_CU * cyc), 4)
                    issue["note"] = ("valu_busy_frac = VALU wave-instructions x measured SIMD cycles per wave-instruction of the coder's class (32-bit integer) "
                                     "/ (1024 SIMDs x kernel cycles at 2.4 GHz)")
                if "SQ_WAVE_CYCLES" in ctr and ctr["SQ_WAVE_CYCLES"]:
                    wc = ctr["SQ_WAVE_CYCLES"]
                    issue["hw_wave_time_shares"] = {k: round(ctr[n] / wc, 4) for k, n in (("issuing_any", "SQ_ACTIVE_INST_ANY"), ("issuing_valu", "SQ_ACTIVE_INST_VALU"),
                                                    ("issuing_scalar", "SQ_ACTIVE_INST_SCA"), ("issuing_lds", "SQ_ACTIVE_INST_LDS"), ("parked_waitcnt_or_sleep", "SQ_WAIT_ANY"),
                                                    ("issue_stalled", "SQ_WAIT_INST_ANY")) if n in ctr}
                    issue["hw_wave_quad_cycles_per_launch"] = int(wc)
                    issue["hw_note"] = "shares of SQ_WAVE_CYCLES (summed over all resident waves; quad-cycle units): what the waves' time went into, no cycles-per-instruction assumption"
                if "SQ_LDS_BANK_CONFLICT" in ctr and ctr.get("SQ_LDS_IDX_ACTIVE"):
                    issue["lds_bank_conflict_share_of_lds_cycles"] = round(ctr["SQ_LDS_BANK_CONFLICT"] / ctr["SQ_LDS_IDX_ACTIVE"], 4)
                line["roofline"]["issue"] = issue
        if world == 1 and not args.no_traffic and not args.no_batch_traffic and device_wl:
            # the same audit for the batch configurations: counters of a child run of that configuration (three passes each)
            for name, (ab, kms, parts, pix) in batch_roof.items():
                try:
                    ctr, src_note = measure_traffic(args, config=name, passes=("stats", "fetch", "write", "sq"))
                    ro = {"bound": "hbm", "achieved": round(ab / (kms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                          "frac": round(ab / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6), "algorithmic_bytes_per_launch": ab, "avg_launch_ms": round(kms, 4),
                          "traffic": None, "traffic_source": src_note + f"; sources {source_digest()}"}
                    if ctr and ctr.get("per_kernel"):
                        tl, table = launch_traffic(ctr["per_kernel"], parts)
                        dom = max((k for k in table if k.startswith("code_units_kernel")), key=lambda k: table[k]["bytes_per_call"], default=None)
                        if dom:
                            ro["kernel"] = dom
                            ro["traffic"] = table[dom]["bytes_per_call"]
                            ro["kernel_dispatches_per_call"] = table[dom]["dispatches_per_call"]
                            if "avg_us_under_rocprof" in table[dom]:
                                ro["kernel_avg_ms_rocprof"] = round(table[dom]["avg_us_under_rocprof"] / 1e3, 4)
                        if tl:
                            ro["traffic_launch"] = tl
                            ro["traffic_launch_over_algorithmic"] = round(tl / ab, 3)
                            ro["traffic_launch_kernels"] = table
                    if ctr and "SQ_INSTS_VALU" in ctr:
                        disp = ro.get("kernel_dispatches_per_call", 1) or 1
                        ro["issue"] = {"valu_wave_insts_per_call": int(ctr["SQ_INSTS_VALU"] * disp), "salu_insts_per_call": int(ctr.get("SQ_INSTS_SALU", 0) * disp),
                                       "lds_insts_per_call": int(ctr.get("SQ_INSTS_LDS", 0) * disp),
                                       "valu_wave_insts_per_pixel": round(ctr["SQ_INSTS_VALU"] * disp / pix, 2),
                                       "note": "code_units_kernel alone (SQ_INSTS_* per dispatch x dispatches per call)"}
                    batch_cfgs[name]["roofline"] = ro
                except Exception as exc:                               # noqa: BLE001 -- secondary figure
                    batch_cfgs[name]["roofline"] = {"error": repr(exc)}
        if world == 1 and args.config == "C2" and device_wl and not args.no_extras:
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
                                    "note": "icerx_encode_host: H2D of the frame, all kernels, D2H of size/rc/stream, nothing overlapped (one frame); caller buffers "
                                            "page-locked with icerx_pin_host when available", "by_caller_memory": hb}
        if oversubscribed:
            line["note"] = (f"DRY RUN of the {world}-rank path on {ndev} physical GPU(s): ranks share devices (ICER_HIP_VIRTUAL_DEVICES={world}, gloo for the "
                            "barrier / MAX / MIN reductions); the value is not a scaling point")
        if world == 1 and args.config == "C2" and device_wl and not args.no_extras:
            try:
                wl.step()
                torch.cuda.synchronize()
                line["dropin"] = dropin_object(np.ascontiguousarray(wl.frames[0].cpu().numpy().view(np.uint16)), wl.gold[0], cfg, wl.enc.coefficients(0, 0))
            except Exception as exc:                                   # noqa: BLE001 -- secondary figure
                line["dropin"] = {"error": repr(exc)}
        if world == 1 and args.config == "C2" and device_wl and not args.no_cpu_baseline:
            host_frame = np.ascontiguousarray(wl.frames[:1].cpu().numpy().view(np.uint16))
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
