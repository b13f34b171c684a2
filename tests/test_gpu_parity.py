"""Parity tests proper: the HIP path, called through the C ABI of libicer_hip.so, against the oracle
(oracle/icer_oracle.c), the reference build where present (oracle/_ref) and the golden digests.
Bar: bit-exact streams, identical return codes, identical in-place side effect on the image planes."""
import hashlib
import zlib

import numpy as np
import pytest

from icer_compression_amd import api, synth

pytestmark = pytest.mark.gpu


def _same(a, b):
    return a[0] == b[0] and a[1] == b[1] and all(np.array_equal(p, q) for p, q in zip(a[2], b[2]))


GRAY_CASES = [
    # w, h, stages, filt, segments, quota
    (64, 64, 2, 0, 4, 1 << 20), (200, 160, 3, 1, 1, 1 << 20), (100, 75, 3, 2, 7, 1 << 20), (128, 128, 4, 3, 16, 5000),
    (96, 96, 2, 4, 3, 3000), (257, 131, 3, 5, 32, 1 << 20), (80, 80, 2, 6, 5, 900), (512, 512, 3, 0, 10, 1 << 20),
    (64, 64, 2, 0, 4, 27), (64, 64, 2, 0, 4, 28), (64, 64, 2, 0, 4, 60), (5, 5, 1, 0, 1, 4096), (24, 24, 3, 0, 9, 1 << 16),
    (1000, 37, 3, 0, 10, 1 << 20), (37, 1000, 3, 2, 10, 1 << 20),
]


@pytest.mark.parametrize("case", GRAY_CASES)
@pytest.mark.parametrize("mode", [0, 1])
def test_dropin_gray_matches_oracle(oracle, case, mode):
    w, h, st, f, sg, q = case
    img = synth.gray_frame(w, h, 7, mode)
    assert _same(api.compress([img], st, f, sg, q), oracle.compress([img], st, f, sg, q))


@pytest.mark.parametrize("case", [(64, 64, 2, 0, 4, 1 << 20), (128, 96, 3, 0, 10, 6000), (100, 100, 3, 2, 6, 20000),
                                  (256, 256, 4, 1, 10, 1 << 22)])
def test_dropin_yuv_matches_oracle(oracle, case):
    w, h, st, f, sg, q = case
    planes = synth.color_frame_yuv(w, h, 3)
    assert _same(api.compress(planes, st, f, sg, q), oracle.compress(planes, st, f, sg, q))


def test_dropin_matches_reference_build(reference):
    for (w, h, st, f, sg, q) in [(300, 200, 3, 0, 1, 1 << 20), (128, 128, 4, 2, 16, 7000)]:
        img = synth.gray_frame(w, h, 11, 0)
        assert _same(api.compress([img], st, f, sg, q), reference.compress([img], st, f, sg, q))


def test_true_16bit_data_and_overflow_paths(oracle):
    rng = np.random.default_rng(3)
    # 12-bit data: more than 9 planes of content, exercises category 3 / large magnitudes
    img = rng.integers(0, 4096, (96, 128)).astype(np.uint16)
    assert _same(api.compress([img], 3, 0, 4, 1 << 20), oracle.compress([img], 3, 0, 4, 1 << 20))
    # full-range data overflows int16 in the transform: rc -1, nothing emitted, image left transformed
    img = rng.integers(0, 65536, (64, 64)).astype(np.uint16)
    a, b = api.compress([img], 2, 0, 4, 1 << 20), oracle.compress([img], 2, 0, 4, 1 << 20)
    assert a[0] == b[0] == api.ICER_INTEGER_OVERFLOW and a[1] == b""
    assert np.array_equal(a[2][0], b[2][0])
    # large unsigned values make the LL mean exceed INT16_MAX without transform overflow
    img = np.full((64, 64), 40000, np.uint16)
    a, b = api.compress([img], 2, 0, 4, 1 << 20), oracle.compress([img], 2, 0, 4, 1 << 20)
    assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2][0], b[2][0])


def test_geometry_errors(oracle):
    img = synth.gray_frame(32, 32)
    assert api.compress([img], 5, 0, 4, 1 << 16)[0] == api.ICER_TOO_MANY_STAGES == oracle.compress([img], 5, 0, 4, 1 << 16)[0]
    out = api.icer_output_data_buf_typedef()
    buf = np.zeros(100, np.uint8)
    assert api.icer_init_output_struct(out, buf, 100, 51) == api.ICER_OUTPUT_BUF_TOO_SMALL


@pytest.mark.parametrize("shift", [1, 2, 3])
def test_device_frames_at_odd_alignments(oracle, shift):
    """icerx_encode_device on frames whose first sample sits 2, 4 or 6 bytes past an 8-byte boundary: the transform's interior path
    (64-bit loads, round 5) must step aside for the generic one -- same streams, same coefficient planes (filter A and C, a size whose
    tiles are interior ones when the pointer is aligned)"""
    import torch
    w, h, st, sg, n = 640, 512, 3, 4, 2
    frames = synth.gray_batch(n, w, h, 77, 1)
    dev = torch.device("cuda", 0)
    flat = torch.zeros(n * w * h + 8, dtype=torch.int16, device=dev)
    view = flat[shift: shift + n * w * h].view(n, h, w)
    view.copy_(torch.from_numpy(frames.view(np.int16)).to(dev))
    assert view.data_ptr() % 8 == 2 * shift
    quota = 2 * w * h
    out = torch.empty((n, quota), dtype=torch.uint8, device=dev)
    sizes = torch.zeros(n, dtype=torch.int64, device=dev)
    rcs = torch.zeros(n, dtype=torch.int32, device=dev)
    for filt in (0, 2):
        enc = api.Encoder(w, h, 1, st, filt, sg, max_frames=n)
        enc.encode_torch(view, quota, out, sizes, rcs)
        torch.cuda.synchronize()
        for k in range(n):
            rc, stream, planes = oracle.compress([frames[k]], st, filt, sg, quota)
            assert int(rcs[k]) == rc and out[k, : int(sizes[k])].cpu().numpy().tobytes() == stream, (shift, filt, k)
            assert np.array_equal(enc.coefficients(k, 0), planes[0])
        enc.close()


def test_batch_extension_equals_per_frame_calls(oracle):
    w, h, st, sg = 256, 192, 3, 6
    frames = synth.gray_batch(5, w, h, 100, 1)
    enc = api.Encoder(w, h, 1, st, 0, sg, max_frames=5)
    for quota in (2 * w * h, 9000):
        got = enc.encode_host(frames, quota)
        for k in range(5):
            rc, stream, planes = oracle.compress([frames[k]], st, 0, sg, quota)
            assert got[k] == (rc, stream)
            if quota > 9000:
                assert np.array_equal(enc.coefficients(k, 0), planes[0])
    enc.close()


def test_batch_routes_blank_units_to_the_small_coder(oracle):
    """in a launch of several frames the all-but-blank coding units (the upper bit planes) are coded by the workgroup coder's
    small instance beside the pipeline kernel (HISTORY.md 4.1c): the streams are the reference's, and the counters show
    that the routing really took place -- and that a single frame goes through the pipeline alone"""
    w, h, st, sg, n = 512, 384, 3, 8, 4
    frames = synth.gray_batch(n, w, h, 7, 1)
    enc = api.Encoder(w, h, 1, st, 0, sg, max_frames=n)
    got = enc.encode_host(frames, 2 * w * h)
    for k in range(n):
        rc, stream, _ = oracle.compress([frames[k]], st, 0, sg, 2 * w * h)
        assert got[k] == (rc, stream)
    r = enc.routing()
    units = enc.info()["units_per_frame"]
    assert r["routed_calls"] == 1 and n * units // 10 < r["routed_units"] < n * units, (r, units)
    # the routing decision itself (it cannot be seen in the streams): round 5 takes a unit's number of blank chunks from the
    # family histogram chunk_sig_kernel leaves; the round-4 kernel, which counted them in the chunk table per unit, routed
    # exactly these many units of these frames (tools/gpu/r05_route_check.py, both libraries on the same box)
    assert r["routed_units"] == 866, r
    assert enc.encode_host(frames[:1], 2 * w * h)[0] == got[0]
    assert enc.routing() == r                                      # one frame: no routing
    assert enc.stats()["unit_timeouts"] == 0
    enc.close()


@pytest.mark.parametrize("w,h,st,sg,n,seed,routed", [(2048, 2048, 4, 16, 4, 12345, 4368), (1024, 768, 2, 2, 3, 11, 183), (640, 480, 4, 5, 6, 3, 811)])
def test_routing_counts_equal_the_per_unit_count_of_round_4(w, h, st, sg, n, seed, routed):
    """the units route_units_kernel sends to the list kernel, from the family histograms: the same numbers the round-4 kernel (a
    workgroup per unit counting blank chunks in the chunk table) gave on these frames (tools/gpu/r05_route_check.py)"""
    frames = synth.gray_batch(n, w, h, seed, 1)
    enc = api.Encoder(w, h, 1, st, 0, sg, max_frames=n)
    enc.encode_host(frames, 2 * w * h)
    r = enc.routing()
    enc.close()
    assert r["routed_units"] == routed, r


def test_launch_info_reports_the_shape_of_a_launch(oracle, monkeypatch):
    """icerx_encoder_launch_info: a lone gray frame whose units are large enough is cut into sub-ranges (ICER_HIP_SPLIT chunks
    each; here 128 so that a small frame qualifies), 8-wave workgroups, window coder beside it; a batch is not split; the
    streams are the reference's either way"""
    monkeypatch.setenv("ICER_HIP_SPLIT", "128")
    w, h, st, sg = 1024, 768, 2, 2
    frames = synth.gray_batch(3, w, h, 11, 1)
    want = [oracle.compress([frames[k]], st, 0, sg, 2 * w * h) for k in range(3)]
    enc = api.Encoder(w, h, 1, st, 0, sg, max_frames=3)
    assert enc.encode_host(frames[:1], 2 * w * h)[0] == (want[0][0], want[0][1])
    li = enc.launch_info()
    assert li["split"] and li["sub_range_workgroups"] > 0 and li["pipeline_waves"] == 8 and li["window_coder_beside"], li
    got = enc.encode_host(frames, 2 * w * h)
    assert [g for g in got] == [(r[0], r[1]) for r in want]
    li = enc.launch_info()
    assert not li["split"] and li["sub_range_workgroups"] == 0 and li["pipeline_waves"] == 8, li
    enc.close()


def test_batch_color(oracle):
    w, h = 128, 128
    fr = np.stack([np.stack(synth.color_frame_yuv(w, h, 50 + k)) for k in range(3)])
    enc = api.Encoder(w, h, 3, 3, 0, 5, max_frames=3)
    got = enc.encode_host(fr, 12000)
    for k in range(3):
        rc, stream, _ = oracle.compress(list(fr[k]), 3, 0, 5, 12000)
        assert got[k] == (rc, stream)
    enc.close()


def test_plain_c_program_links_and_matches(oracle, tmp_path):
    """A C program written against the lib_icer call sequence, compiled with gcc against include/icer_hip.h and
    linked with libicer_hip.so (tests/c_abi/dropin_example.c): streams, return codes and the in-place side effect
    on the planes equal the oracle's."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "dropin_example")
    libdir = os.path.join(root, "icer_compression_amd")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c_abi", "dropin_example.c"),
                           "-L", libdir, "-licer_hip", "-Wl,-rpath," + libdir, "-o", exe])
    for planes, st, f, sg, q in [([synth.gray_frame(320, 200, 5, 1)], 3, 0, 7, 2 * 320 * 200),
                                 ([synth.gray_frame(128, 128, 6, 0)], 4, 2, 16, 6000),
                                 (list(synth.color_frame_yuv(160, 120, 7)), 3, 1, 4, 15000)]:
        h, w = planes[0].shape
        np.stack(planes).astype("<u2").tofile(tmp_path / "in.raw")
        r = subprocess.run([exe, str(tmp_path / "in.raw"), str(w), str(h), str(len(planes)), str(st), str(f), str(sg), str(q),
                            str(tmp_path / "out.bin"), str(tmp_path / "coef.raw")], capture_output=True, text=True)
        rc, stream, left = oracle.compress(planes, st, f, sg, q)
        assert r.returncode == (0 if rc == 0 else 5), r.stdout + r.stderr
        assert f"rc={rc} size_used={len(stream)}" in r.stdout
        assert (tmp_path / "out.bin").read_bytes() == stream
        assert np.array_equal(np.fromfile(tmp_path / "coef.raw", "<u2").reshape(len(planes), h, w), np.stack(left))


def test_batch_over_devices_and_two_threads_from_c(oracle, tmp_path):
    """icerx_compress_batch_uint16 (a batch over the node's GPUs, one host thread + encoder per device) and two
    icerx_encoders driven from two pthreads at once (two devices when there are two) from plain C
    (tests/c_abi/batch_devices.c): all ways give the oracle's per-frame streams."""
    import os
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "batch_devices")
    libdir = os.path.join(root, "icer_compression_amd")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-pthread", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c_abi", "batch_devices.c"),
                           "-L", libdir, "-licer_hip", "-Wl,-rpath," + libdir, "-o", exe])
    n, w, h, st, f, sg = 5, 200, 144, 3, 0, 6
    q = 2 * w * h
    frames = synth.gray_batch(n, w, h, 31, 1)
    frames.astype("<u2").tofile(tmp_path / "in.raw")
    # (second run: three LOGICAL devices on the node's physical ones -- the batch call's per-device threads and pipelines, and
    # the second hand-driven encoder on "device 1", without a second GPU)
    for virtual in (None, "3"):
        env = dict(os.environ)
        if virtual:
            env["ICER_HIP_VIRTUAL_DEVICES"] = virtual
        r = subprocess.run([exe, str(tmp_path / "in.raw"), str(n), str(w), str(h), str(st), str(f), str(sg), str(q), str(tmp_path / "out.bin")],
                           capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 0 and f"ok frames={n}" in r.stdout, r.stdout + r.stderr
        if virtual:
            assert "devices=3" in r.stdout and "second_thread_device=1" in r.stdout, r.stdout
        blob = (tmp_path / "out.bin").read_bytes()
        off = 0
        for k in range(n):
            size, rc = struct.unpack_from("<Qi", blob, off)
            off += 12
            want = oracle.compress([frames[k]], st, f, sg, q)
            assert (rc, blob[off: off + size]) == (want[0], want[1]), k
            off += size


def test_async_halves_equal_the_synchronous_call(oracle):
    """icerx_encode_device_async + icerx_encoder_wait: same streams as icerx_encode_device; a second call while one is
    pending is refused; a wait without a pending call is a no-op"""
    import torch
    dev = torch.device("cuda", 0)
    w, h, st, sg, n = 320, 200, 3, 7, 3
    quota = 2 * w * h
    frames = synth.gray_batch(n, w, h, 77, 1)
    d_frames = torch.from_numpy(frames.view(np.int16)).to(dev)
    out = torch.zeros((n, quota), dtype=torch.uint8, device=dev)
    sizes = torch.zeros(n, dtype=torch.int64, device=dev)
    rcs = torch.zeros(n, dtype=torch.int32, device=dev)
    enc = api.Encoder(w, h, 1, st, 0, sg, max_frames=n)
    enc.wait()
    side = torch.cuda.Stream(dev)
    args = (d_frames.data_ptr(), n, quota, out.data_ptr(), out.stride(0), sizes.data_ptr(), rcs.data_ptr(), side.cuda_stream)
    torch.cuda.synchronize()
    enc.encode_device_async_ptrs(*args)
    with pytest.raises(api.IcerHipError):
        enc.encode_device_async_ptrs(*args)
    with pytest.raises(api.IcerHipError):
        enc.encode_device_ptrs(*args)
    enc.wait()
    enc.wait()
    for k in range(n):
        rc, stream, _ = oracle.compress([frames[k]], st, 0, sg, quota)
        assert int(rcs[k]) == rc and out[k, : int(sizes[k])].cpu().numpy().tobytes() == stream
    enc.close()


def test_two_encoders_take_launches_in_turns(oracle):
    """the streaming pattern of INTEGRATION.md 4 / bench.py `two_launches_in_flight`: two encoders, a plain HIP stream each,
    launch k is submitted before the host waits for launch k - 1; every stream of every launch equals the oracle's"""
    import ctypes
    import torch
    dev = torch.device("cuda", 0)
    w, h, st, sg, n, launches = 448, 320, 4, 9, 4, 7
    quota = 2 * w * h
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    streams = [ctypes.c_void_p() for _ in range(2)]
    for s_ in streams:
        assert hip.hipStreamCreateWithFlags(ctypes.byref(s_), 1) == 0
    encs = [api.Encoder(w, h, 1, st, 0, sg, max_frames=n) for _ in range(2)]
    frames = [synth.gray_batch(n, w, h, 500 + 10 * k, 1) for k in range(launches)]
    d_frames = [torch.from_numpy(f.view(np.int16)).to(dev) for f in frames]
    outs = [torch.zeros((n, quota), dtype=torch.uint8, device=dev) for _ in range(launches)]
    sizes = [torch.zeros(n, dtype=torch.int64, device=dev) for _ in range(launches)]
    rcs = [torch.full((n,), -99, dtype=torch.int32, device=dev) for _ in range(launches)]
    torch.cuda.synchronize()
    for k in range(launches):
        encs[k & 1].encode_device_async_ptrs(d_frames[k].data_ptr(), n, quota, outs[k].data_ptr(), outs[k].stride(0), sizes[k].data_ptr(), rcs[k].data_ptr(),
                                             streams[k & 1].value)
        if k:
            encs[(k - 1) & 1].wait()                   # (launch k - 1 is complete; launch k is in flight)
            for i in range(n):                           # ... and may be consumed while launch k runs
                rc, stream, _ = oracle.compress([frames[k - 1][i]], st, 0, sg, quota)
                assert int(rcs[k - 1][i]) == rc and outs[k - 1][i, : int(sizes[k - 1][i])].cpu().numpy().tobytes() == stream, (k - 1, i)
    encs[(launches - 1) & 1].wait()
    for i in range(n):
        rc, stream, _ = oracle.compress([frames[-1][i]], st, 0, sg, quota)
        assert int(rcs[-1][i]) == rc and outs[-1][i, : int(sizes[-1][i])].cpu().numpy().tobytes() == stream
    for e in encs:
        e.close()
    for s_ in streams:
        assert hip.hipStreamDestroy(s_) == 0


@pytest.mark.parametrize("sub", ["1", "2", "5", "", "1/2"])
def test_host_batch_is_pipelined_in_sub_batches_and_exact(oracle, monkeypatch, sub):
    """icerx_compress_batch_uint16_devices: a device's block goes through copy-in / kernels / copy-out streams in
    sub-batches over three buffer sets with an encoder each (ICER_HIP_BATCH_SUB pins their size: whole, ragged last one,
    one frame; "1/2": one frame per sub-batch over two sets, ICER_HIP_BATCH_SETS); pinned and pageable caller
    memory; the pooled encoder is re-used by the next call and re-made for another geometry; a row too short for its
    stream is an error, never an overrun"""
    if sub:
        monkeypatch.setenv("ICER_HIP_BATCH_SUB", sub.split("/")[0])
    if "/" in sub:
        monkeypatch.setenv("ICER_HIP_BATCH_SETS", sub.split("/")[1])
    n, w, h, st, sg = 5, 256, 192, 3, 6
    quota = 2 * w * h
    frames = synth.gray_batch(n, w, h, 303, 1)
    want = [oracle.compress([frames[k]], st, 0, sg, quota) for k in range(n)]
    for pinned in (False, True):
        out = np.full((n, quota), 0xAB, np.uint8)
        sizes, rcs = np.zeros(n, np.uint64), np.zeros(n, np.int32)
        if pinned:
            assert api.pin_host(frames) and api.pin_host(out)
        for _ in range(2):
            assert api.compress_batch(frames, st, 0, sg, quota, out, sizes, rcs, devices=[0]) == 0, api.load_library().icerx_last_error()
            for k in range(n):
                assert (int(rcs[k]), out[k, : int(sizes[k])].tobytes()) == (want[k][0], want[k][1]), k
                assert (out[k, int(sizes[k]):] == 0xAB).all()           # exactly size bytes per row are written
        if pinned:
            api.unpin_host(frames); api.unpin_host(out)
    # another geometry, colour, quota cut
    yuv = np.stack([np.stack(synth.color_frame_yuv(160, 96, 5 + k)) for k in range(3)])
    out = np.zeros((3, 9000), np.uint8)
    sizes, rcs = np.zeros(3, np.uint64), np.zeros(3, np.int32)
    assert api.compress_batch(yuv, 2, 0, 4, 9000, out, sizes, rcs, devices=[0]) == 0
    for k in range(3):
        rc, stream, _ = oracle.compress(list(yuv[k]), 2, 0, 4, 9000)
        assert (int(rcs[k]), out[k, : int(sizes[k])].tobytes()) == (rc, stream)
    # rows shorter than the streams
    small = np.zeros((n, 1000), np.uint8)
    sizes, rcs = np.zeros(n, np.uint64), np.zeros(n, np.int32)
    assert api.compress_batch(frames, st, 0, sg, quota, small, sizes, rcs, devices=[0]) == api.ICER_OUTPUT_BUF_TOO_SMALL
    api.load_library().icerx_batch_release()


def test_host_batch_slot_retry_inside_the_pipeline(oracle, monkeypatch):
    """a coding unit that outgrows its provisioned slot in the middle of a pipelined host batch: that sub-batch is coded
    again with larger slots, the ones around it are unaffected"""
    monkeypatch.setenv("ICER_HIP_SLOT_BPP", "1")
    monkeypatch.setenv("ICER_HIP_BATCH_SUB", "2")
    n, w, h = 6, 256, 256
    frames = synth.gray_batch(n, w, h, 9, 1)
    frames[3] = synth.gray_batch(1, w, h, 9, 0)[0]               # noise: needs more than 1 bit per pixel
    out = np.zeros((n, 1 << 18), np.uint8)
    sizes, rcs = np.zeros(n, np.uint64), np.zeros(n, np.int32)
    assert api.compress_batch(frames, 1, 0, 1, 1 << 18, out, sizes, rcs, devices=[0]) == 0, api.load_library().icerx_last_error()
    for k in range(n):
        rc, stream, _ = oracle.compress([frames[k]], 1, 0, 1, 1 << 18)
        assert (int(rcs[k]), out[k, : int(sizes[k])].tobytes()) == (rc, stream), k
    assert api.process_stats()["slot_retries"] >= 1
    api.load_library().icerx_batch_release()


def test_encode_host_row_length_is_checked_for_one_frame_too(oracle):
    frames = synth.gray_batch(1, 128, 128, 3, 0)
    enc = api.Encoder(128, 128, 1, 2, 0, 2, max_frames=1)
    out = np.zeros((1, 500), np.uint8)
    sizes, rcs = np.zeros(1, np.uint64), np.zeros(1, np.int32)
    rc = enc.lib.icerx_encode_host(enc.handle, frames.ctypes.data, 1, 1 << 16, out.ctypes.data, 500, sizes.ctypes.data, rcs.ctypes.data)
    assert rc == api.ICER_OUTPUT_BUF_TOO_SMALL and int(sizes[0]) > 500 and not out.any()
    big = np.zeros((1, int(sizes[0])), np.uint8)                   # a row of exactly the stream's length is enough
    assert enc.lib.icerx_encode_host(enc.handle, frames.ctypes.data, 1, 1 << 16, big.ctypes.data, big.shape[1], sizes.ctypes.data, rcs.ctypes.data) == 0
    assert big[0].tobytes() == oracle.compress([frames[0]], 2, 0, 2, 1 << 16)[1]
    enc.close()


def test_async_overlap_by_hand_from_c(oracle, tmp_path):
    """tests/c_abi/async_overlap.c: two encoders, each with a stream and device buffers of its own, driven from plain C
    through icerx_encode_device_async / icerx_encoder_wait with the HIP runtime's C API for the copies"""
    import os
    import struct
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "async_overlap")
    libdir = os.path.join(root, "icer_compression_amd")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_abi", "async_overlap.c"), "-L", libdir, "-licer_hip", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    n, w, h, st, f, sg = 7, 224, 160, 3, 0, 5
    q = 2 * w * h
    frames = synth.gray_batch(n, w, h, 57, 1)
    frames.astype("<u2").tofile(tmp_path / "in.raw")
    r = subprocess.run([exe, str(tmp_path / "in.raw"), str(n), str(w), str(h), str(st), str(f), str(sg), str(q), str(tmp_path / "out.bin")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and f"ok frames={n}" in r.stdout, r.stdout + r.stderr
    blob = (tmp_path / "out.bin").read_bytes()
    off = 0
    for k in range(n):
        size, rc = struct.unpack_from("<Qi", blob, off)
        off += 12
        want = oracle.compress([frames[k]], st, f, sg, q)
        assert (rc, blob[off: off + size]) == (want[0], want[1]), k
        off += size


def test_frontend_fusion_u8_and_rgb8(oracle):
    """next-3: 8-bit gray widening and packed RGB888 -> YCbCr on the device give the same streams as the
    reference callers' host-side conversion followed by the uint16 encoders."""
    import torch
    dev = torch.device("cuda", 0)
    w, h, st, sg, quota = 192, 160, 3, 5, 2 * 192 * 160
    gray = synth.gray_batch(2, w, h, 21, 0).astype(np.uint8)
    enc = api.Encoder(w, h, 1, st, 0, sg, max_frames=2)
    out = torch.empty((2, quota), dtype=torch.uint8, device=dev)
    sizes = torch.zeros(2, dtype=torch.int64, device=dev)
    rcs = torch.zeros(2, dtype=torch.int32, device=dev)
    enc.encode_torch_frontend(torch.from_numpy(gray).to(dev), quota, out, sizes, rcs)
    for k in range(2):
        rc, stream, _ = oracle.compress([gray[k].astype(np.uint16)], st, 0, sg, quota)
        assert int(rcs[k]) == rc and out[k, : int(sizes[k])].cpu().numpy().tobytes() == stream
    enc.close()

    rng = np.random.default_rng(4)
    rgb = rng.integers(0, 256, (2, h, w, 3)).astype(np.uint8)
    rgb[1] = (np.linspace(0, 255, w)[None, :, None] * np.ones((h, 1, 3))).astype(np.uint8)      # smooth ramp
    enc = api.Encoder(w, h, 3, st, 0, sg, max_frames=2)
    enc.encode_torch_frontend(torch.from_numpy(rgb).to(dev), 20000, out, sizes, rcs)
    for k in range(2):
        r, g, b = (rgb[k, :, :, c].astype(np.int64) for c in range(3))
        clip = lambda v: np.clip(v, 0, 255)
        y = clip((19595 * r + 38470 * g + 7471 * b) >> 16)
        cb = clip(((36962 * (b - y)) >> 16) + 128)
        cr = clip(((46727 * (r - y)) >> 16) + 128)
        rc, stream, _ = oracle.compress([p.astype(np.uint16) for p in (y, cb, cr)], st, 0, sg, 20000)
        assert int(rcs[k]) == rc and out[k, : int(sizes[k])].cpu().numpy().tobytes() == stream
    enc.close()


def test_slot_bound_retry(oracle, monkeypatch):
    """Per-unit payload slots are provisioned at a bits-per-pixel bound; a unit that needs more must be
    noticed and the batch redone with larger slots.  Forced here by starting from 1 bit/pixel on noise."""
    monkeypatch.setenv("ICER_HIP_SLOT_BPP", "1")
    frames = synth.gray_batch(2, 256, 256, 9, 0)
    enc = api.Encoder(256, 256, 1, 1, 0, 1, max_frames=2)
    assert enc.info()["slot_bits_per_pixel"] == 1
    got = enc.encode_host(frames, 1 << 18)
    assert enc.info()["slot_bits_per_pixel"] > 1            # the retry happened
    for k in range(2):
        rc, stream, _ = oracle.compress([frames[k]], 1, 0, 1, 1 << 18)
        assert got[k] == (rc, stream)
    enc.close()
    assert _same(api.compress([np.zeros((64, 64), np.uint16)], 1, 0, 1, 1 << 16),
                 oracle.compress([np.zeros((64, 64), np.uint16)], 1, 0, 1, 1 << 16))


GOLDEN_ON_GPU = ["kat_512_m1", "kat_512_m0", "kat_512_quota30000", "kat_odd_517x389_filtC", "kat_1024_16seg",
                 "kat_color_512_quota", "C2_4096_gray_5st_10seg", "C2_4096_gray_noise", "C3_4096_yuv_quota70000",
                 "C4_2048_frame0", "C4_2048_frame1", "C5_8192_frame0"]


@pytest.mark.parametrize("name", GOLDEN_ON_GPU)
def test_golden_vectors_full_size(golden, name):
    g = golden[name]
    planes = [synth.gray_frame(g["w"], g["h"], g["seed"], g["mode"])] if g["kind"] == "gray" else \
        list(synth.color_frame_yuv(g["w"], g["h"], g["seed"]))
    enc = api.Encoder(g["w"], g["h"], len(planes), g["stages"], g["filt"], g["segments"], max_frames=1)
    (rc, stream), = enc.encode_host(np.stack(planes)[None], g["quota"])
    enc.close()
    assert rc == g["rc"] and len(stream) == g["size"]
    assert "%08x" % zlib.crc32(stream) == g["crc32"]
    assert hashlib.sha256(stream).hexdigest()[:16] == g["sha256_16"]


# ---- the production sub-range split through other tables (filter A / 8-bit content were its only goldens) --------
SPLIT_GOLDEN = ["split_4096_filtC", "split_4096_12bit_filtB", "split_4096_u8_gray"]


@pytest.mark.parametrize("name", SPLIT_GOLDEN)
def test_production_split_through_other_tables(golden, name):
    """A lone 4096 x 4096 plane is cut into sub-ranges at the PRODUCTION setting (the piece size plan.hpp auto_split_chunks picks: 2 048 chunks, no env knob): the
    reference build's streams for the filter with the W3 quirk, for 12-bit content through filter B, and for the uint8
    twin's 7 planes -- and the launch really was a split one"""
    import torch
    g = golden[name]
    w, h = g["w"], g["h"]
    u8 = g["kind"] == "gray8"
    plane = synth.gray_frame_u8(w, h, g["seed"], g["mode"]) if u8 else \
        synth.gray_frame_12bit(w, h, g["seed"], g["mode"]) if g["kind"] == "gray12" else synth.gray_frame(w, h, g["seed"], g["mode"])
    enc = api.Encoder(w, h, 1, g["stages"], g["filt"], g["segments"], max_frames=1, sample_bits=8 if u8 else 16)
    if u8:
        (rc, stream), = enc.encode_torch_s8(torch.from_numpy(plane[None]).cuda(), g["quota"])
    else:
        (rc, stream), = enc.encode_host(plane[None, None], g["quota"])
    li, st = enc.launch_info(), enc.stats()
    enc.close()
    assert li["split"] and li["sub_range_workgroups"] > 0, li
    assert st["unit_timeouts"] == 0 and st["fallback_batches"] == 0, st
    assert (rc, len(stream), "%08x" % zlib.crc32(stream), hashlib.sha256(stream).hexdigest()[:16]) == (g["rc"], g["size"], g["crc32"], g["sha256_16"])


@pytest.mark.parametrize("w,h,st,sg,filt,want_subs", [(2048, 2048, 4, 4, 0, True), (3000, 2000, 5, 10, 2, True), (2048, 2048, 4, 16, 0, False)])
def test_split_size_chosen_per_geometry(oracle, w, h, st, sg, filt, want_subs):
    """plan.hpp auto_split_chunks (round 6): a lone frame's sub-range size follows its geometry -- pieces of 1 024 chunks where the
    headline frame takes 2 048 (2048^2 with 4 segments: K = 4 per level-1 unit; 3000 x 2000: K = 2), none where the units are short --
    and the spliced streams are the oracle's"""
    frame = synth.gray_frame(w, h, 4711, 1)
    quota = 2 * w * h
    want = oracle.compress([frame], st, filt, sg, quota)
    enc = api.Encoder(w, h, 1, st, filt, sg, max_frames=1)
    (rc, stream), = enc.encode_host(frame[None, None], quota)
    li, stt = enc.launch_info(), enc.stats()
    enc.close()
    assert (rc, stream) == (want[0], want[1])
    assert li["split"] == want_subs and (li["sub_range_workgroups"] > 0) == want_subs, li
    assert stt["unit_timeouts"] == 0 and stt["fallback_batches"] == 0


# ---- a rank != 0 share of the two batch configurations (frames 32*rank .. / 8*rank ..), both ways -----------------------------
@pytest.mark.parametrize("name,rank", [("C4", 5), ("C5", 3)])
def test_batch_share_of_another_rank(name, rank):
    """What rank `rank` of the 8-GPU job codes -- C4: frames 160..191 of the 256, C5: frames 24..31 of the 64 -- on this GPU:
    device-resident launches (what bench.py times) AND icerx_compress_batch_uint16_devices from page-locked host memory,
    every frame's return code, length and CRC-32 against the reference CPU encoder's (tests/golden/batch_golden.json)"""
    import os
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    dev = torch.device("cuda", 0)
    wl = bench.Workload(name, rank, dev, 0)
    assert wl.first == rank * bench.CONFIGS[name]["per_gpu"] and wl.B == bench.CONFIGS[name]["per_gpu"]
    wl.step()
    bad, nbytes = wl.verify()
    stats = wl.enc.stats()
    wl.close()
    del wl
    torch.cuda.empty_cache()
    assert not bad and nbytes > 0, (name, rank, bad)
    assert stats["unit_timeouts"] == 0, stats
    hw = bench.HostWorkload(name, rank, dev, 0)
    hw.step()
    bad, _ = hw.verify()
    hw.close()
    assert not bad, (name, rank, bad)


@pytest.mark.parametrize("prio", ["0", "1"])
def test_host_batch_in_a_crowded_process(monkeypatch, prio):
    """The host-fed batch call beside streams that are not the library's: two torch streams that have run kernels and stay alive
    and a live decoder (side streams of its own).  The runtime shares its hardware queues out over every live stream, so this is
    where the pipeline's six streams per device double up; what must hold whatever the queue lottery gives: every frame
    bit-exact, no unit time-out, no fall-back.  Both kinds of encoder streams: plain ones (what the library takes when the process
    has GPU_MAX_HW_QUEUES >= 6, as this test process does) and high-priority ones (its choice when hardware queues are scarce)."""
    monkeypatch.setenv("ICER_HIP_STREAM_PRIO", prio)
    api.load_library().icerx_batch_release()                          # (the pooled pipeline is rebuilt with the streams asked for)
    import os
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    from icer_compression_amd import decoder
    dev = torch.device("cuda", 0)
    others = [torch.cuda.Stream(device=dev) for _ in range(2)]
    keep = []
    for st in others:
        with torch.cuda.stream(st):
            keep.append(torch.ones(1 << 18, device=dev) + 1)
    torch.cuda.synchronize(dev)
    c = bench.CONFIGS["C4"]
    dec = decoder.Decoder(1, c["stages"], 0, c["segments"])
    before = api.process_stats()
    hw = bench.HostWorkload("C4", 2, dev, 0)
    try:
        for _ in range(3):
            hw.step()
            with torch.cuda.stream(others[0]):                     # the other streams are busy in between, too
                keep[0].add_(1)
        bad, nbytes = hw.verify()
    finally:
        hw.close()
        dec.close()
    after = api.process_stats()
    assert not bad and nbytes > 0, bad
    assert after["unit_timeouts"] == before["unit_timeouts"] and after["fallback_batches"] == before["fallback_batches"], (before, after)


# ---- the multi-device code path on one GPU: ICER_HIP_VIRTUAL_DEVICES -----------------------------------------------------------
def test_virtual_devices_batch(monkeypatch):
    """ICER_HIP_VIRTUAL_DEVICES=4: icerx_compress_batch_uint16 over four logical devices (four host threads, four pooled
    pipelines of 3 encoders / 5 streams / 3 staging sets on the one physical GPU) on C4-sized frames: frames 0..9 of the C4
    batch against the reference goldens, blocks of 3 + 3 + 2 + 2; a named device list; an encoder on logical device 3; the
    error of every failing device in the aggregated message; without the variable the node has its real device count"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    L = api.load_library()
    real = L.icerx_device_count()
    assert real >= 1
    monkeypatch.setenv("ICER_HIP_VIRTUAL_DEVICES", "4")
    assert L.icerx_device_count() == 4
    c = bench.CONFIGS["C4"]
    n, w, h = 10, c["w"], c["h"]
    quota = 2 * w * h
    frames = synth.gray_batch(n, w, h, synth.DEFAULT_SEED, 1)
    gold = bench.frame_goldens("C4", 0, 0, n)
    out = np.zeros((n, w * h), np.uint8)
    sizes, rcs = np.zeros(n, np.uint64), np.zeros(n, np.int32)

    def check():
        for k in range(n):
            assert int(rcs[k]) == 0 and int(sizes[k]) == gold[k][0] and "%08x" % zlib.crc32(out[k, : gold[k][0]].tobytes()) == gold[k][1], k
    assert api.compress_batch(frames, c["stages"], 0, c["segments"], quota, out, sizes, rcs) == 0, L.icerx_last_error()      # all 4
    check()
    out[:] = 0; sizes[:] = 0
    assert api.compress_batch(frames, c["stages"], 0, c["segments"], quota, out, sizes, rcs, devices=[3, 1]) == 0, L.icerx_last_error()
    check()
    # an encoder on a logical device
    enc = api.Encoder(w, h, 1, c["stages"], 0, c["segments"], max_frames=1, device=3)
    (rc, stream), = enc.encode_host(frames[:1], quota)
    enc.close()
    assert rc == 0 and len(stream) == gold[0][0] and "%08x" % zlib.crc32(stream) == gold[0][1]
    with pytest.raises(api.IcerHipError):
        api.Encoder(w, h, 1, c["stages"], 0, c["segments"], max_frames=1, device=4)
    # every failing device is named: rows too short for the streams on all four
    small = np.zeros((n, 1000), np.uint8)
    assert api.compress_batch(frames, c["stages"], 0, c["segments"], quota, small, sizes, rcs) == api.ICER_OUTPUT_BUF_TOO_SMALL
    msg = L.icerx_last_error().decode()
    assert all(f"device {d}:" in msg for d in range(4)), msg
    # ... and the pipelines are usable again afterwards (the error path drained them)
    out[:] = 0; sizes[:] = 0
    assert api.compress_batch(frames, c["stages"], 0, c["segments"], quota, out, sizes, rcs) == 0, L.icerx_last_error()
    check()
    L.icerx_batch_release()
    monkeypatch.delenv("ICER_HIP_VIRTUAL_DEVICES")
    assert L.icerx_device_count() == real


# ---- uint8 twins ------------------------------------------------------------------------------------
def _u8_planes(g):
    if g["kind"] == "gray8":
        return [synth.gray_frame_u8(g["w"], g["h"], g["seed"], g["mode"])]
    if g["kind"] == "gray8full":
        return [synth.gray_frame(g["w"], g["h"], g["seed"], g["mode"]).astype(np.uint8)]
    return list(synth.color_frame_yuv_u8(g["w"], g["h"], g["seed"]))


U8_GOLDEN = ["u8_512_gray", "u8_517x389_filtB_quota", "u8_2048_gray_4st_16seg", "u8_512_yuv_4st", "u8_512_yuv_quota",
             "u8_256_yuv_5st_packet_table", "u8_512_gray_full_range_overflow"]


@pytest.mark.parametrize("name", U8_GOLDEN)
def test_uint8_twins_golden_vectors(golden, oracle, name):
    """icer_compress_image_uint8 / icer_compress_image_yuv_uint8 of libicer_hip.so against the reference's digests and,
    for the in-place side effect on the image, against the oracle."""
    g = golden[name]
    planes = _u8_planes(g)
    rc, stream, left = api.compress_u8(planes, g["stages"], g["filt"], g["segments"], g["quota"])
    assert rc == g["rc"] and len(stream) == g["size"]
    assert "%08x" % zlib.crc32(stream) == g["crc32"] and hashlib.sha256(stream).hexdigest()[:16] == g["sha256_16"]
    if rc in (0, -5):
        want = oracle.compress_u8(planes, g["stages"], g["filt"], g["segments"], g["quota"])
        assert all(np.array_equal(a, b) for a, b in zip(left, want[2]))


def test_uint8_twins_random_cases(oracle):
    rng = np.random.default_rng(31)
    seen = set()
    for trial in range(60):
        w, h = int(rng.integers(8, 300)), int(rng.integers(8, 300))
        st = int(rng.integers(1, 7))
        while ((w + (1 << st) - 1) >> st) < 3 or ((h + (1 << st) - 1) >> st) < 3:
            st -= 1
        sg = min(int(rng.integers(1, 33)), ((w + (1 << st) - 1) >> st) * ((h + (1 << st) - 1) >> st))
        ch = 3 if trial % 3 == 0 else 1
        amp, base = int(rng.choice([4, 8, 16, 30, 60, 127, 255])), int(rng.choice([0, 10, 40]))
        planes = [np.clip(base + rng.integers(0, amp + 1, (h, w)), 0, 255).astype(np.uint8) for _ in range(ch)]
        quota = int(rng.choice([w * h * 2 + 100, w * h * 2 + 100, 500, 3000]))
        filt = int(rng.integers(0, 7))
        a, b = api.compress_u8(planes, st, filt, sg, quota), oracle.compress_u8(planes, st, filt, sg, quota)
        seen.add(b[0])
        assert a[0] == b[0] and a[1] == b[1], (trial, w, h, st, filt, sg, ch, amp, base, quota, a[0], b[0])
        if b[0] in (0, -5):
            assert all(np.array_equal(p, q) for p, q in zip(a[2], b[2]))
    assert {0, -1, -5} <= seen


def test_uint8_twins_device_resident_batch(oracle):
    """icerx_encoder_create_ex(sample_bits = 8) + icerx_encode_device_s8 on a batch of frames"""
    import torch
    w, h, st, sg, n = 256, 192, 3, 6, 5
    frames = np.stack([synth.gray_frame_u8(w, h, 100 + k, 1) for k in range(n)])
    quota = 2 * w * h
    enc = api.Encoder(w, h, 1, st, 0, sg, max_frames=n, sample_bits=8)
    res = enc.encode_torch_s8(torch.from_numpy(frames).cuda(), quota)
    enc.close()
    for k in range(n):
        rc, stream, _ = oracle.compress_u8([frames[k]], st, 0, sg, quota)
        assert res[k] == (rc, stream)


def test_cli_matches_reference_cli_semantics(oracle, tmp_path):
    """tools/icer_util_hip.c: same options as the reference's `icer_util compress`, same .bin for the
    same pixels (gray widening, RGB -> Y/Cb/Cr, stb_image's channel conversions, quota rule)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "icer_util_hip")
    libdir = os.path.join(root, "icer_compression_amd")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(root, "include"), os.path.join(root, "tools", "icer_util_hip.c"),
                           "-L", libdir, "-licer_hip", "-licer_hip_dec", "-Wl,-rpath," + libdir, "-o", exe])
    rng = np.random.default_rng(9)
    w, h = 200, 144
    gray = synth.gray_frame(w, h, 3, 1).astype(np.uint8)
    rgb = np.stack([synth.gray_frame(w, h, 10 + c, 1).astype(np.uint8) for c in range(3)], axis=-1)
    rgb[::7, ::5] = rng.integers(0, 256, rgb[::7, ::5].shape)
    (tmp_path / "g.pgm").write_bytes(b"P5\n# comment\n%d %d\n255\n" % (w, h) + gray.tobytes())
    (tmp_path / "c.ppm").write_bytes(b"P6 %d %d 255\n" % (w, h) + rgb.tobytes())
    stride = (3 * w + 3) // 4 * 4
    rows = np.zeros((h, stride), np.uint8)
    rows[:, : 3 * w] = rgb[::-1, :, ::-1].reshape(h, 3 * w)                   # bottom-up, BGR
    hdr = b"BM" + (54 + stride * h).to_bytes(4, "little") + bytes(4) + (54).to_bytes(4, "little") + (40).to_bytes(4, "little") + \
        w.to_bytes(4, "little") + h.to_bytes(4, "little") + (1).to_bytes(2, "little") + (24).to_bytes(2, "little") + bytes(24)
    (tmp_path / "c.bmp").write_bytes(hdr + rows.tobytes())

    def ycbcr(img):
        r, g, b = (img[:, :, c].astype(np.int64) for c in range(3))
        clip = lambda v: np.clip(v, 0, 255)
        y = clip((19595 * r + 38470 * g + 7471 * b) >> 16)
        return [p.astype(np.uint16) for p in (y, clip(((36962 * (b - y)) >> 16) + 128), clip(((46727 * (r - y)) >> 16) + 128))]

    luma = lambda img: ((77 * img[:, :, 0].astype(np.uint32) + 150 * img[:, :, 1].astype(np.uint32) + 29 * img[:, :, 2].astype(np.uint32)) >> 8).astype(np.uint16)
    cases = [("g.pgm", [], [gray.astype(np.uint16)], 4, 0, 6, w * h),                                 # defaults of the reference CLI
             ("g.pgm", ["-s", "3", "-f", "C", "-g", "9", "-t", "4000"], [gray.astype(np.uint16)], 3, 2, 9, 4000),
             ("g.pgm", ["--color", "-s", "2"], ycbcr(np.repeat(gray[:, :, None], 3, axis=2)), 2, 0, 6, 3 * w * h),
             ("c.ppm", ["--stages", "3", "--segments", "4"], ycbcr(rgb), 3, 0, 4, 3 * w * h),
             ("c.ppm", ["-G", "-f", "q"], [luma(rgb)], 4, 6, 6, w * h),
             ("c.bmp", ["-t", "9000", "-g", "3"], ycbcr(rgb), 4, 0, 3, 9000)]
    for name, opts, planes, st, f, sg, q in cases:
        r = subprocess.run([exe, "compress", str(tmp_path / name), str(tmp_path / "o.bin")] + opts, capture_output=True, text=True)
        rc, stream, _ = oracle.compress(planes, st, f, sg, q)
        assert r.returncode == 0 and rc in (0, -5), r.stdout + r.stderr
        assert (tmp_path / "o.bin").read_bytes() == stream, (name, opts)
    assert subprocess.run([exe, "decompress", str(tmp_path / "o.bin"), "b"], capture_output=True).returncode == 1      # needs -c or -G, like the reference's
    assert subprocess.run([exe, "compress", str(tmp_path / "g.pgm"), str(tmp_path / "o.bin"), "-c", "-G"], capture_output=True).returncode == 1


def test_progressive_mode_batch_with_different_cut_points(oracle):
    """Small byte quotas (units launched in priority order, stopping once the quota is spent): frames of one batch
    whose cuts fall at very different places, one of them an all-zero frame, plus a YUV batch."""
    w, h, st, sg = 384, 320, 4, 9
    frames = np.stack([synth.gray_frame(w, h, 50, 0), synth.gray_frame(w, h, 51, 1), np.zeros((h, w), np.uint16),
                       (synth.gray_frame(w, h, 52, 1) // 16).astype(np.uint16), synth.gray_frame(w, h, 53, 0)])
    for quota in (300, 2500, 20000, w * h // 2 - 1):
        enc = api.Encoder(w, h, 1, st, 0, sg, max_frames=len(frames))
        res = enc.encode_host(frames[:, None], quota)
        enc.close()
        for k in range(len(frames)):
            rc, stream, _ = oracle.compress([frames[k]], st, 0, sg, quota)
            assert res[k] == (rc, stream), (quota, k, res[k][0], rc, len(res[k][1]), len(stream))
    planes = [np.stack(synth.color_frame_yuv(256, 192, 60 + k)) for k in range(3)]
    enc = api.Encoder(256, 192, 3, 3, 1, 5, max_frames=3)
    res = enc.encode_host(np.stack(planes), 9000)
    enc.close()
    for k in range(3):
        rc, stream, _ = oracle.compress(list(planes[k]), 3, 1, 5, 9000)
        assert res[k] == (rc, stream)


def test_round_trip_through_the_reference_decoder(reference):
    """Encode on the GPU, decode with the reference's own decoder (oracle/_ref), compare with the input: the
    size-independent property at BASELINE configs[1] size (4096 x 4096, lossless) and on a YUV frame."""
    for planes, st, f, sg in [([synth.gray_frame(4096, 4096, 12345, 1)], 5, 0, 10), ([synth.gray_frame(517, 389, 9, 1)], 4, 4, 7),
                              (list(synth.color_frame_yuv(512, 384, 4)), 4, 0, 10)]:
        h, w = planes[0].shape
        enc = api.Encoder(w, h, len(planes), st, f, sg, max_frames=1)
        (rc, stream), = enc.encode_host(np.stack(planes)[None], 2 * w * h * len(planes))
        enc.close()
        drc, back = reference.decompress(stream, len(planes), st, f, sg)
        assert rc == 0 and drc == 0 and all(np.array_equal(a, b) for a, b in zip(back, planes))


# ---- real image content: the reference repository's own fixtures (BASELINE configs[0]; tests/golden/fixture_*) ---------------
from tests.test_reference_fixtures import FIXTURE_GOLDEN, fixture_planes  # noqa: E402


@pytest.mark.parametrize("name", sorted(FIXTURE_GOLDEN))
def test_reference_fixtures_through_the_drop_in_entry_points(name):
    """boat.512 / boatcolor.512 with the parameters of the reference's CLI and example programs through
    icer_compress_image_[yuv_]uint16: the reference build's streams (seven filters: the CLI binary's own known answers,
    SURVEY 8c), then decoded by libicer_hip_dec.so: the reference decoder's images"""
    from icer_compression_amd import decoder
    g = FIXTURE_GOLDEN[name]
    planes = fixture_planes(name)
    rc, stream, _ = api.compress(planes, g["stages"], g["filt"], g["segments"], g["quota"])
    assert (rc, len(stream), "%08x" % zlib.crc32(stream), hashlib.sha256(stream).hexdigest()[:16]) == (g["rc"], g["size"], g["crc32"], g["sha256_16"])
    drc, w, h, back = decoder.decompress(stream, g["channels"], g["stages"], g["filt"], g["segments"])
    hsh = hashlib.sha256()
    for p in back:
        hsh.update(p.tobytes())
    assert (drc, w, h, hsh.hexdigest()[:16]) == (g["decoded_rc"], g["w"], g["h"], g["decoded_sha256_16"])


def test_reference_fixtures_through_the_command_line_tool(tmp_path):
    """the same files through tools/icer_util_hip (the reference's `icer_util compress` flags): boat.512 as PGM with
    --grayscale -s 3 -g 10 -f A..Q, boatcolor.512 as PPM with -s 4 -g 10; and `decompress` of the results"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "icer_util_hip")
    libdir = os.path.join(root, "icer_compression_amd")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(root, "include"), os.path.join(root, "tools", "icer_util_hip.c"),
                           "-L", libdir, "-licer_hip", "-licer_hip_dec", "-Wl,-rpath," + libdir, "-o", exe])
    z = np.load(os.path.join(root, "tests", "golden", "fixture_planes.npz"))
    (tmp_path / "boat.pgm").write_bytes(b"P5 512 512 255\n" + z["boat512_gray"].tobytes())
    (tmp_path / "boatcolor.ppm").write_bytes(b"P6 512 512 255\n" + z["boatcolor512_rgb"].tobytes())
    for name, g in sorted(FIXTURE_GOLDEN.items()):
        if not name.startswith("cli_"):
            continue
        src = "boat.pgm" if g["channels"] == 1 else "boatcolor.ppm"
        opts = (["--grayscale"] if g["channels"] == 1 else []) + ["-s", str(g["stages"]), "-g", str(g["segments"]), "-f", "ABCDEFQ"[g["filt"]]]
        r = subprocess.run([exe, "compress", str(tmp_path / src), str(tmp_path / "o.bin")] + opts, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        blob = (tmp_path / "o.bin").read_bytes()
        assert (len(blob), hashlib.sha256(blob).hexdigest()[:16]) == (g["size"], g["sha256_16"]), name
        # and back: the tool writes PGM / PPM; lossless for every filter but C (quirk W3)
        out = "back.pgm" if g["channels"] == 1 else "back.ppm"
        r = subprocess.run([exe, "decompress", str(tmp_path / "o.bin"), str(tmp_path / out)] + opts + ([] if g["channels"] == 1 else ["--color"]),
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        back = (tmp_path / out).read_bytes()
        if g["channels"] == 1 and g["filt"] != 2:
            assert back.endswith(z["boat512_gray"].tobytes())
