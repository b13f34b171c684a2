"""The unit time-out recovery path on hardware (api.hip encode_verdict / report_timeouts).

A coding unit of the eight-wave pipeline that gives up at a bounded spin reports kUnitFailed; the host prints where, counts it and
codes the batch again with the barrier-only window coder.  No real time-out has occurred in millions of stress encodes, so the path is
forced: ICER_HIP_TEST_FAIL_UNIT=<frame>:<unit>[:<calls>] (read when an encoder is created) makes code_units_kernel report a time-out for
that unit in the next <calls> pipeline launches -- the unit itself is coded normally, only its verdict is replaced.  The streams must
still be the oracle's, the event must be counted and reported, and the call after that must be back on the pipeline."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from icer_compression_amd import api, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _encoder_with_hook(spec, *args, **kw):
    os.environ["ICER_HIP_TEST_FAIL_UNIT"] = spec
    try:
        return api.Encoder(*args, **kw)
    finally:
        del os.environ["ICER_HIP_TEST_FAIL_UNIT"]


def _run(enc, d_frames, n, quota, out, sizes, rcs, use_async):
    import torch
    if use_async:
        side = torch.cuda.Stream(d_frames.device)
        enc.encode_device_async_ptrs(d_frames.data_ptr(), n, quota, out.data_ptr(), out.stride(0), sizes.data_ptr(), rcs.data_ptr(), side.cuda_stream)
        enc.wait()
    else:
        enc.encode_torch(d_frames, quota, out, sizes, rcs)
    torch.cuda.synchronize()


@pytest.mark.parametrize("use_async", [False, True])
def test_forced_timeout_is_reported_counted_and_recoded(oracle, capfd, use_async):
    """a batch of six frames (the synchronous call enqueues it in two parts: the failing frame is in the second one)"""
    import torch
    dev = torch.device("cuda", 0)
    w, h, st, sg, n = 320, 256, 3, 6, 6
    quota = 2 * w * h
    frames = synth.gray_batch(n, w, h, 505, 1)
    want = [oracle.compress([frames[k]], st, 0, sg, quota) for k in range(n)]
    d_frames = torch.from_numpy(frames.view(np.int16)).to(dev)
    out = torch.zeros((n, quota), dtype=torch.uint8, device=dev)
    sizes = torch.zeros(n, dtype=torch.int64, device=dev)
    rcs = torch.zeros(n, dtype=torch.int32, device=dev)
    enc = _encoder_with_hook("4:7", w, h, 1, st, 0, sg, max_frames=n)
    before, pbefore = enc.stats(), api.process_stats()
    capfd.readouterr()
    _run(enc, d_frames, n, quota, out, sizes, rcs, use_async)
    err = capfd.readouterr().err
    for k in range(n):
        assert int(rcs[k]) == want[k][0] and out[k, : int(sizes[k])].cpu().numpy().tobytes() == want[k][1], k
    after = enc.stats()
    assert after["unit_timeouts"] == before["unit_timeouts"] + 1 and after["fallback_batches"] == before["fallback_batches"] + 1
    assert api.process_stats()["fallback_batches"] == pbefore["fallback_batches"] + 1
    assert "time-out in frame 4 unit 7" in err and "wave 255 gave up at coder_core.hpp:65535" in err and "barrier-only coder" in err
    assert enc.launch_info()["pipeline_waves"] == 0                    # the re-run was the window coder's
    # the next call is an ordinary one: the pipeline again, nothing counted, nothing printed
    out.zero_()
    _run(enc, d_frames, n, quota, out, sizes, rcs, use_async)
    err = capfd.readouterr().err
    for k in range(n):
        assert out[k, : int(sizes[k])].cpu().numpy().tobytes() == want[k][1], k
    assert enc.stats() == after and enc.launch_info()["pipeline_waves"] == 8 and "time-out" not in err
    enc.close()


def test_forced_timeout_in_a_split_unit_of_a_lone_frame(oracle, capfd):
    """a launch of one frame cuts its dense level-1 units into sub-ranges: the failing unit is one of those (the last unit in priority
    order: level 1, HH, plane 0), reported through its SubRecord and splice_units_kernel"""
    import torch
    dev = torch.device("cuda", 0)
    w, h, st, sg = 2048, 2048, 3, 2
    quota = 2 * w * h
    frame = synth.gray_frame(w, h, 606, 1)
    want = oracle.compress([frame], st, 0, sg, quota)
    d_frames = torch.from_numpy(frame.view(np.int16)[None]).to(dev)
    out = torch.zeros((1, quota), dtype=torch.uint8, device=dev)
    sizes = torch.zeros(1, dtype=torch.int64, device=dev)
    rcs = torch.zeros(1, dtype=torch.int32, device=dev)
    probe = api.Encoder(w, h, 1, st, 0, sg, max_frames=1)
    last = probe.info()["units_per_frame"] - 1
    probe.close()
    enc = _encoder_with_hook(f"0:{last}", w, h, 1, st, 0, sg, max_frames=1)
    capfd.readouterr()
    enc.encode_torch(d_frames, quota, out, sizes, rcs)
    torch.cuda.synchronize()
    err = capfd.readouterr().err
    assert int(rcs[0]) == want[0] and out[0, : int(sizes[0])].cpu().numpy().tobytes() == want[1]
    assert enc.stats()["fallback_batches"] == 1 and f"time-out in frame 0 unit {last}" in err
    enc.encode_torch(d_frames, quota, out, sizes, rcs)
    torch.cuda.synchronize()
    info = enc.launch_info()
    assert info["split"] and info["sub_range_workgroups"] > 0 and enc.stats()["fallback_batches"] == 1
    assert out[0, : int(sizes[0])].cpu().numpy().tobytes() == want[1]
    enc.close()


def test_forced_timeout_inside_the_host_batch_pipeline(oracle, tmp_path):
    """icerx_compress_batch_uint16_devices: a sub-batch of the host-fed pipeline hits the time-out; the call still returns every frame's
    stream (a process of its own: the pooled encoders of the batch entry point read the hook when they are created)"""
    n, w, h, st, sg = 8, 256, 192, 3, 5
    quota = 2 * w * h
    frames = synth.gray_batch(n, w, h, 707, 1)
    np.save(tmp_path / "frames.npy", frames)
    code = f"""
import json, sys
import numpy as np
sys.path.insert(0, {ROOT!r})
from icer_compression_amd import api
frames = np.load({str(tmp_path / 'frames.npy')!r})
n = frames.shape[0]
out = np.zeros((n, {quota}), np.uint8); sizes = np.zeros(n, np.uint64); rcs = np.zeros(n, np.int32)
rc = api.compress_batch(frames, {st}, 0, {sg}, {quota}, out, sizes, rcs, devices=[0])
import zlib
print(json.dumps({{"rc": rc, "rcs": rcs.tolist(), "crc": [zlib.crc32(out[k, :int(sizes[k])].tobytes()) for k in range(n)],
                  "sizes": sizes.tolist(), "stats": api.process_stats()}}))
"""
    env = dict(os.environ, ICER_HIP_TEST_FAIL_UNIT="1:3")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])
    import zlib
    assert got["rc"] == 0
    for k in range(n):
        rc, stream, _ = oracle.compress([frames[k]], st, 0, sg, quota)
        assert got["rcs"][k] == rc and got["sizes"][k] == len(stream) and got["crc"][k] == zlib.crc32(stream), k
    assert got["stats"]["fallback_batches"] >= 1 and got["stats"]["unit_timeouts"] >= 1
    assert "time-out in frame 1 unit 3" in r.stderr and "barrier-only coder" in r.stderr
