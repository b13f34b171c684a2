"""GPU parity tests of the decoder (libicer_hip_dec.so through its C ABI) against the decoder oracle: random gray / YUV,
uint16 / uint8 streams incl. quota-cut ones, wrong decode parameters, damaged / truncated / re-ordered streams, the golden
decoder digests up to 4096 x 4096 and the batch decoder object, for all three decode kernels (one thread per chain, one
wavefront per chain with a lane per bit plane, one wavefront per bit plane -- the default since round 4).  See HISTORY.md 6b (summary: DESIGN.md 8).
"""
import hashlib
import json
import os

import numpy as np
import pytest

from icer_compression_amd import synth
from oracle.binding import Oracle
from tests.test_oracle_decoder import random_case

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "golden.json")))


@pytest.fixture(scope="module")
def orc():
    return Oracle()


@pytest.fixture(scope="module")
def dec():
    from icer_compression_amd import decoder
    decoder.load_library()
    return decoder


def same(a, b):
    return a[0] == b[0] and a[1:3] == b[1:3] and all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))


@pytest.fixture(params=["0", "1", "2"], ids=["thread-per-chain", "wave-per-chain", "wave-per-plane"])
def kernel(request):
    """ICER_DEC_WAVE: which decode kernel decoder.hip launches (read per call)"""
    old = os.environ.get("ICER_DEC_WAVE")
    os.environ["ICER_DEC_WAVE"] = request.param
    yield request.param
    if old is None:
        os.environ.pop("ICER_DEC_WAVE", None)
    else:
        os.environ["ICER_DEC_WAVE"] = old


@pytest.mark.timeout(600)
def test_random_streams(dec, orc, kernel):
    rng = np.random.default_rng(99)
    done = 0
    for _ in range(120):
        planes, st, filt, sg, ch, bits, quota = random_case(rng)
        rc, stream, _ = (orc.compress if bits == 16 else orc.compress_u8)(planes, st, filt, sg, quota)
        if not stream:
            continue
        dsg = sg if rng.random() < 0.9 else int(rng.integers(1, 33))
        h, w = planes[0].shape
        a = dec.decompress(stream, ch, st, filt, dsg, bufsize=w * h, bits=bits)
        b = orc.decompress(stream, ch, st, filt, dsg, bufsize=w * h, bits=bits)
        assert same(a, b), (planes[0].shape, st, filt, sg, dsg, ch, bits, quota, a[0], b[0])
        done += 1
    assert done > 80


@pytest.mark.timeout(300)
def test_very_narrow_segments(dec, orc, kernel):
    """segments a few samples wide put the planes of a chain many rows apart: sizes the wave kernel's row ring
    (ring_rows_for in decoder_wave.hpp); same cases as tests/test_emu_decoder.py::test_wave_kernel_on_very_narrow_segments"""
    rng = np.random.default_rng(5150)
    for w, h, st, sg in ((6, 180, 1, 6), (9, 150, 2, 8), (7, 96, 1, 3), (12, 200, 2, 12), (17, 120, 3, 9), (24, 160, 3, 32),
                         (6, 6, 1, 1), (200, 6, 1, 2), (40, 130, 4, 20)):
        img = rng.integers(0, 256, (h, w)).astype(np.uint16)
        rc, stream, _ = orc.compress([img], st, 0, sg, 4 * w * h + 40000)
        assert stream
        assert same(dec.decompress(stream, 1, st, 0, sg, bufsize=w * h), orc.decompress(stream, 1, st, 0, sg, bufsize=w * h)), (w, h, st, sg)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("name", ["kat_512_m1", "kat_512_quota30000", "kat_odd_517x389_filtC", "kat_color_512_quota",
                                  "u8_512_gray", "u8_517x389_filtB_quota", "u8_512_yuv_4st", "C2_4096_gray_5st_10seg"])
def test_golden_decodes(dec, orc, name, kernel):
    """digests of the reference DECODER's output (tests/golden/make_golden.py)"""
    g = GOLDEN[name]
    kind, w, h = g["kind"], g["w"], g["h"]
    if kind == "gray":
        planes = [synth.gray_frame(w, h, g["seed"], g["mode"])]
    elif kind == "yuv":
        planes = list(synth.color_frame_yuv(w, h, g["seed"]))
    elif kind == "gray8":
        planes = [synth.gray_frame_u8(w, h, g["seed"], g["mode"])]
    else:
        planes = list(synth.color_frame_yuv_u8(w, h, g["seed"]))
    u8 = kind.endswith("8")
    rc, stream, _ = (orc.compress_u8 if u8 else orc.compress)(planes, g["stages"], g["filt"], g["segments"], g["quota"])
    assert hashlib.sha256(stream).hexdigest()[:16] == g["sha256_16"]
    drc, dw, dh, out = dec.decompress(stream, len(planes), g["stages"], g["filt"], g["segments"], bits=8 if u8 else 16)
    hsh = hashlib.sha256()
    for p in out:
        hsh.update(p.tobytes())
    assert (drc, dw, dh, hsh.hexdigest()[:16]) == (g["decoded_rc"], g["decoded_w"], g["decoded_h"], g["decoded_sha256_16"])


@pytest.mark.timeout(300)
def test_error_paths(dec, orc, kernel):
    img = synth.gray_frame(160, 120, 3, 1)
    rc, stream, _ = orc.compress([img], 3, 1, 5, 2 * 160 * 120)
    assert dec.decompress(stream, 1, 3, 1, 5, bufsize=160 * 120 - 1)[0] == -5
    for s in (b"", stream[: len(stream) // 2], stream[5:], b"\x00" * 9 + stream + b"\x5b\x60\x00"):
        assert same(dec.decompress(s, 1, 3, 1, 5, bufsize=160 * 120), orc.decompress(s, 1, 3, 1, 5, bufsize=160 * 120)), len(s)


@pytest.mark.timeout(600)
def test_batch_decoder_object(dec, orc, kernel):
    """Part 2: mixed sizes, an empty stream and an oversized frame in one batch; host buffers and device buffers (torch)"""
    import torch
    rng = np.random.default_rng(5)
    for bits in (16, 8):
        top = 200 if bits == 16 else 50
        comp = orc.compress if bits == 16 else orc.compress_u8
        dt = np.uint16 if bits == 16 else np.uint8
        imgs = [rng.integers(0, top, shp).astype(dt) for shp in ((64, 96), (64, 96), (40, 56), (96, 128), (64, 96))]
        streams = [comp([im], 2, 0, 6, 4 * im.size)[1] for im in imgs]
        streams[1] = b""
        stride = 64 * 96 + 11
        want = [orc.decompress(s, 1, 2, 0, 6, bufsize=stride, bits=bits) for s in streams]
        d = dec.Decoder(1, 2, 0, 6, bits=bits)
        rc, res = d.decode_host(streams, stride)
        assert rc == 0
        for (rk, wk, hk, pk), wnt in zip(res, want):
            assert (rk, wk, hk) == wnt[:3]
            assert rk == -5 or np.array_equal(pk[0][: wk * hk], wnt[3][0][: wk * hk])
        blob, offs, lens = d._pack(streams)
        d_blob = torch.from_numpy(blob).cuda()
        d_out = torch.zeros(len(streams) * stride, dtype=torch.int16 if bits == 16 else torch.uint8, device="cuda")
        rc, rcs, ws, hs = d.decode_device(len(streams), d_blob.data_ptr(), offs, lens, d_out.data_ptr(), stride)
        assert rc == 0 and rcs == [w_[0] for w_ in want]
        out = d_out.cpu().numpy().view(dt)
        for k, wnt in enumerate(want):
            if rcs[k] == 0:
                assert np.array_equal(out[k * stride: k * stride + ws[k] * hs[k]], wnt[3][0][: ws[k] * hs[k]])
        d.close()
