"""The decoder's DEVICE code (icer_compression_amd/csrc/decoder_core.hpp + decoder_plan.hpp), compiled by g++ and driven
like decoder.hip drives it on the GPU (tests/emu/decoder_emu.cpp), against the decoder oracle -- which
tests/test_oracle_decoder.py pins to the reference decoder.  CPU only: this is how the device decoder is checked in a
container without a GPU.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle.binding import Oracle
from tests.test_oracle_decoder import packets, random_case

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emu", "decoder_emu.cpp")
LIB = os.path.join(HERE, "emu", "libdecoder_emu.so")
CSRC = os.path.join(HERE, "..", "icer_compression_amd", "csrc")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


@pytest.fixture(scope="module")
def emu():
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("decoder_core.hpp", "decoder_plan.hpp", "decoder_wave.hpp", "decoder_planes.hpp", "wave.hpp", "plan.hpp",
                                                    "icer_tables.hpp")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-DICER_WAVE_EMU", "-o", LIB, SRC])
    lib = C.CDLL(LIB)
    sz = C.c_size_t
    lib.emu_decompress.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(sz), C.POINTER(sz), sz, u8p, sz, C.c_int, C.c_int,
                                   C.c_uint, C.c_int]

    def decompress(stream, channels, stages, filt, segments, bufsize=None, bits=16, w0=0, h0=0):
        if bufsize is None:
            bufsize = int.from_bytes(stream[8:12], "little") * int.from_bytes(stream[12:16], "little") if len(stream) >= 28 else 0
        buf = np.frombuffer(stream, dtype=np.uint8).copy() if len(stream) else np.zeros(1, np.uint8)
        planes = [np.zeros(max(bufsize, 1), np.uint16) for _ in range(channels)]
        ptrs = (C.c_void_p * channels)(*[p.ctypes.data for p in planes])
        w, h = sz(w0), sz(h0)
        rc = lib.emu_decompress(ptrs, channels, C.byref(w), C.byref(h), bufsize, buf, len(stream), stages, filt, segments, bits)
        if bits == 8:
            planes = [p.astype(np.uint8) for p in planes]
        return rc, w.value, h.value, planes
    decompress.lib = lib
    return decompress


@pytest.fixture(scope="module")
def orc():
    return Oracle()


def same(a, b):
    return a[0] == b[0] and a[1:3] == b[1:3] and all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))


@pytest.fixture(params=[0, 3, 4], ids=["thread_per_chain", "wave_per_chain", "wave_per_plane"])
def kernel(request, emu):
    """0: decode_chain (one thread per chain, plane after plane); 3: decode_chain_wave (planes side by side, one decision per
    round: plane_decision / entropy_decode_fast); 4: pw_step of decoder_planes.hpp (one wavefront per bit plane, wave-uniform
    decisions, the planes' waves stepped in turns), chains with a very short packet through kernel 3"""
    emu.lib.emu_decoder_mode(request.param)
    yield request.param
    emu.lib.emu_decoder_mode(0)


def test_fast_code_word_table_covers_the_code(emu):
    """the 5-bit look-up of entropy_decode_fast stands for the reference's search on all 1024 ten-bit inputs of every bin"""
    assert emu.lib.emu_decoder_lut_ok() == 1


def test_device_decoder_matches_the_oracle_on_random_streams(emu, orc, kernel):
    rng = np.random.default_rng(424242 + kernel)
    done, rcs = 0, set()
    for _ in range(220):
        planes, st, filt, sg, ch, bits, quota = random_case(rng)
        rc, stream, _ = (orc.compress if bits == 16 else orc.compress_u8)(planes, st, filt, sg, quota)
        if not stream:
            continue
        dsg = sg if rng.random() < 0.9 else int(rng.integers(1, 33))
        dst = st if rng.random() < 0.9 else int(rng.integers(1, 7))
        a = emu(stream, ch, dst, filt, dsg, bits=bits)
        b = orc.decompress(stream, ch, dst, filt, dsg, bits=bits)
        if b[0] == -3 and bits == 8:
            # (a grid error leaves sign-magnitude words behind; the oracle's uint8 wrapper narrows them the same way)
            pass
        assert same(a, b), (planes[0].shape, st, dst, filt, sg, dsg, ch, bits, quota, a[0], b[0])
        rcs.add(a[0])
        done += 1
    assert done > 150 and {0, -3} <= rcs


def test_device_decoder_on_damaged_streams(emu, orc, kernel):
    from icer_compression_amd import synth
    rng = np.random.default_rng(8)
    img = synth.gray_frame(160, 120, 3, 1)
    rc, stream, _ = orc.compress([img], 3, 1, 5, 2 * 160 * 120)
    pk = packets(stream)
    variants = [b"", b"\x5b\x60" * 40, stream[: len(stream) // 2], stream[: len(stream) - 1], stream[5:],
                b"".join(reversed(pk)), b"".join(pk + pk[:7]), b"\x00" * 9 + stream + b"\x5b\x60\x00"]
    for _ in range(10):
        s = bytearray(stream)
        for _ in range(int(rng.integers(1, 6))):
            s[int(rng.integers(0, len(s)))] ^= 1 << int(rng.integers(0, 8))
        variants.append(bytes(s))
    for s in variants:
        assert same(emu(s, 1, 3, 1, 5, bufsize=160 * 120), orc.decompress(s, 1, 3, 1, 5, bufsize=160 * 120)), len(s)
    assert emu(stream, 1, 3, 1, 5, bufsize=160 * 120 - 1)[0] == -5
    assert emu(stream, 2, 3, 1, 5)[0] == -11 and emu(stream, 1, 7, 1, 5)[0] == -4


def test_device_decoder_headline_frame(emu, orc, kernel):
    """1024 x 1024 (the largest frame the CPU build of the device code decodes in a few seconds): lossless round trip"""
    from icer_compression_amd import synth
    img = synth.gray_frame(1024, 1024, 12345, 1)
    rc, stream, _ = orc.compress([img], 4, 0, 16, 2 * 1024 * 1024)
    rc2, w, h, planes = emu(stream, 1, 4, 0, 16)
    assert (rc, rc2, w, h) == (0, 0, 1024, 1024) and np.array_equal(planes[0].reshape(1024, 1024), img)


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_planes_side_by_side_schedule(emu, orc, mode):
    """every packet as a job of its own: a plane may decode sample (r, c) once the plane above has finished
    (r + 1, c + 1); a failing plane takes back what the planes below it wrote.  Same images as the serial order --
    including quota-cut streams, one-code-word packets (refused by the reference decoder) and data above the coded planes,
    where the decoder derails and planes fail at arbitrary points."""
    # 1: planes top-down inside a lock-step iteration, 2: bottom-up, 3: the wave kernel (decoder_wave.hpp, LDS row ring)
    emu.lib.emu_decoder_mode(mode)
    try:
        rng = np.random.default_rng(77 + mode)
        done = 0
        for _ in range(200):
            planes, st, filt, sg, ch, bits, quota = random_case(rng)
            rc, stream, _ = (orc.compress if bits == 16 else orc.compress_u8)(planes, st, filt, sg, quota)
            if not stream:
                continue
            assert same(emu(stream, ch, st, filt, sg, bits=bits), orc.decompress(stream, ch, st, filt, sg, bits=bits)), \
                (planes[0].shape, st, filt, sg, ch, bits, quota)
            done += 1
        stats = (C.c_ulonglong * 4)()
        emu.lib.emu_decoder_stats(stats)
        assert done > 120
        assert stats[1] / stats[0] > 3.0          # samples per lock-step iteration: the planes really overlap
        # planes that fail half-way, with planes below them already at work: a middle plane's packet is cut down to 4 or 5
        # bits (CRCs fixed up); the decoder then refuses the first code word of that length (icer_decoding.c:159), which
        # comes up some rows into the plane
        import zlib
        from icer_compression_amd import synth
        img = synth.gray_frame(192, 160, 5, 1)
        rc, stream, _ = orc.compress([img], 3, 0, 4, 2 * 192 * 160)
        emu.lib.emu_decoder_mode(mode)
        hit = 0
        for trial in range(10):
            out = []
            for p in packets(stream):
                p = bytearray(p)
                if (p[7] & 15) in (4, 5, 6) and len(p) > 28 + 8 and rng.random() < 0.3:
                    p = p[:29]
                    p[16:20] = int(rng.integers(4, 6)).to_bytes(4, "little")
                    p[20:24] = zlib.crc32(bytes(p[28:])).to_bytes(4, "little")
                    p[24:28] = zlib.crc32(bytes(p[:24])).to_bytes(4, "little")
                out.append(bytes(p))
            s2 = b"".join(out)
            a, b = emu(s2, 1, 3, 0, 4), orc.decompress(s2, 1, 3, 0, 4)
            assert same(a, b), trial
            hit += not np.array_equal(a[3][0].reshape(160, 192), img)
        emu.lib.emu_decoder_stats(stats)
        assert hit > 0 and stats[2] > 0           # roll-backs happened
        assert stats[3] == 0                      # (wave kernel) every chain wrote all its rows back
    finally:
        emu.lib.emu_decoder_mode(0)


def test_wave_kernel_on_very_narrow_segments(emu, orc):
    """segments one to a few samples wide: the planes of a chain then lie many rows apart (each needs w + 2 samples of
    lead), which is what sizes the row ring (ring_rows_for); too few rows and the side-by-side schedule would lock up"""
    emu.lib.emu_decoder_mode(3)
    try:
        rng = np.random.default_rng(5150)
        stats = (C.c_ulonglong * 4)()
        for w, h, st, sg in ((6, 180, 1, 6), (9, 150, 2, 8), (7, 96, 1, 3), (12, 200, 2, 12), (17, 120, 3, 9), (24, 160, 3, 32),
                             (6, 6, 1, 1), (200, 6, 1, 2), (40, 130, 4, 20)):
            img = rng.integers(0, 256, (h, w)).astype(np.uint16)
            rc, stream, _ = orc.compress([img], st, 0, sg, 4 * w * h + 40000)
            assert stream
            assert same(emu(stream, 1, st, 0, sg), orc.decompress(stream, 1, st, 0, sg)), (w, h, st, sg)
        emu.lib.emu_decoder_stats(stats)
        assert stats[3] == 0
    finally:
        emu.lib.emu_decoder_mode(0)


def test_wave_per_plane_kernel_schedules_and_shapes(emu, orc):
    """decoder_planes.hpp: the waves of a chain meet only through progress counters, so the image must not depend on the order
    in which they get to run (here: drawn at random, several seeds), and the ring of planes + 4 rows must never lock up --
    segments from one sample wide (a block is then a whole row) to several blocks per row, rows that end inside a block,
    8-bit streams (7 planes), quota-cut streams whose lower planes are missing, every subband's context tables"""
    stats = (C.c_ulonglong * 4)()
    rng = np.random.default_rng(31337)
    cases = []
    for w, h, st, sg in ((6, 180, 1, 6), (9, 150, 2, 8), (7, 96, 1, 3), (17, 120, 3, 9), (24, 160, 3, 32), (6, 6, 1, 1), (200, 6, 1, 2),
                         (40, 130, 4, 20), (130, 70, 1, 1), (260, 48, 1, 2), (191, 33, 2, 3), (320, 200, 3, 4)):
        img = rng.integers(0, 256, (h, w)).astype(np.uint16)
        for quota in (4 * w * h + 40000, max(600, w * h // 3)):
            rc, stream, _ = orc.compress([img], st, int(rng.integers(0, 7)), sg, quota)
            if stream:
                cases.append((stream, 1, st, sg, 16))
    for _ in range(12):
        planes, st, filt, sg, ch, bits, quota = random_case(rng)
        rc, stream, _ = (orc.compress if bits == 16 else orc.compress_u8)(planes, st, filt, sg, quota)
        if stream:
            cases.append((stream, ch, st, sg, bits))
    try:
        for seed in (0, 1, 2, 77):
            emu.lib.emu_decoder_mode(4)
            emu.lib.emu_decoder_order_seed(seed)
            for stream, ch, st, sg, bits in cases:
                # (the filter only matters to the inverse transform: decode with filter A everywhere but for the random cases)
                want = orc.decompress(stream, ch, st, 0, sg, bits=bits)
                assert same(emu(stream, ch, st, 0, sg, bits=bits), want), (seed, len(stream), ch, st, sg, bits)
            emu.lib.emu_decoder_stats(stats)
            assert stats[0] > 0 and stats[3] == 0, (seed, list(stats))        # the new kernel ran, and never locked up
        runs = (C.c_ulonglong * 2)()
        emu.lib.emu_decoder_run_stats(runs)
        assert runs[0] > 100 and runs[1] > 5 * runs[0], list(runs)            # ... and took its zero-run shortcut (pw_zero_run)
    finally:
        emu.lib.emu_decoder_order_seed(0)
        emu.lib.emu_decoder_mode(0)


def test_host_pipeline_on_a_mock_hip_runtime(orc, tmp_path):
    """decoder.hip itself -- its C ABI, allocations, copies, launch geometries and clean-up -- compiled by g++ against
    tests/emu/hip_mock.h (device memory = poisoned host memory, a launch = a loop over the grid) and called through
    icer_compression_amd/decoder.py like the GPU library will be: both decode kernels, error paths, odd streams.
    (The same build under -fsanitize=address,undefined is how the host code was checked for wrong sizes.)"""
    from icer_compression_amd import decoder, synth
    root = os.path.dirname(HERE)
    lib_path = str(tmp_path / "libdecoder_mock.so")
    subprocess.check_call(["g++", "-x", "c++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas",
                           "-DICER_HOST_MOCK", "-DICER_WAVE_EMU", "-include", os.path.join(HERE, "emu", "hip_mock.h"),
                           "-o", lib_path, os.path.join(root, "icer_compression_amd", "csrc", "decoder.hip")])
    lib = decoder.bind(lib_path)
    old = os.environ.get("ICER_DEC_WAVE")
    try:
        rng = np.random.default_rng(2024)
        done = 0
        for _ in range(45):
            planes, st, filt, sg, ch, bits, quota = random_case(rng)
            rc, stream, _ = (orc.compress if bits == 16 else orc.compress_u8)(planes, st, filt, sg, quota)
            if not stream:
                continue
            dsg = sg if rng.random() < 0.9 else int(rng.integers(1, 33))
            h, w = planes[0].shape
            want = orc.decompress(stream, ch, st, filt, dsg, bufsize=w * h, bits=bits)
            for mode in ("0", "1", "2"):                         # thread per chain, lane per plane, wavefront per plane
                os.environ["ICER_DEC_WAVE"] = mode
                assert same(decoder.decompress(stream, ch, st, filt, dsg, bufsize=w * h, bits=bits, lib=lib), want), \
                    (mode, planes[0].shape, st, filt, sg, dsg, ch, bits, quota)
            done += 1
        assert done > 30
        img = synth.gray_frame(160, 120, 3, 1)
        rc, stream, _ = orc.compress([img], 3, 1, 5, 2 * 160 * 120)
        for s in (b"", b"\x5b\x60" * 40, stream[: len(stream) // 2], stream[5:], b"\x00" * 9 + stream + b"\x5b\x60\x00",
                  b"".join(reversed(packets(stream)))):
            for mode in ("0", "1", "2"):
                os.environ["ICER_DEC_WAVE"] = mode
                assert same(decoder.decompress(s, 1, 3, 1, 5, bufsize=160 * 120, lib=lib),
                            orc.decompress(s, 1, 3, 1, 5, bufsize=160 * 120)), len(s)
        assert decoder.decompress(stream, 1, 3, 1, 5, bufsize=160 * 120 - 1, lib=lib)[0] == -5
        assert decoder.decompress(stream, 1, 3, 1, 5, lib=lib)[1:3] == (160, 120)
        # batches (Part 2): streams of different sizes, one that stops on a grid error, an empty one, one that does not
        # fit the frame stride; host and "device" (= host, in the mock) buffers, 16- and 8-bit
        for bits in (16, 8):
            top = 200 if bits == 16 else 50
            comp = orc.compress if bits == 16 else orc.compress_u8
            dt = np.uint16 if bits == 16 else np.uint8
            imgs = [rng.integers(0, top, shp).astype(dt) for shp in ((64, 96), (64, 96), (40, 56), (96, 128), (64, 96))]
            streams = [comp([im], 2, 0, 6, 4 * im.size)[1] for im in imgs]
            streams[1] = b""
            stride = 64 * 96 + 11
            want = [orc.decompress(s_, 1, 2, 0, 6, bufsize=stride, bits=bits) for s_ in streams]
            assert [w_[0] for w_ in want] == [0, -3, 0, -5, 0]        # (an empty stream: no size, so no segment grid)
            for mode in ("0", "1", "2"):
                os.environ["ICER_DEC_WAVE"] = mode
                dec = decoder.Decoder(1, 2, 0, 6, bits=bits, lib=lib)
                rc, res = dec.decode_host(streams, stride)
                assert rc == 0
                for (rk, wk, hk, pk), wnt in zip(res, want):
                    assert (rk, wk, hk) == wnt[:3]
                    assert np.array_equal(pk[0][: wk * hk], wnt[3][0][: wk * hk]) or rk == -5
                blob, offs, lens = dec._pack(streams)
                out = np.full(len(streams) * stride, 7, dt)
                rc, rcs, ws, hs = dec.decode_device(len(streams), blob.ctypes.data, offs, lens, out.ctypes.data, stride)
                assert rc == 0 and rcs == [w_[0] for w_ in want]
                for k, wnt in enumerate(want):
                    if rcs[k] == 0:
                        assert np.array_equal(out[k * stride: k * stride + ws[k] * hs[k]], wnt[3][0][: ws[k] * hs[k]]), (bits, mode, k)
                dec.close()
        # a batch in which a frame stops with ICER_TOO_MANY_SEGMENTS (decoded with more segments than its subbands hold)
        small = rng.integers(0, 100, (24, 24)).astype(np.uint16)
        s_small = orc.compress([small], 3, 0, 2, 4 * small.size)[1]
        s_ok = orc.compress([imgs16 := rng.integers(0, 100, (24, 24)).astype(np.uint16)], 3, 0, 2, 4 * 576)[1]
        dec = decoder.Decoder(1, 3, 0, 20, lib=lib)
        rc, res = dec.decode_host([s_small, s_ok], 576)
        wnt = [orc.decompress(s_, 1, 3, 0, 20, bufsize=576) for s_ in (s_small, s_ok)]
        assert rc == 0 and [r[0] for r in res] == [w_[0] for w_ in wnt] and -3 in [r[0] for r in res]
        for r, w_ in zip(res, wnt):
            assert np.array_equal(r[3][0][:576], w_[3][0][:576])
        dec.close()
        os.environ["ICER_MOCK_NO_DEVICE"] = "1"
        assert decoder.decompress(stream, 1, 3, 1, 5, lib=lib)[0] == -10
    finally:
        os.environ.pop("ICER_MOCK_NO_DEVICE", None)
        if old is None:
            os.environ.pop("ICER_DEC_WAVE", None)
        else:
            os.environ["ICER_DEC_WAVE"] = old
