"""The workgroup-window coder (csrc/coder_wg.hpp: one workgroup per coding unit, windows of one chunk per wave, barriers
only) in its CPU lane-loop build (tests/emu/wg_emu.cpp, tests/emu/coder_emu.cpp) against the oracle.  The waves of a
region run in order, backwards and shuffled: between two barriers their order must not matter.  The real parity tests
are tests/test_gpu_parity.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from icer_compression_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", params=[16, 4, 2, 1], ids=["16_waves", "4_waves", "2_waves", "1_wave"])
def wg(request):
    """the instances the product builds: icer::wg (16 wavefronts per workgroup), icer::wg4 (4: the list kernel of a lone frame),
    icer::wgs (2) and icer::wg1 (a single wavefront: no barrier skew at all) -- coder_wg_small.hpp"""
    src = os.path.join(ROOT, "tests", "emu", "wg_emu.cpp")
    so = os.path.join(ROOT, "tests", "emu", f"libwg_emu_{request.param}.so")
    csrc = os.path.join(ROOT, "icer_compression_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc))
    if not os.path.exists(so) or os.path.getmtime(so) < max(newest, os.path.getmtime(src)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", f"-DICER_WG_WAVES={request.param}", "-o", so, src])
    L = C.CDLL(so)
    assert L.emu_wg_waves() == request.param
    u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
    L.emu_wg_code_unit.restype = C.c_long
    L.emu_wg_code_unit.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, u8p, C.c_size_t, C.c_int]
    L.emu_wg_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]

    class Wg:
        lib = L

        @staticmethod
        def code_unit(plane, x, y, w, h, sb, lsb, cap=None, order=0, stop=0):
            cap = cap if cap is not None else ((w * h * 3 + 64) + 3) // 4 * 4
            out = np.zeros(cap + 8, np.uint8)
            L.emu_wg_set_order(order)
            bits = L.emu_wg_code_unit(plane.ctypes.data + 2 * (y * plane.shape[1] + x), w, h, plane.shape[1], sb, lsb, out, cap, stop)
            assert L.emu_wg_assert_line() == 0
            return bits, bytes(out[: (max(bits, 0) + 7) // 8])

        @staticmethod
        def stats(reset=True):
            buf = (C.c_ulonglong * 8)()
            L.emu_wg_stats(buf, 1 if reset else 0)
            return dict(windows=buf[0], detailed=buf[1], exact_chunks=buf[2], forced_flushes=buf[3], blank_runs=buf[4], steady_steps=buf[5])
    return Wg


def _sparse_plane(rng, w, h, amp, dens):
    mag = (rng.integers(0, amp + 1, (h, w)) * (rng.random((h, w)) < dens)).astype(np.uint16)
    return (mag | ((rng.integers(0, 2, (h, w)).astype(np.uint16) << 15) * (mag > 0))).astype(np.uint16)


def test_small_divisions_are_exact(wg):
    """floor_div_small / ceil_div_small (a float reciprocal estimate, corrected): the operand ranges blank_run uses"""
    L = wg.lib
    L.emu_wg_floor_div.restype = L.emu_wg_ceil_div.restype = C.c_uint
    L.emu_wg_floor_div.argtypes = L.emu_wg_ceil_div.argtypes = [C.c_uint, C.c_uint]
    rng = np.random.default_rng(99)
    dens = [1, 2, 5, 6, 7, 11, 17, 31, 44, 70, 144, 200, 206, 250, 512, 979, 4096, 65535]
    nums = [0, 1, 2, 199, 200, 201, 249, 250, 251, 511, 512, 513, 65535, 65536, 131071, (1 << 24) - 1] + [int(x) for x in rng.integers(0, 1 << 24, 300)]
    for d in dens + [int(x) for x in rng.integers(1, 1 << 16, 60)]:
        for n in nums:
            assert L.emu_wg_floor_div(n, d) == n // d, (n, d)
            if n and d >= 144:                                          # (ceil_div_small: 0 < num < 2^25, den >= 144)
                assert L.emu_wg_ceil_div(n, d) == -(-n // d), (n, d)


def test_units_small_and_degenerate(wg, oracle):
    img = synth.gray_frame(256, 192, 3, 0)
    coef = oracle.compress([img], 3, 0, 1, 1 << 22)[2][0]
    for sb in range(4):
        for lsb in range(9):
            for (x, y, w, h) in [(128 * (sb & 1), 96 * (sb >> 1), 128, 96), (5, 7, 33, 21), (0, 0, 1, 1), (3, 0, 70, 1), (3, 0, 1, 70), (9, 9, 64, 2),
                                 (0, 0, 64, 16), (0, 0, 64, 17), (1, 1, 63, 65)]:
                assert wg.code_unit(coef, x, y, w, h, sb, lsb, order=sb + lsb) == oracle.code_unit(coef, x, y, w, h, sb, lsb), (sb, lsb, x, y, w, h)


@pytest.mark.parametrize("order", [0, 1, 99])
def test_units_random_planes_with_ring_pressure(wg, oracle, order):
    rng = np.random.default_rng(5)
    wg.stats()
    for trial in range(24):
        w, h = int(rng.integers(1, 400)), int(rng.integers(1, 300))
        amp = int(rng.choice([1, 2, 4, 16, 64, 300, 2000, 30000]))
        dens = float(rng.choice([0.001, 0.01, 0.1, 0.5, 1.0]))
        plane = _sparse_plane(rng, w, h, amp, dens)
        for sb in (0, 1, 3):
            for lsb in (0, 1, 3, 8):
                assert wg.code_unit(plane, 0, 0, w, h, sb, lsb, order=order) == oracle.code_unit(plane, 0, 0, w, h, sb, lsb), (trial, sb, lsb)
    st = wg.stats()
    # the forced-flush machinery (E5) is really exercised by these planes
    assert st["exact_chunks"] > 20 and st["forced_flushes"] > 20, st


def test_forced_flush_heavy_units(wg, oracle):
    """long units whose rare bins keep words open across thousands of other words (icer_flush_encode, icer_encoding.c:141-189)"""
    rng = np.random.default_rng(77)
    wg.stats()
    for trial in range(10):
        w, h = int(rng.integers(300, 700)), int(rng.integers(200, 500))
        # mostly dense noise with rare large values: many bins in use, some of them rarely
        plane = _sparse_plane(rng, w, h, int(rng.choice([3, 7, 40, 500])), float(rng.choice([0.3, 0.7, 1.0])))
        plane[rng.random((h, w)) < 0.002] = 0x4000
        for sb, lsb in ((0, 0), (3, 1), (1, 2)):
            assert wg.code_unit(plane, 0, 0, w, h, sb, lsb, order=trial) == oracle.code_unit(plane, 0, 0, w, h, sb, lsb), (trial, sb, lsb)
    st = wg.stats()
    assert st["forced_flushes"] > 200, st


def test_blank_runs_in_closed_form(wg, oracle):
    """runs of blank chunks (chunk table, blank_run): long empty stretches, stretches broken by single significant pixels,
    units that end in a blank run, stale open words while a blank stretch fills the ring; with and without the table"""
    rng = np.random.default_rng(11)
    wg.stats()
    for trial in range(14):
        w, h = int(rng.integers(64, 500)), int(rng.integers(40, 400))
        plane = np.zeros((h, w), np.uint16)
        kind = trial % 4
        if kind == 0:                                   # a busy top, then nothing
            plane[: h // 5] = _sparse_plane(rng, w, h // 5, 200, 0.5)
        elif kind == 1:                                 # isolated pixels
            plane[rng.random((h, w)) < 0.0005] = int(rng.integers(1, 30000))
        elif kind == 2:                                 # busy columns: every row leaves and re-enters blank chunks
            plane[:, : w // 6] = _sparse_plane(rng, w // 6, h, 60, 0.8)
        else:                                           # a few busy rows far apart
            for r in rng.integers(0, h, 3):
                plane[r] = _sparse_plane(rng, w, 1, 1000, 0.7)[0]
        for sb, lsb in ((0, 0), (1, 2), (3, 5), (2, 8)):
            want = oracle.code_unit(plane, 0, 0, w, h, sb, lsb)
            assert wg.code_unit(plane, 0, 0, w, h, sb, lsb, order=trial) == want, (trial, sb, lsb)
            wg.lib.emu_wg_use_table(0)
            try:
                assert wg.code_unit(plane, 0, 0, w, h, sb, lsb, order=trial) == want, (trial, sb, lsb, "no table")
            finally:
                wg.lib.emu_wg_use_table(1)
    st = wg.stats()
    assert st["blank_runs"] > 300 and st["forced_flushes"] > 0, st


def test_blank_planes_in_steady_state(wg, oracle):
    """all-blank stretches long enough for the counts of context 0 to reach their cycle (blank_run's steady-state step:
    all remaining whole rescale cycles at once): empty planes of many sizes, planes that are busy first / in between /
    at the end (the step starts from every state of the Golomb bins and open words), all subbands"""
    rng = np.random.default_rng(2718)
    wg.stats()
    for trial in range(16):
        w, h = int(rng.integers(100, 640)), int(rng.integers(100, 500))
        plane = np.zeros((h, w), np.uint16)
        kind = trial % 4
        if kind == 1:                                   # busy first rows
            k = int(rng.integers(1, 12))
            plane[:k] = _sparse_plane(rng, w, k, 300, float(rng.choice([0.02, 0.3, 0.9])))
        elif kind == 2:                                 # busy band in the middle
            r0 = h // 2
            plane[r0:r0 + 3] = _sparse_plane(rng, w, 3, 90, 0.5)
        elif kind == 3:                                 # busy last rows and a lone pixel
            plane[-2:] = _sparse_plane(rng, w, 2, 2000, 0.4)
            plane[h // 3, w // 2] = 7
        for sb, lsb in ((0, 0), (1, 3), (3, 8), (2, 6)):
            want = oracle.code_unit(plane, 0, 0, w, h, sb, lsb)
            assert wg.code_unit(plane, 0, 0, w, h, sb, lsb, order=trial) == want, (trial, w, h, sb, lsb)
    st = wg.stats()
    assert st["steady_steps"] > 100 and st["blank_runs"] > 50, st


def test_slot_capacity_rule_and_stop(wg, oracle):
    rng = np.random.default_rng(6)
    plane = rng.integers(0, 512, (80, 150)).astype(np.uint16)
    bits, payload = oracle.code_unit(plane, 0, 0, 150, 80, 0, 0)
    nbytes = (bits + 7) // 8
    for cap in (4, 64, 1024, (bits // 8) // 4 * 4, (bits // 8) // 4 * 4 + 4, nbytes + 8):
        got = wg.code_unit(plane, 0, 0, 150, 80, 0, 0, cap=cap)
        if bits // 8 < cap:
            assert got == (bits, payload)
        else:
            assert got[0] == -5
    # progressive mode: the stop flag is seen at a window check; every wave leaves
    big = rng.integers(0, 512, (300, 400)).astype(np.uint16)
    assert wg.code_unit(big, 0, 0, 400, 300, 0, 0, stop=1)[0] == -3


CASES = [(64, 64, 2, 0, 4, 1 << 20), (200, 160, 3, 1, 1, 1 << 20), (100, 75, 3, 2, 7, 1 << 20), (128, 128, 4, 3, 16, 5000),
         (257, 131, 3, 5, 32, 1 << 20), (80, 80, 2, 6, 5, 900), (512, 512, 3, 0, 10, 1 << 20), (64, 64, 2, 0, 4, 28), (24, 24, 3, 0, 9, 1 << 16)]


@pytest.mark.parametrize("case", CASES)
def test_pipeline_gray(emu, oracle, case):
    w, h, st, f, sg, q = case
    try:
        for mode in (0, 1):
            emu.lib.emu_set_coder(1, mode * 31)
            img = synth.gray_frame(w, h, 7, mode)
            a, b = oracle.compress([img], st, f, sg, q), emu.compress([img], st, f, sg, q)
            assert emu.lib.emu_wg_assert_line() == 0
            assert a[0] == b[0] and a[1] == b[1] and (a[0] not in (0, -5) or np.array_equal(a[2][0], b[2][0])), (b[0], b[3])
    finally:
        emu.lib.emu_set_coder(0, 0)


def test_pipeline_yuv_and_u8(emu, oracle):
    try:
        emu.lib.emu_set_coder(1, 5)
        for (w, h, st, f, sg, q) in [(64, 64, 2, 0, 4, 1 << 20), (128, 96, 3, 0, 10, 6000)]:
            planes = synth.color_frame_yuv(w, h, 3)
            a, b = oracle.compress(planes, st, f, sg, q), emu.compress(planes, st, f, sg, q)
            assert a[0] == b[0] and a[1] == b[1] and all(np.array_equal(p, r) for p, r in zip(a[2], b[2]))
        rng = np.random.default_rng(3)
        planes = [np.clip(10 + rng.integers(0, 60, (70, 90)), 0, 255).astype(np.uint8)]
        a, b = oracle.compress_u8(planes, 3, 0, 5, 1 << 16), emu.compress_u8(planes, 3, 0, 5, 1 << 16)
        assert a[0] == b[0] and a[1] == b[1]
        assert emu.lib.emu_wg_assert_line() == 0
    finally:
        emu.lib.emu_set_coder(0, 0)
