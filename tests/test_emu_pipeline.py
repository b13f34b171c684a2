"""The kernel sources (csrc/*_core.hpp) and the host planner, built as a CPU lane-loop emulation (tests/emu),
against the oracle.  This is how the wave-parallel algorithm is debugged in a container without a GPU; the
real parity tests are tests/test_gpu_parity.py."""
import numpy as np
import pytest

from icer_compression_amd import synth


@pytest.fixture(params=[1, 2], ids=["8-waves", "11-waves"], autouse=True)
def pipeline_shape(request, emu, monkeypatch):
    """both shapes of the pipeline's workgroup (code_units_kernel<8>: one pixel wave, one golomb wave; <11>: two pixel waves,
    golomb state wave + two workers)"""
    emu.lib.emu_set_shape(request.param, 0 if request.param == 1 else 2)
    monkeypatch.setenv("ICER_EMU_SHAPE", str(request.param))
    yield request.param
    emu.lib.emu_set_shape(2, 2)


def test_dwt_core(emu, oracle):
    rng = np.random.default_rng(2)
    for filt in range(7):
        for (w, h, st) in [(64, 64, 3), (37, 53, 2), (96, 40, 3), (25, 25, 2), (130, 70, 4), (5, 5, 1), (6, 7, 1), (11, 13, 2),
                           (300, 200, 2), (517, 389, 3), (264, 136, 1), (700, 300, 2)]:   # (the last four: tiles on the interior fast path too)
            for hi in (256, 65536):
                img = rng.integers(0, hi, (h, w)).astype(np.uint16)
                a, b = oracle.dwt(img, st, filt), emu.dwt(img, st, filt)
                assert a[0] == b[0] and np.array_equal(a[1], b[1]), (filt, w, h, st, hi)
    # both tile paths ran: interior tiles through dwt_fast_*, and the very same frames through the generic phases alone
    assert emu.lib.emu_dwt_fast_tiles() > 50
    emu.lib.emu_dwt_fast(0)
    try:
        for filt in (0, 2, 5):
            img = rng.integers(0, 65536, (300, 700)).astype(np.uint16)
            a, b = oracle.dwt(img, 2, filt), emu.dwt(img, 2, filt)
            assert a[0] == b[0] and np.array_equal(a[1], b[1])
    finally:
        emu.lib.emu_dwt_fast(1)


def test_coding_units_small_and_degenerate(emu, oracle):
    img = synth.gray_frame(256, 192, 3, 0)
    _, coef = oracle.dwt(img, 3, 0)
    coef = oracle.compress([img], 3, 0, 1, 1 << 22)[2][0]
    for sb in range(4):
        for lsb in range(9):
            for (x, y, w, h) in [(128 * (sb & 1), 96 * (sb >> 1), 128, 96), (5, 7, 33, 21), (0, 0, 1, 1), (3, 0, 70, 1), (3, 0, 1, 70), (9, 9, 64, 2)]:
                assert emu.code_unit(coef, x, y, w, h, sb, lsb) == oracle.code_unit(coef, x, y, w, h, sb, lsb)


def test_coding_units_random_planes_with_ring_pressure(emu, oracle):
    rng = np.random.default_rng(5)
    for trial in range(24):
        w, h = int(rng.integers(1, 400)), int(rng.integers(1, 300))
        amp = int(rng.choice([1, 2, 4, 16, 64, 300, 2000, 30000]))
        dens = float(rng.choice([0.001, 0.01, 0.1, 0.5, 1.0]))
        mag = (rng.integers(0, amp + 1, (h, w)) * (rng.random((h, w)) < dens)).astype(np.uint16)
        plane = (mag | ((rng.integers(0, 2, (h, w)).astype(np.uint16) << 15) * (mag > 0))).astype(np.uint16)
        for sb in (0, 1, 3):
            for lsb in (0, 1, 3, 8):
                assert emu.code_unit(plane, 0, 0, w, h, sb, lsb) == oracle.code_unit(plane, 0, 0, w, h, sb, lsb), (trial, sb, lsb)


def test_coding_units_with_predictable_signs(emu, oracle):
    """the golomb wave has two forms of its per-bin work: a reduced one for chunks without a sign event in a Golomb bin (the
    rule: a sign is close to a coin toss) and the general one.  Planes whose signs are all alike make the sign contexts
    confident, so their events reach the Golomb bins: both forms run (counted) and agree with the oracle."""
    rng = np.random.default_rng(15)
    emu.chunk_stats()
    for trial, sign in enumerate((0, 1, 2)):
        w, h = int(rng.integers(150, 330)), int(rng.integers(60, 200))
        mag = rng.integers(0, 200, (h, w)).astype(np.uint16)
        sg = np.full((h, w), sign & 1, np.uint16) if sign < 2 else ((np.add.outer(np.arange(h), np.arange(w)) // 3) & 1).astype(np.uint16)
        plane = (mag | ((sg << 15) * (mag > 0))).astype(np.uint16)
        for sb in (0, 1, 3):
            for lsb in (0, 2, 4, 6):
                assert emu.code_unit(plane, 0, 0, w, h, sb, lsb) == oracle.code_unit(plane, 0, 0, w, h, sb, lsb), (trial, sb, lsb)
    st = emu.chunk_stats()
    assert st[2] > 100 and st[3] > 100, st


def test_slot_capacity_rule(emu, oracle):
    rng = np.random.default_rng(6)
    plane = rng.integers(0, 512, (40, 50)).astype(np.uint16)
    bits, payload = oracle.code_unit(plane, 0, 0, 50, 40, 0, 0)
    nbytes = (bits + 7) // 8
    for cap in (4, 64, (bits // 8) // 4 * 4, (bits // 8) // 4 * 4 + 4, nbytes + 8):
        got = emu.code_unit(plane, 0, 0, 50, 40, 0, 0, cap=cap)
        if bits // 8 < cap:
            assert got == (bits, payload)
        else:
            assert got[0] == -5


CASES = [(64, 64, 2, 0, 4, 1 << 20), (200, 160, 3, 1, 1, 1 << 20), (100, 75, 3, 2, 7, 1 << 20), (128, 128, 4, 3, 16, 5000),
         (96, 96, 2, 4, 3, 3000), (257, 131, 3, 5, 32, 1 << 20), (80, 80, 2, 6, 5, 900), (512, 512, 3, 0, 10, 1 << 20),
         (64, 64, 2, 0, 4, 27), (64, 64, 2, 0, 4, 28), (64, 64, 2, 0, 4, 29), (64, 64, 2, 0, 4, 60), (5, 5, 1, 0, 1, 4096),
         (24, 24, 3, 0, 9, 1 << 16), (24, 24, 3, 0, 10, 1 << 16)]


@pytest.mark.parametrize("case", CASES)
def test_pipeline_gray(emu, oracle, case):
    w, h, st, f, sg, q = case
    for mode in (0, 1):
        img = synth.gray_frame(w, h, 7, mode)
        a, b = oracle.compress([img], st, f, sg, q), emu.compress([img], st, f, sg, q)
        assert a[0] == b[0] and a[1] == b[1] and (a[0] not in (0, -5) or np.array_equal(a[2][0], b[2][0])), (b[0], b[3])


def test_pipeline_yuv(emu, oracle):
    for (w, h, st, f, sg, q) in [(64, 64, 2, 0, 4, 1 << 20), (128, 96, 3, 0, 10, 6000), (100, 100, 3, 2, 6, 20000)]:
        planes = synth.color_frame_yuv(w, h, 3)
        a, b = oracle.compress(planes, st, f, sg, q), emu.compress(planes, st, f, sg, q)
        assert a[0] == b[0] and a[1] == b[1] and all(np.array_equal(p, r) for p, r in zip(a[2], b[2]))


def test_slot_bound_overflow_is_reported(emu, oracle):
    img = synth.gray_frame(256, 256, 9, 0)          # noise: > 1 bit/pixel in the low planes
    want = oracle.compress([img], 1, 0, 1, 1 << 18)
    got1 = emu.compress([img], 1, 0, 1, 1 << 18, bpp=1)
    assert got1[3] == 1                     # bound too small -> flagged, host would retry
    got = emu.compress([img], 1, 0, 1, 1 << 18, bpp=24)
    assert got[3] == 0 and got[:2] == want[:2]


def test_planner_units_match_oracle_geometry(emu, oracle):
    for (w, h, ch, st, sg) in [(4096, 4096, 1, 5, 10), (2048, 2048, 1, 4, 16), (517, 389, 3, 4, 7), (100, 75, 1, 3, 32)]:
        n, units = emu.plan_units(w, h, ch, st, sg)
        assert n == (3 * st + 1) * 9 * sg * ch
        pk = oracle.packets(st, ch)
        # priority order: packet-major, segment-minor; rectangles tile each subband exactly
        for i, (lv, sb, lsb, c, _) in enumerate(pk):
            block = units[i * sg:(i + 1) * sg]
            assert (block[:, 5] == lv).all() and (block[:, 6] == sb).all() and (block[:, 7] == lsb).all() and (block[:, 4] == c).all()
            assert list(block[:, 8]) == list(range(sg))
            assert int((block[:, 2].astype(np.int64) * block[:, 3]).sum()) == _subband_area(w, h, lv, sb)


def test_chunk_table_work_list_covers_every_family_once(emu):
    """Plan::sig_blocks (the grid of family_events_kernel): every block of 64 chunks of every family (channel, level, subband,
    segment) exactly once, through one unit of the family; the families' table areas tile the per-frame area"""
    import ctypes as C
    emu.lib.emu_plan_sig_blocks.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
    for (w, h, ch, st, sg) in [(4096, 4096, 1, 5, 10), (2048, 2048, 1, 4, 16), (517, 389, 3, 4, 7), (100, 75, 1, 3, 32), (8192, 8192, 1, 6, 32),
                               (16384, 16384, 1, 2, 1)]:   # (a coding unit above 2^24 pixels: more than 4096 blocks of 64 chunks)
        buf = np.zeros((400000, 5), np.uint32)
        sig_bytes = C.c_size_t(0)
        n = emu.lib.emu_plan_sig_blocks(w, h, ch, st, sg, buf.ctypes.data, len(buf), C.byref(sig_bytes))
        assert 0 < n <= len(buf)
        e = buf[:n]
        fams = {}
        for unit, blk, off, nchunks, _ in e.tolist():
            fams.setdefault((unit, off, nchunks), []).append(blk)
        assert len(fams) == (3 * st + 1) * sg * ch
        areas = []
        for (unit, off, nchunks), blks in fams.items():
            assert sorted(blks) == list(range((nchunks + 63) // 64))   # every block once
            areas.append((off, nchunks))
        areas.sort()
        for (o0, n0), (o1, _) in zip(areas, areas[1:]):
            assert o0 + n0 <= o1                                       # table areas do not overlap
        assert areas[-1][0] + areas[-1][1] <= sig_bytes.value


def test_sub_range_size_follows_the_geometry(emu):
    """plan.hpp auto_split_chunks: the smallest piece size whose pieces over the lower half of the bit planes stay within two per compute
    unit (256 of them); the values the GPU measurements of round 6 chose"""
    import ctypes as C
    emu.lib.emu_auto_split.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    for (w, h, st, sg, chunks, split) in [(4096, 4096, 5, 10, 2048, True), (4096, 4096, 5, 12, 2048, True), (4096, 4096, 5, 16, 1536, True), (4096, 2048, 5, 6, 1024, True),
                                          (2048, 2048, 4, 4, 1024, True), (3000, 2000, 5, 10, 1024, True),
                                          (2048, 2048, 4, 16, 1024, False), (1024, 1024, 3, 8, 1024, False), (8192, 8192, 6, 32, 4096, True)]:
        c, x = C.c_uint32(0), C.c_uint32(0)
        assert emu.lib.emu_auto_split(w, h, 1, st, sg, 256, C.byref(c), C.byref(x)) == 0
        assert c.value == chunks and (x.value > 0) == split, (w, h, sg, c.value, x.value)
        # pieces over ALL planes (the run-time routing keeps the dense ones): at most 9 / 5 of the budget of 512
        assert x.value <= 512 * 9 // 5


def test_position_major_launch_order_is_a_bijection(emu):
    """plan.hpp position_major (the one-dimensional grid of a batch's pipeline kernel): every (frame, launch position) exactly once for
    unit counts that are and are not multiples of eight and 1 .. 33 frames; in full groups of eight, workgroup b sits on launch position
    b % 8 (mod 8): the XCD of its family"""
    import ctypes as C
    f, p = C.c_uint32(0), C.c_uint32(0)
    emu.lib.emu_position_major.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    for per_frame in (1, 7, 8, 9, 63, 64, 1440 % 97 + 90, 1872, 1875):
        for n_frames in (1, 2, 3, 8, 32, 33):
            seen = set()
            for b in range(per_frame * n_frames):
                emu.lib.emu_position_major(b, per_frame, n_frames, C.byref(f), C.byref(p))
                assert f.value < n_frames and p.value < per_frame
                seen.add((f.value, p.value))
                if p.value < per_frame - per_frame % 8:
                    assert p.value % 8 == b % 8
            assert len(seen) == per_frame * n_frames


def test_a_family_is_one_rectangle_even_under_the_stale_grid_quirk(emu):
    """The units of a family share the chunk table and the event bytes, so they must code the SAME rectangle.  Under quirk P1 (a
    failed segment grid keeps the previous packet's one) the planes of one (channel, level, subband, segment) can have different
    rectangles: the planner must make those families of their own (151 x 66, 5 stages, 15 segments: 224 such units)."""
    import ctypes as C
    emu.lib.emu_plan_families.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]
    for (w, h, ch, st, sg) in [(151, 66, 1, 5, 15), (151, 66, 3, 5, 15), (100, 75, 1, 3, 32), (517, 389, 3, 4, 7), (2048, 2048, 1, 4, 16)]:
        buf = np.zeros((20000, 6), np.uint32)
        nf = C.c_uint32(0)
        n = emu.lib.emu_plan_families(w, h, ch, st, sg, buf.ctypes.data, len(buf), C.byref(nf))
        assert 0 < n <= len(buf)
        fams = {}
        for fam, off, x0, y0, uw, uh in buf[:n].tolist():
            assert fams.setdefault(fam, (off, x0, y0, uw, uh)) == (off, x0, y0, uw, uh)
        assert sorted(fams) == list(range(nf.value))
        areas = sorted((off, (uw * uh + 63) // 64) for off, _, _, uw, uh in fams.values())
        for (o0, n0), (o1, _) in zip(areas, areas[1:]):
            assert o0 + n0 <= o1
        if (w, h, sg) == (151, 66, 15):
            assert nf.value > (3 * st + 1) * sg * ch                     # the quirk is exercised
        if (w, h) == (2048, 2048):
            assert nf.value == (3 * st + 1) * sg * ch


def test_launch_orders_are_permutations_and_keep_families_on_their_xcd(emu):
    """Plan::work_order / Plan::split_launch: every unit (and every sub-range workgroup) exactly once; the nine planes of a family sit
    at positions of one residue mod 8 (workgroup b runs on XCD b % 8: one L2 reads the family's coefficients) -- all of them where the
    eight lists are equally long, all but the entries borrowed for an exhausted list's positions otherwise; family indices are dense"""
    import ctypes as C
    emu.lib.emu_plan_orders.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]
    for (w, h, ch, st, sg, split) in [(4096, 4096, 1, 5, 10, 0), (4096, 4096, 1, 5, 10, 3072), (2048, 2048, 1, 4, 16, 0), (8192, 8192, 1, 6, 32, 0),
                                      (517, 389, 3, 4, 7, 0), (1024, 768, 1, 2, 2, 128)]:
        buf = np.zeros((40000, 3), np.uint32)
        nf = C.c_uint32(0)
        n = emu.lib.emu_plan_orders(w, h, ch, st, sg, split, buf.ctypes.data, len(buf), C.byref(nf))
        assert 0 < n <= len(buf)
        e = buf[:n]
        units = (3 * st + 1) * 9 * sg * ch
        assert nf.value == (3 * st + 1) * sg * ch and set(e[:, 1].tolist()) == set(range(nf.value))
        whole = e[e[:, 2] == 0]
        assert sorted(whole[:, 0].tolist()) == list(range(units))          # every unit once as itself
        if not split:
            assert n == units
        off = 0
        home = {}
        for pos, (_, fam, _) in enumerate(e.tolist()):
            if fam not in home:
                home[fam] = pos % 8
            elif home[fam] != pos % 8:
                off += 1
        if nf.value % 8 == 0:                                            # (else the eight lists differ in length by whole families: entries are borrowed)
            assert off <= n // 50, (w, h, split, off, n)                 # (C2: 0 of 1 440, 12 of 1 710 with sub-ranges)
        if (w, h, split) == (4096, 4096, 0):
            assert off == 0


def _subband_area(w, h, lv, sb):
    low = lambda d, l: -(-d // (1 << l))
    high = lambda d, l: low(d, l - 1) // 2
    sw = low(w, lv) if sb in (0, 2) else high(w, lv)
    sh = low(h, lv) if sb in (0, 1) else high(h, lv)
    return sw * sh


def _u8_cases():
    rng = np.random.default_rng(21)
    for trial in range(40):
        w, h = int(rng.integers(8, 150)), int(rng.integers(8, 150))
        st = int(rng.integers(1, 6))
        while ((w + (1 << st) - 1) >> st) < 3 or ((h + (1 << st) - 1) >> st) < 3:
            st -= 1
        sg = int(rng.integers(1, 17))
        sg = min(sg, ((w + (1 << st) - 1) >> st) * ((h + (1 << st) - 1) >> st))
        ch = 3 if trial % 3 == 0 else 1
        amp, base = int(rng.choice([4, 8, 16, 30, 60, 127, 255])), int(rng.choice([0, 10, 40]))
        planes = [np.clip(base + rng.integers(0, amp + 1, (h, w)), 0, 255).astype(np.uint8) for _ in range(ch)]
        quota = int(rng.choice([w * h * 2 + 100, w * h * 2 + 100, 500, 3000]))
        yield planes, st, int(rng.integers(0, 7)), sg, quota


def test_uint8_twins_pipeline(emu, oracle):
    """7 planes, int8 overflow rules, the 300-packet table and the upward final order of the uint8 YUV variant."""
    seen = set()
    for planes, st, filt, sg, quota in _u8_cases():
        a, b = oracle.compress_u8(planes, st, filt, sg, quota), emu.compress_u8(planes, st, filt, sg, quota)
        seen.add(a[0])
        assert a[0] == b[0] and a[1] == b[1], (len(planes), planes[0].shape, st, filt, sg, quota, a[0], b[0])
        if a[0] in (0, -5):
            assert all(np.array_equal(x, y) for x, y in zip(a[2], b[2]))
    assert {0, -1, -5} <= seen, seen


def test_blank_chunk_shortcuts(emu, oracle):
    """Long runs of blank chunks (all-zero events of context 0 wandering through the Golomb bins and the count rescales),
    interrupted by isolated significant pixels: the pixel / compaction / walker / golomb shortcuts against the oracle."""
    rng = np.random.default_rng(17)
    for (w, h, spikes) in [(300, 257, 0), (300, 257, 3), (640, 200, 40), (64, 1000, 7), (1000, 70, 200)]:
        plane = np.zeros((h, w), np.uint16)
        for _ in range(spikes):
            plane[int(rng.integers(0, h)), int(rng.integers(0, w))] = int(rng.integers(1, 300)) | (int(rng.integers(0, 2)) << 15)
        for sb in (0, 1, 2, 3):
            for lsb in (0, 3, 8):
                assert emu.code_unit(plane, 0, 0, w, h, sb, lsb) == oracle.code_unit(plane, 0, 0, w, h, sb, lsb), (w, h, spikes, sb, lsb)


def test_random_wave_schedules(emu, oracle):
    """The eight waves of a coding unit under a random scheduler (one wave at a time, picked at random, runs one chunk if
    its inputs are there): every interleaving must give the oracle's bits, and none may dead-lock (bits = -10)."""
    rng = np.random.default_rng(23)
    for trial in range(12):
        w, h = int(rng.integers(64, 400)), int(rng.integers(64, 300))
        if trial % 3 == 0:
            v = rng.normal(0, float(rng.choice([2, 6, 20])), (h, w)).astype(np.int32)
        elif trial % 3 == 1:
            v = (rng.integers(-300, 300, (h, w)) * (rng.random((h, w)) < 0.05)).astype(np.int32)
        else:
            v = rng.normal(0, 3, (h, w))
            v[:, : w // 3] *= 0.1
            v[h // 2:, :] *= 20
            v = v.astype(np.int32)
        plane = np.ascontiguousarray(np.minimum(np.abs(v), 32767).astype(np.uint16) | ((v < 0).astype(np.uint16) << 15))
        for sb, lsb in ((0, 0), (3, 1), (1, 4)):
            want = oracle.code_unit(plane, 0, 0, w, h, sb, lsb)
            for k in range(4):
                assert emu.code_unit_random(plane, 0, 0, w, h, sb, lsb, 1000 * trial + k) == want, (trial, sb, lsb, k)


def test_waves_as_real_threads(oracle, tmp_path):
    """The lane-loop build with every wave on its own CPU thread and real waits (tests/emu/threads_main.cpp): the hand-off
    protocol under true concurrency must neither dead-lock (bits = -10) nor change a bit.  (The same program built with
    -fsanitize=thread and tests/emu/tsan.supp is how the protocol was checked for data races.)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "threads_main")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-Wno-unknown-pragmas", "-o", exe,
                           os.path.join(root, "tests", "emu", "threads_main.cpp")])

    def fnv(data):
        hsh = 1469598103934665603
        for x in data:
            hsh = ((hsh ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return "%016x" % hsh

    rng = np.random.default_rng(29)
    for trial in range(6):
        w, h = int(rng.integers(64, 300)), int(rng.integers(32, 200))
        v = rng.normal(0, float(rng.choice([2, 6, 20])), (h, w)) if trial % 2 == 0 else \
            rng.integers(-300, 300, (h, w)) * (rng.random((h, w)) < 0.1)
        v = v.astype(np.int32)
        plane = np.ascontiguousarray(np.minimum(np.abs(v), 32767).astype(np.uint16) | ((v < 0).astype(np.uint16) << 15))
        plane.tofile(tmp_path / "plane.raw")
        for sb, lsb in ((0, 0), (3, 2)):
            bits, payload = oracle.code_unit(plane, 0, 0, w, h, sb, lsb)
            r = subprocess.run([exe, str(tmp_path / "plane.raw"), str(w), str(h), str(sb), str(lsb), "3"], capture_output=True,
                               text=True, timeout=300)
            assert r.returncode == 0 and r.stdout.split("\n")[:3] == [f"{bits} {fnv(payload)}"] * 3, (trial, sb, lsb, r.stdout, r.stderr[:500])


def test_waves_as_real_threads_wind_down(oracle, tmp_path):
    """Abandoning a unit under true concurrency (same threaded build): the progressive-mode stop raised by another thread at
    an arbitrary moment, and a payload slot that is too small.  Every wave must leave -- no dead-lock (-10), no hang (the
    subprocess time-out) -- and a unit that does finish must still be bit-exact."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "threads_main")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-Wno-unknown-pragmas", "-o", exe,
                           os.path.join(root, "tests", "emu", "threads_main.cpp")])

    def fnv(data):
        hsh = 1469598103934665603
        for x in data:
            hsh = ((hsh ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return "%016x" % hsh

    rng = np.random.default_rng(31)
    seen = set()
    for trial in range(3):
        w, h = int(rng.integers(200, 400)), int(rng.integers(150, 300))
        v = rng.normal(0, float(rng.choice([6, 20])), (h, w)).astype(np.int32)
        plane = np.ascontiguousarray(np.minimum(np.abs(v), 32767).astype(np.uint16) | ((v < 0).astype(np.uint16) << 15))
        plane.tofile(tmp_path / "plane.raw")
        bits, payload = oracle.code_unit(plane, 0, 0, w, h, 0, 0)
        good = f"{bits} {fnv(payload)}"
        # the stop arrives somewhere inside (or just after) the unit's run time
        for stop_us, cap_div in ((3000000, 1), (20000, 1), (0, 40), (20000, 40)):
            r = subprocess.run([exe, str(tmp_path / "plane.raw"), str(w), str(h), "0", "0", "8", str(stop_us), str(cap_div)],
                               capture_output=True, text=True, timeout=600)
            lines = [x for x in r.stdout.split("\n") if x]
            assert r.returncode == 0 and len(lines) == 8, (r.stdout, r.stderr[:500])
            for x in lines:
                assert x in (good, "-3 0", "-5 0"), (trial, stop_us, cap_div, x)
                assert not (x == "-3 0" and stop_us == 0) and not (x == "-5 0" and cap_div == 1), (trial, stop_us, cap_div, x)
                seen.add(x if x != good else "ok")
    assert seen == {"ok", "-3 0", "-5 0"}, seen
