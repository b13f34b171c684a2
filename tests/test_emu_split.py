"""Sub-ranges (csrc/coder_core.hpp "Sub-ranges"): a coding unit cut into K pieces, one workgroup each -- started cold with
the exact adaptive counts, spliced where a workgroup's complete coder state equals a later one's snapshot -- gives the
oracle's payload bit for bit, whether the workgroups meet or not.  CPU lane-loop build of the kernel source (tests/emu),
with snapshots every 4 chunks so that small units exercise the matching, and with the product's spacing on a unit of the
size of a level-1 segment of the headline frame."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from icer_compression_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")


def _lib(snap_every):
    src = os.path.join(ROOT, "tests", "emu", "coder_emu.cpp")
    so = os.path.join(ROOT, "tests", "emu", f"libcoder_emu_snap{snap_every}.so")
    csrc = os.path.join(ROOT, "icer_compression_amd", "csrc")
    newest = max([os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc)] +
                 [os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "tests", "emu", "coder_emu_drivers.hpp"))])
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", f"-DICER_SNAP_EVERY={snap_every}", "-o", so, src])
    L = C.CDLL(so)
    L.emu_code_unit_split.restype = C.c_long
    L.emu_code_unit_split.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, u8p, C.c_size_t, C.c_uint, C.c_uint,
                                      C.POINTER(C.c_uint)]
    return L


def _split(L, plane, x, y, w, h, sb, lsb, k, order):
    cap = w * h * 3 + 256
    out = np.zeros(cap + 8, np.uint8)
    m = C.c_uint(0)
    bits = L.emu_code_unit_split(plane.ctypes.data + 2 * (y * plane.shape[1] + x), w, h, plane.shape[1], sb, lsb, out, cap, k, order, C.byref(m))
    return bits, bytes(out[: (max(bits, 0) + 7) // 8]), m.value


def _coefficients(oracle, w, h, seed, mode, stages):
    rc, c = oracle.dwt(synth.gray_frame(w, h, seed, mode), stages, 0)
    c = np.ascontiguousarray(c)
    oracle.lib.orc_sign_magnitude.argtypes = [u16p, C.c_size_t]
    oracle.lib.orc_sign_magnitude(c.reshape(-1), c.size)
    return c


def test_small_units_cut_into_sub_ranges(oracle):
    L = _lib(4)
    rng = np.random.default_rng(5)
    cases = matched = 0
    for case in range(40):
        c = _coefficients(oracle, 256, 256, 100 + case, int(rng.integers(0, 2)), 2)
        w, h = int(rng.integers(20, 200)), int(rng.integers(20, 120))
        x, y, sb, lsb = int(rng.integers(0, 256 - w)), int(rng.integers(0, 256 - h)), int(rng.integers(0, 4)), int(rng.integers(0, 6))
        want = oracle.code_unit(c, x, y, w, h, sb, lsb)
        for k in (2, 3, 5, 8):
            if (w * h + 63) // 64 < 2 * k:
                continue
            for order in (0, 1, 2):                # which workgroups have finished when another one gets to their snapshots
                got = _split(L, c, x, y, w, h, sb, lsb, k, order)
                assert (got[0], got[1]) == want, (case, w, h, sb, lsb, k, order)
                cases += 1
                matched += got[2]
    assert cases > 300 and matched > 100           # the splice is exercised, not only the run-to-the-end path


def test_level1_sized_unit_with_the_products_snapshot_spacing(oracle):
    L = _lib(64)
    c = _coefficients(oracle, 2048, 2048, 12345, 1, 4)
    for lsb, k, order, min_matches in ((0, 8, 0, 7), (2, 4, 0, 3), (1, 8, 2, 3), (0, 8, 1, 0), (4, 8, 0, 0)):
        want = oracle.code_unit(c, 1024, 0, 682, 614, 1, lsb)
        got = _split(L, c, 1024, 0, 682, 614, 1, lsb, k, order)
        assert (got[0], got[1]) == want, (lsb, k, order)
        assert got[2] >= min_matches, (lsb, k, order, got[2])      # dense planes meet within 1024 chunks; sparse ones need not
