import os
import sys

import pytest

# the host-fed batch tests keep six streams per device busy: more hardware queues than the HIP runtime's default of 4 (a
# process-wide setting, read once when the runtime starts; the library itself never touches the environment)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_first():
    """torch ships its own HIP runtime; when it is initialised AFTER libicer_hip*.so (which link the system one) it finds
    no device.  On a GPU box bring torch's runtime up first, so both see the GPU in one process (bench.py does the same)."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:                                   # noqa: BLE001  (no torch / no GPU: nothing to order)
        pass
    yield


@pytest.fixture(scope="session")
def oracle():
    from oracle.binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle.binding import Reference, have_reference, build
    build()
    if not have_reference():
        pytest.skip("oracle/_ref/libicer_ref.so not built (needs /root/reference)")
    return Reference()


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def emu():
    """CPU lane-loop build of the kernel cores (tests/emu) -- debugging aid for the no-GPU container."""
    import ctypes as C
    import subprocess
    import numpy as np
    src = os.path.join(ROOT, "tests", "emu", "coder_emu.cpp")
    so = os.path.join(ROOT, "tests", "emu", "libcoder_emu.so")
    csrc = os.path.join(ROOT, "icer_compression_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc))
    if not os.path.exists(so) or os.path.getmtime(so) < max(newest, os.path.getmtime(src)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-o", so, src])
    L = C.CDLL(so)
    u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
    u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
    L.emu_code_unit.restype = C.c_long
    L.emu_code_unit.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, u8p, C.c_size_t]
    L.emu_dwt.argtypes = [u16p, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
    L.emu_compress.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int,
                               C.c_size_t, C.c_uint, u8p, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    L.emu_plan_units.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.emu_dwt_fast_tiles.restype = C.c_ulonglong

    class Emu:
        lib = L

        @staticmethod
        def compress(planes, stages, filt, segs, quota, bpp=3):
            work = [np.ascontiguousarray(p, dtype=np.uint16).copy() for p in planes]
            h, w = work[0].shape
            ptrs = (C.c_void_p * len(work))(*[p.ctypes.data for p in work])
            out = np.zeros(max(quota, 1) + 64, np.uint8)
            used, bo = C.c_size_t(0), C.c_int(0)
            rc = L.emu_compress(ptrs, len(work), w, h, stages, filt, segs, quota, bpp, out, C.byref(used), C.byref(bo))
            return rc, bytes(out[: used.value]), work, bo.value

        @staticmethod
        def compress_u8(planes, stages, filt, segs, quota, bpp=3):
            """uint8 twins through the emulated pipeline; returns the planes as int8 sign-magnitude bytes"""
            work = [np.ascontiguousarray(p, dtype=np.uint8).astype(np.int8).astype(np.int16).view(np.uint16).copy() for p in planes]
            h, w = work[0].shape
            ptrs = (C.c_void_p * len(work))(*[p.ctypes.data for p in work])
            out = np.zeros(max(quota, 1) + 64, np.uint8)
            used, bo = C.c_size_t(0), C.c_int(0)
            L.emu_compress_bits.argtypes = L.emu_compress.argtypes + [C.c_int]
            rc = L.emu_compress_bits(ptrs, len(work), w, h, stages, filt, segs, quota, bpp, out, C.byref(used), C.byref(bo), 8)
            back = [(((p >> 8) & 0x80) | (p & 0x7F)).astype(np.uint8) for p in work]
            return rc, bytes(out[: used.value]), back, bo.value

        @staticmethod
        def code_unit(plane, x, y, w, h, sb, lsb, cap=None):
            cap = cap if cap is not None else ((w * h * 3 + 64) + 3) // 4 * 4
            out = np.zeros(cap + 8, np.uint8)
            bits = L.emu_code_unit(plane.ctypes.data + 2 * (y * plane.shape[1] + x), w, h, plane.shape[1], sb, lsb, out, cap)
            return bits, bytes(out[: (max(bits, 0) + 7) // 8])

        @staticmethod
        def code_unit_random(plane, x, y, w, h, sb, lsb, seed):
            """the same unit under the random wave scheduler (code_unit_emu_random); bits = -10: the waves dead-locked"""
            cap = ((w * h * 3 + 64) + 3) // 4 * 4
            out = np.zeros(cap + 8, np.uint8)
            L.emu_code_unit_random.restype = C.c_long
            L.emu_code_unit_random.argtypes = L.emu_code_unit.argtypes + [C.c_uint32]
            bits = L.emu_code_unit_random(plane.ctypes.data + 2 * (y * plane.shape[1] + x), w, h, plane.shape[1], sb, lsb, out, cap, seed)
            return bits, bytes(out[: (max(bits, 0) + 7) // 8])

        @staticmethod
        def chunk_stats(reset=True):
            """[fast-path chunks, exact-path chunks, golomb wave: chunks in the general form, in the reduced form] since the last reset"""
            out = (C.c_ulonglong * 4)()
            L.emu_chunk_stats(out, 1 if reset else 0)
            return list(out)

        @staticmethod
        def dwt(img, stages, filt):
            b = np.ascontiguousarray(img, dtype=np.uint16).copy()
            rc = L.emu_dwt(b, b.shape[1], b.shape[0], stages, filt)
            return rc, b

        @staticmethod
        def plan_units(w, h, channels, stages, segments):
            buf = np.zeros((8000, 9), np.uint32)
            n = L.emu_plan_units(w, h, channels, stages, segments, buf.ctypes.data, 8000)
            return n, buf[: max(n, 0)]
    return Emu
