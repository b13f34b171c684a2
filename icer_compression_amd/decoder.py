"""Host-side mirror of the reference's DECODING interface over the C ABI of libicer_hip_dec.so
(include/icer_hip_dec.h).

STATUS: the device code behind it is checked against the decoder oracle in its CPU builds (tests/test_emu_decoder.py) and,
on an MI355X, by tests/test_gpu_decoder.py (gray / YUV, 16 / 8 bit, damaged streams, golden digests, the batch object;
all in the default `pytest -m gpu` run, HISTORY.md 6b (summary: DESIGN.md 8)).

The reference's callers do (example/src/example_decode.c, example/src/icer_util.c `decompress`)
    icer_get_image_dimensions(stream, len, &w, &h);  buf = malloc(w * h * 2);
    rc = icer_decompress_image_uint16(buf, &w, &h, w * h, stream, len, stages, filt, segments);
The functions below keep those names, argument order and return codes, numpy arrays standing in for the pointers.
There is no CPU fallback: every decompress call needs libicer_hip_dec.so and a HIP device and fails loudly otherwise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libicer_hip_dec.so")
ICER_DECODER_OUT_OF_DATA = -7
ICER_DECODED_INVALID_DATA = -8

_lib = None
_sz = C.c_size_t
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def bind(path: str) -> C.CDLL:
    """dlopen a build of the decoder library (RTLD_LOCAL: it exports the same icer_* names as the reference) and declare
    its entry points.  (tests/test_emu_decoder.py binds a CPU mock build of decoder.hip's host pipeline this way.)"""
    lib = C.CDLL(path, mode=os.RTLD_LOCAL)
    lib.icer_get_image_dimensions.argtypes = [_u8p, _sz, C.POINTER(_sz), C.POINTER(_sz)]
    tail = [C.POINTER(_sz), C.POINTER(_sz), _sz, _u8p, _sz, C.c_uint8, C.c_int, C.c_uint8]
    for name, n in (("icer_decompress_image_uint16", 1), ("icer_decompress_image_yuv_uint16", 3),
                    ("icer_decompress_image_uint8", 1), ("icer_decompress_image_yuv_uint8", 3)):
        getattr(lib, name).argtypes = [C.c_void_p] * n + tail
    lib.icerx_decoder_last_error.restype = C.c_char_p
    return lib


def load_library() -> C.CDLL:
    """libicer_hip_dec.so of this package"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} is missing: run `python -m icer_compression_amd.build` (there is no CPU fallback)")
        _lib = bind(LIB_PATH)
    return _lib


def icer_get_image_dimensions(stream: bytes, lib=None):
    """-> (rc, w, h)"""
    lib = lib or load_library()
    buf = np.frombuffer(stream, dtype=np.uint8).copy() if len(stream) else np.zeros(1, np.uint8)
    w, h = _sz(0), _sz(0)
    rc = lib.icer_get_image_dimensions(buf, len(stream), C.byref(w), C.byref(h))
    return rc, w.value, h.value


def decompress(stream: bytes, channels: int, stages: int, filt: int, segments: int, bufsize: int | None = None, bits: int = 16,
               lib=None):
    """icer_decompress_image_[yuv_]uint16 / _uint8 on a host stream -> (rc, w, h, [flat planes of bufsize samples])."""
    lib = lib or load_library()
    if channels not in (1, 3):
        raise ValueError("channels must be 1 or 3")
    if bufsize is None:
        rc, w, h = icer_get_image_dimensions(stream, lib)
        bufsize = w * h if rc == 0 else 0
    buf = np.frombuffer(stream, dtype=np.uint8).copy() if len(stream) else np.zeros(1, np.uint8)
    planes = [np.zeros(max(bufsize, 1), np.uint16 if bits == 16 else np.uint8) for _ in range(channels)]
    fn = getattr(lib, "icer_decompress_image_" + ("yuv_" if channels == 3 else "") + ("uint16" if bits == 16 else "uint8"))
    w, h = _sz(0), _sz(0)
    rc = fn(*[p.ctypes.data for p in planes], C.byref(w), C.byref(h), bufsize, buf, len(stream), stages, filt, segments)
    return rc, w.value, h.value, planes


class Decoder:
    """Batch / device-resident extension (icerx_decoder_*, include/icer_hip_dec.h Part 2)."""

    def __init__(self, channels: int, stages: int, filt: int, segments: int, bits: int = 16, device: int = -1, lib=None):
        self.lib = lib or load_library()
        self.channels, self.bits = channels, bits
        self.lib.icerx_decoder_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int]
        self.lib.icerx_decoder_destroy.argtypes = [C.c_void_p]
        self.lib.icerx_decoder_destroy.restype = None
        tail = [C.POINTER(_sz), C.POINTER(_sz), C.c_void_p, _sz, C.POINTER(C.c_int), C.POINTER(_sz), C.POINTER(_sz)]
        self.lib.icerx_decode_host.argtypes = [C.c_void_p, C.c_int, C.c_void_p] + tail
        self.lib.icerx_decode_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p] + tail
        self.handle = C.c_void_p()
        rc = self.lib.icerx_decoder_create(C.byref(self.handle), device, channels, stages, filt, segments, bits)
        if rc != 0:
            raise RuntimeError(f"icerx_decoder_create: {rc} {self.lib.icerx_decoder_last_error().decode()}")

    def close(self):
        if self.handle:
            self.lib.icerx_decoder_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _pack(self, streams):
        lens = [len(s) for s in streams]
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64) if streams else np.zeros(0, np.uint64)
        blob = np.frombuffer(b"".join(streams), dtype=np.uint8).copy() if sum(lens) else np.zeros(1, np.uint8)
        n = len(streams)
        return blob, (_sz * n)(*[int(o) for o in offs]), (_sz * n)(*lens)

    def decode_host(self, streams, frame_stride: int):
        """-> (rc, [(rc_k, w_k, h_k, [flat planes])]) for a list of streams in host memory"""
        n = len(streams)
        blob, offs, lens = self._pack(streams)
        dt = np.uint16 if self.bits == 16 else np.uint8
        planes = [np.zeros(max(frame_stride, 1), dt) for _ in range(n * self.channels)]
        ptrs = (C.c_void_p * max(len(planes), 1))(*[p.ctypes.data for p in planes])
        rcs, ws, hs = (C.c_int * max(n, 1))(), (_sz * max(n, 1))(), (_sz * max(n, 1))()
        rc = self.lib.icerx_decode_host(self.handle, n, blob.ctypes.data, offs, lens, ptrs, frame_stride, rcs, ws, hs)
        return rc, [(rcs[k], ws[k], hs[k], planes[k * self.channels: (k + 1) * self.channels]) for k in range(n)]

    def decode_device(self, n: int, d_data: int, offsets, lens, d_out: int, frame_stride: int):
        """raw device pointers (e.g. torch tensors' data_ptr()); -> (rc, rcs, ws, hs)"""
        offs = (_sz * max(n, 1))(*[int(o) for o in offsets])
        ln = (_sz * max(n, 1))(*[int(x) for x in lens])
        rcs, ws, hs = (C.c_int * max(n, 1))(), (_sz * max(n, 1))(), (_sz * max(n, 1))()
        rc = self.lib.icerx_decode_device(self.handle, n, d_data, offs, ln, d_out, frame_stride, rcs, ws, hs)
        return rc, list(rcs)[:n], list(ws)[:n], list(hs)[:n]
