"""ctypes loaders for the two CPU checkers (TEST INFRASTRUCTURE ONLY).

* ``Oracle``    -- oracle/liboracle.so, our plain-C restatement (icer_oracle.c)
* ``Reference`` -- oracle/_ref/libicer_ref.so, the untouched reference sources compiled by
                   oracle/Makefile (+ our ref_tap.c).  Present wherever ``make -C oracle`` ran
                   with /root/reference available; the prebuilt file travels to the GPU box.

Both expose the same Python surface so tests can be parametrised over them.  Nothing in the
product package imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libicer_ref.so")

u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(force: bool = False) -> None:
    """(Re)build liboracle.so, and _ref/libicer_ref.so when the reference sources exist."""
    need = force or not os.path.exists(ORACLE_SO) or \
        os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "icer_oracle.c"))
    if need:
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/lib_icer/src") and (force or not os.path.exists(REF_SO)):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


def build_examples(force: bool = False) -> bool:
    """The reference's own example programs, unmodified, linked with the reference build (_ref/ref_*) and with the product
    libraries (_ref/hip_*): `make examples` (tests/test_gpu_examples.py).  Only where the reference sources are mounted and
    the product libraries have been built; the binaries travel to the GPU box with the snapshot."""
    progs = [os.path.join(HERE, "_ref", f"{k}_{p}") for k in ("ref", "hip") for p in ("compress", "decompress", "compress_color", "decompress_color", "icer_util")]
    if not os.path.isdir("/root/reference/example/src") or not os.path.exists(REF_SO):
        return all(os.path.exists(p) for p in progs)
    if force or not all(os.path.exists(p) for p in progs):
        subprocess.check_call(["make", "-C", HERE, "examples"], stdout=subprocess.DEVNULL)
    return all(os.path.exists(p) for p in progs)


def have_reference() -> bool:
    return os.path.exists(REF_SO)


class _OutBuf(C.Structure):          # icer_output_data_buf_typedef, icer.h:307-312
    _fields_ = [("size_used", C.c_size_t), ("size_allocated", C.c_size_t),
                ("data_start", C.c_void_p), ("rearrange_start", C.c_void_p)]


class Oracle:
    name = "oracle"

    def __init__(self):
        build()
        L = self.lib = C.CDLL(ORACLE_SO, mode=os.RTLD_LOCAL)
        L.orc_dwt_stages_u16.argtypes = [u16p, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
        L.orc_code_unit.restype = C.c_long
        L.orc_code_unit.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, u8p, C.c_size_t]
        L.orc_partition_make.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint]
        L.orc_crc32.restype = C.c_uint32
        L.orc_crc32.argtypes = [u8p, C.c_size_t]
        L.orc_pick_bin.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_compress_u16.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                       C.c_uint, C.c_size_t, u8p, C.POINTER(C.c_size_t)]

    def dwt(self, img: np.ndarray, stages: int, filt: int):
        buf = np.ascontiguousarray(img, dtype=np.uint16).copy()
        rc = self.lib.orc_dwt_stages_u16(buf, buf.shape[1], buf.shape[0], stages, filt)
        return rc, buf

    def code_unit(self, plane: np.ndarray, x, y, w, h, subband, lsb):
        """Payload of the coding unit covering plane[y:y+h, x:x+w] (sign-magnitude words)."""
        plane = np.ascontiguousarray(plane, dtype=np.uint16)
        cap = w * h * 3 + 64
        out = np.zeros(cap + 8, dtype=np.uint8)
        ptr = plane.ctypes.data + 2 * (y * plane.shape[1] + x)
        bits = self._code_unit(ptr, w, h, plane.shape[1], subband, lsb, out, cap)
        return bits, bytes(out[: (max(bits, 0) + 7) // 8])

    def _code_unit(self, ptr, w, h, stride, subband, lsb, out, cap):
        return self.lib.orc_code_unit(ptr, w, h, stride, subband, lsb, out, cap)

    def partition(self, w, h, segments):
        p = (C.c_uint16 * 15)()
        rc = self.lib.orc_partition_make(p, w, h, segments)
        return rc, list(p)

    def crc32(self, data: bytes) -> int:
        a = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(0, np.uint8)
        return self.lib.orc_crc32(np.ascontiguousarray(a), len(data))

    def packets(self, stages, channels):
        class P(C.Structure):
            _fields_ = [("level", C.c_uint8), ("subband", C.c_uint8), ("lsb", C.c_uint8), ("chan", C.c_uint8),
                        ("priority", C.c_uint64)]
        arr = (P * 800)()
        n = self.lib.orc_packet_list(arr, stages, channels)
        return [(a.level, a.subband, a.lsb, a.chan, a.priority) for a in arr[:n]]

    def compress(self, planes, stages, filt, segments, quota):
        """planes: list of 1 or 3 (h, w) uint16 arrays.  Returns (rc, stream, mutated planes)."""
        work = [np.ascontiguousarray(p, dtype=np.uint16).copy() for p in planes]
        h, w = work[0].shape
        ptrs = (C.c_void_p * len(work))(*[p.ctypes.data for p in work])
        out = np.zeros(max(quota, 1), dtype=np.uint8)
        used = C.c_size_t(0)
        rc = self.lib.orc_compress_u16(ptrs, len(work), w, h, stages, filt, segments, quota, out, C.byref(used))
        return rc, bytes(out[: used.value]), work

    def compress_u8(self, planes, stages, filt, segments, quota):
        """uint8 twins: planes are (h, w) uint8 arrays (int8 storage).  Returns (rc, stream, mutated planes)."""
        work = [np.ascontiguousarray(p, dtype=np.uint8).copy() for p in planes]
        h, w = work[0].shape
        ptrs = (C.c_void_p * len(work))(*[p.ctypes.data for p in work])
        out = np.zeros(max(quota, 1), dtype=np.uint8)
        used = C.c_size_t(0)
        self.lib.orc_compress_u8.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_uint,
                                             C.c_size_t, u8p, C.POINTER(C.c_size_t)]
        rc = self.lib.orc_compress_u8(ptrs, len(work), w, h, stages, filt, segments, quota, out, C.byref(used))
        return rc, bytes(out[: used.value]), work


    def decompress(self, stream: bytes, channels, stages, filt, segments, bufsize=None, bits=16):
        """Decoder restatement (orc_decompress_u16 / _u8).  Returns (rc, w, h, [planes])."""
        sz = C.c_size_t
        buf = np.frombuffer(stream, dtype=np.uint8).copy() if len(stream) else np.zeros(1, np.uint8)
        w, h = sz(0), sz(0)
        if bufsize is None:
            bufsize = _stream_dims(stream)
        dt = np.uint16 if bits == 16 else np.uint8
        planes = [np.zeros(max(bufsize, 1), dt) for _ in range(channels)]
        ptrs = (C.c_void_p * channels)(*[p.ctypes.data for p in planes])
        fn = self.lib.orc_decompress_u16 if bits == 16 else self.lib.orc_decompress_u8
        fn.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(sz), C.POINTER(sz), sz, u8p, sz, C.c_int, C.c_int, C.c_uint]
        rc = fn(ptrs, channels, C.byref(w), C.byref(h), bufsize, buf, len(stream), stages, filt, segments)
        return rc, w.value, h.value, planes


def _stream_dims(stream: bytes) -> int:
    """w * h from the first packet header of a stream written by an encoder (test helper)."""
    if len(stream) < 28:
        return 0
    return int.from_bytes(stream[8:12], "little") * int.from_bytes(stream[12:16], "little")


class Reference:
    """The reference library itself (symbols of lib_icer + ref_tap.c), loaded RTLD_LOCAL because it
    exports the same icer_* names as the product library."""
    name = "reference"

    def __init__(self):
        build()
        if not have_reference():
            raise FileNotFoundError(REF_SO)
        L = self.lib = C.CDLL(REF_SO, mode=os.RTLD_LOCAL)
        L.icer_init()
        L.icer_wavelet_transform_stages_uint16.argtypes = [u16p, C.c_size_t, C.c_size_t, C.c_uint8, C.c_int]
        L.ref_tap_code_unit.restype = C.c_long
        L.ref_tap_code_unit.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, u8p, C.c_size_t]
        L.ref_tap_partition.argtypes = [C.c_size_t, C.c_size_t, C.c_uint, C.c_void_p]
        L.icer_init_output_struct.argtypes = [C.POINTER(_OutBuf), C.c_void_p, C.c_size_t, C.c_size_t]
        L.icer_compress_image_uint16.argtypes = [u16p, C.c_size_t, C.c_size_t, C.c_uint8, C.c_int, C.c_uint8, C.POINTER(_OutBuf)]
        L.icer_compress_image_yuv_uint16.argtypes = [u16p, u16p, u16p, C.c_size_t, C.c_size_t, C.c_uint8, C.c_int,
                                                     C.c_uint8, C.POINTER(_OutBuf)]
        L.icer_compute_bin.argtypes = [C.c_uint32, C.c_uint32]
        L.crc32buf.restype = C.c_uint32
        L.crc32buf.argtypes = [u8p, C.c_size_t]

    def dwt(self, img, stages, filt):
        buf = np.ascontiguousarray(img, dtype=np.uint16).copy()
        rc = self.lib.icer_wavelet_transform_stages_uint16(buf, buf.shape[1], buf.shape[0], stages, filt)
        return rc, buf

    code_unit = Oracle.code_unit

    def _code_unit(self, ptr, w, h, stride, subband, lsb, out, cap):
        return self.lib.ref_tap_code_unit(ptr, w, h, stride, subband, lsb, out, cap)

    def partition(self, w, h, segments):
        p = (C.c_uint16 * 15)()
        rc = self.lib.ref_tap_partition(w, h, segments, p)
        return rc, list(p)

    def crc32(self, data: bytes) -> int:
        a = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(0, np.uint8)
        return self.lib.crc32buf(np.ascontiguousarray(a), len(data))

    def packets(self, stages, channels):
        """Sorted packet list left in the global table by a tiny real compress call."""
        side = 3 << stages
        planes = [np.zeros((side, side), np.uint16) for _ in range(channels)]
        self.compress(planes, stages, 0, 1, 1 << 20)
        out = []
        lv, sb, lsb, ch, pr = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_ulonglong()
        for i in range((3 * stages + 1) * 9 * channels):
            self.lib.ref_tap_get_packet(i, C.byref(lv), C.byref(sb), C.byref(lsb), C.byref(ch), C.byref(pr))
            out.append((lv.value, sb.value, lsb.value, ch.value, pr.value))
        return out

    def compress(self, planes, stages, filt, segments, quota):
        work = [np.ascontiguousarray(p, dtype=np.uint16).copy() for p in planes]
        h, w = work[0].shape
        buf = np.zeros(2 * quota + 64, dtype=np.uint8)
        ob = _OutBuf()
        rc = self.lib.icer_init_output_struct(C.byref(ob), buf.ctypes.data, buf.size, quota)
        assert rc == 0
        if len(work) == 1:
            rc = self.lib.icer_compress_image_uint16(work[0], w, h, stages, filt, segments, C.byref(ob))
        else:
            rc = self.lib.icer_compress_image_yuv_uint16(work[0], work[1], work[2], w, h, stages, filt, segments, C.byref(ob))
        return rc, bytes(buf[quota: quota + ob.size_used]), work

    def compress_u8(self, planes, stages, filt, segments, quota):
        work = [np.ascontiguousarray(p, dtype=np.uint8).copy() for p in planes]
        h, w = work[0].shape
        buf = np.zeros(2 * quota + 64, dtype=np.uint8)
        ob = _OutBuf()
        rc = self.lib.icer_init_output_struct(C.byref(ob), buf.ctypes.data, buf.size, quota)
        assert rc == 0
        sz, u8 = C.c_size_t, C.c_uint8
        if len(work) == 1:
            self.lib.icer_compress_image_uint8.argtypes = [u8p, sz, sz, u8, C.c_int, u8, C.POINTER(_OutBuf)]
            rc = self.lib.icer_compress_image_uint8(work[0], w, h, stages, filt, segments, C.byref(ob))
        else:
            self.lib.icer_compress_image_yuv_uint8.argtypes = [u8p, u8p, u8p, sz, sz, u8, C.c_int, u8, C.POINTER(_OutBuf)]
            rc = self.lib.icer_compress_image_yuv_uint8(work[0], work[1], work[2], w, h, stages, filt, segments, C.byref(ob))
        return rc, bytes(buf[quota: quota + ob.size_used]), work

    def decompress(self, stream: bytes, channels, stages, filt, segments):
        """The reference DECODER (icer_decompress_image_[yuv_]uint16, icer.h:461-466) on a stream: returns
        (rc, [planes]).  Used for encode -> decode round trips of the HIP encoder's output."""
        sz = C.c_size_t
        buf = np.frombuffer(stream, dtype=np.uint8).copy()
        w, h = sz(0), sz(0)
        self.lib.icer_get_image_dimensions.argtypes = [u8p, sz, C.POINTER(sz), C.POINTER(sz)]
        rc = self.lib.icer_get_image_dimensions(buf, buf.size, C.byref(w), C.byref(h))
        if rc != 0:
            return rc, []
        planes = [np.zeros((h.value, w.value), np.uint16) for _ in range(channels)]
        n = w.value * h.value
        if channels == 1:
            self.lib.icer_decompress_image_uint16.argtypes = [u16p, C.POINTER(sz), C.POINTER(sz), sz, u8p, sz, C.c_uint8, C.c_int, C.c_uint8]
            rc = self.lib.icer_decompress_image_uint16(planes[0], C.byref(w), C.byref(h), n, buf, buf.size, stages, filt, segments)
        else:
            self.lib.icer_decompress_image_yuv_uint16.argtypes = [u16p, u16p, u16p, C.POINTER(sz), C.POINTER(sz), sz, u8p, sz, C.c_uint8,
                                                                  C.c_int, C.c_uint8]
            rc = self.lib.icer_decompress_image_yuv_uint16(planes[0], planes[1], planes[2], C.byref(w), C.byref(h), n, buf, buf.size,
                                                           stages, filt, segments)
        return rc, planes

    def decompress_raw(self, stream: bytes, channels, stages, filt, segments, bufsize=None, bits=16):
        """Same surface as Oracle.decompress: (rc, w, h, [flat planes of bufsize samples])."""
        sz = C.c_size_t
        if bufsize is None:
            bufsize = _stream_dims(stream)
        # The reference decoder has no end-of-packet check: a unit whose contexts it cannot reproduce (any coefficient
        # above the coded planes) makes it read on, past the packet and past the end of the stream.  Zeros behind the
        # stream make that read defined here -- and equal to the restatement's rule (bits past the end read as 0).
        buf = np.zeros(len(stream) + 4 * bufsize + 4096, np.uint8)
        buf[: len(stream)] = np.frombuffer(stream, dtype=np.uint8)
        w, h = sz(0), sz(0)
        dt, p_t = (np.uint16, u16p) if bits == 16 else (np.uint8, u8p)
        planes = [np.zeros(max(bufsize, 1), dt) for _ in range(channels)]
        tail = [C.POINTER(sz), C.POINTER(sz), sz, u8p, sz, C.c_uint8, C.c_int, C.c_uint8]
        if channels == 1:
            fn = self.lib.icer_decompress_image_uint16 if bits == 16 else self.lib.icer_decompress_image_uint8
            fn.argtypes = [p_t] + tail
            rc = fn(planes[0], C.byref(w), C.byref(h), bufsize, buf, len(stream), stages, filt, segments)
        else:
            fn = self.lib.icer_decompress_image_yuv_uint16 if bits == 16 else self.lib.icer_decompress_image_yuv_uint8
            fn.argtypes = [p_t, p_t, p_t] + tail
            rc = fn(planes[0], planes[1], planes[2], C.byref(w), C.byref(h), bufsize, buf, len(stream), stages, filt, segments)
        return rc, w.value, h.value, planes

    def decode_entry(self, b, code):
        """icer_custom_decode_scheme[b][code] -> (code bits, reversed pattern, pattern bits)"""
        arr = (C.c_uint8 * (17 * 32 * 3)).in_dll(self.lib, "icer_custom_decode_scheme")
        k = (b * 32 + code) * 3
        return arr[k], arr[k + 2], arr[k + 1]

    # table taps -------------------------------------------------------------------------
    def custom_code(self, b, prefix):
        a, o, c = C.c_int(), C.c_int(), C.c_int()
        self.lib.ref_tap_custom_code(b, prefix, C.byref(a), C.byref(o), C.byref(c))
        return a.value, o.value, c.value

    def flush_entry(self, b, prefix, nbits):
        f, n = C.c_int(), C.c_int()
        self.lib.ref_tap_flush(b, prefix, nbits, C.byref(f), C.byref(n))
        return f.value, n.value

    def golomb(self, b):
        m, l, i = C.c_int(), C.c_int(), C.c_int()
        self.lib.ref_tap_golomb(b, C.byref(m), C.byref(l), C.byref(i))
        return m.value, l.value, i.value
