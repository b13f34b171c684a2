"""Host-side mirror of the reference interface for the encode path, over the C ABI of
libicer_hip.so (include/icer_hip.h).

The reference is a C library (lib_icer); its own callers do
    icer_init(); icer_init_output_struct(&out, buf, len, quota);
    rc = icer_compress_image_uint16(img, w, h, stages, filt, segments, &out);
    fwrite(out.rearrange_start, out.size_used)
(example/src/example_encode.c:36-77, example/src/icer_util.c:186-227).  The functions below keep the
same names, argument order, in-place side effect on the image and return codes, with numpy arrays
standing in for the raw pointers.  `Encoder` wraps the batched / device-resident extension
(icerx_*), which is what bench.py times.

There is no CPU fallback: importing works anywhere, but every compress call needs libicer_hip.so
and a HIP device and fails loudly otherwise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
# (ICER_HIP_LIB: another build of the same library, e.g. one made with different tuning flags -- measurements only)
LIB_PATH = os.environ.get("ICER_HIP_LIB") or os.path.join(PKG, "libicer_hip.so")

# enum icer_status (lib_icer/inc/icer.h:92-105)
ICER_RESULT_OK = 0
ICER_INTEGER_OVERFLOW = -1
ICER_OUTPUT_BUF_TOO_SMALL = -2
ICER_TOO_MANY_SEGMENTS = -3
ICER_TOO_MANY_STAGES = -4
ICER_BYTE_QUOTA_EXCEEDED = -5
ICER_BITPLANE_OUT_OF_RANGE = -6
ICER_PACKET_COUNT_EXCEEDED = -9
ICER_FATAL_ERROR = -10
ICER_INVALID_INPUT = -11
# enum icer_filter_types (icer.h:107-115)
ICER_FILTER_A, ICER_FILTER_B, ICER_FILTER_C, ICER_FILTER_D, ICER_FILTER_E, ICER_FILTER_F, ICER_FILTER_Q = range(7)
ICERX_NUM_STAGES = 4
STAGE_NAMES = ("dwt", "ll_mean+sign_magnitude", "code_units", "scan+gather")


class icer_output_data_buf_typedef(C.Structure):       # icer.h:307-312
    _fields_ = [("size_used", C.c_size_t), ("size_allocated", C.c_size_t),
                ("data_start", C.c_void_p), ("rearrange_start", C.c_void_p)]


class IcerHipError(RuntimeError):
    pass


_lib = None


def load_library() -> C.CDLL:
    """dlopen libicer_hip.so (RTLD_LOCAL: it exports the same icer_* names as the reference)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IcerHipError(f"{LIB_PATH} is missing: build it with `python -m icer_compression_amd.build` "
                           "(hipcc, gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH, mode=os.RTLD_LOCAL)
    u16 = C.c_void_p
    L.icer_init.restype = C.c_int
    L.icer_init_output_struct.argtypes = [C.POINTER(icer_output_data_buf_typedef), C.c_void_p, C.c_size_t, C.c_size_t]
    L.icer_compress_image_uint16.argtypes = [u16, C.c_size_t, C.c_size_t, C.c_uint8, C.c_int, C.c_uint8,
                                             C.POINTER(icer_output_data_buf_typedef)]
    L.icer_compress_image_yuv_uint16.argtypes = [u16, u16, u16, C.c_size_t, C.c_size_t, C.c_uint8, C.c_int, C.c_uint8,
                                                 C.POINTER(icer_output_data_buf_typedef)]
    L.icer_compress_image_uint8.argtypes = L.icer_compress_image_uint16.argtypes
    L.icer_compress_image_yuv_uint8.argtypes = L.icer_compress_image_yuv_uint16.argtypes
    L.icerx_encoder_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int]
    L.icerx_encoder_create_ex.argtypes = L.icerx_encoder_create.argtypes + [C.c_int]
    L.icerx_encoder_destroy.argtypes = [C.c_void_p]
    L.icerx_encoder_destroy.restype = None
    L.icerx_encode_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
    L.icerx_encode_device_async.argtypes = L.icerx_encode_device.argtypes
    L.icerx_encoder_wait.argtypes = [C.c_void_p]
    L.icerx_compress_batch_uint16.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                              C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    L.icerx_compress_batch_uint16_devices.argtypes = L.icerx_compress_batch_uint16.argtypes[:-1] + [C.POINTER(C.c_int), C.c_int]
    L.icerx_batch_release.restype = None
    L.icerx_encode_device_u8.argtypes = L.icerx_encode_device.argtypes
    L.icerx_encode_device_rgb8.argtypes = L.icerx_encode_device.argtypes
    L.icerx_encode_device_s8.argtypes = L.icerx_encode_device.argtypes
    L.icerx_encode_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.icerx_get_coefficients.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.icerx_timing_enable.argtypes = [C.c_void_p, C.c_int]
    L.icerx_timing_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
    L.icerx_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.icerx_encoder_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.icerx_encoder_routing.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.icerx_encoder_launch_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.icerx_encoder_parts.argtypes = [C.c_void_p]
    L.icerx_process_stats.argtypes = [C.POINTER(C.c_uint64)]
    L.icerx_pin_host.argtypes = [C.c_void_p, C.c_size_t]
    L.icerx_unpin_host.argtypes = [C.c_void_p]
    L.icerx_last_error.restype = C.c_char_p
    _lib = L
    return L


# ---- lib_icer-shaped entry points --------------------------------------------------------------
def icer_init() -> int:
    return load_library().icer_init()


def icer_init_output_struct(out: icer_output_data_buf_typedef, data: np.ndarray, buf_len: int, byte_quota: int) -> int:
    return load_library().icer_init_output_struct(C.byref(out), data.ctypes.data, buf_len, byte_quota)


def _check_plane(a: np.ndarray, w: int, h: int, dtype=np.uint16) -> None:
    if a.dtype != dtype or not a.flags["C_CONTIGUOUS"] or a.size != w * h:
        raise ValueError(f"image planes must be C-contiguous {np.dtype(dtype).name} arrays of w*h elements")


def icer_compress_image_uint16(image: np.ndarray, image_w: int, image_h: int, stages: int, filt: int, segments: int,
                               output_data: icer_output_data_buf_typedef) -> int:
    _check_plane(image, image_w, image_h)
    return load_library().icer_compress_image_uint16(image.ctypes.data, image_w, image_h, stages, filt, segments,
                                                     C.byref(output_data))


def icer_compress_image_yuv_uint16(y: np.ndarray, u: np.ndarray, v: np.ndarray, image_w: int, image_h: int, stages: int,
                                   filt: int, segments: int, output_data: icer_output_data_buf_typedef) -> int:
    for p in (y, u, v):
        _check_plane(p, image_w, image_h)
    return load_library().icer_compress_image_yuv_uint16(y.ctypes.data, u.ctypes.data, v.ctypes.data, image_w, image_h,
                                                         stages, filt, segments, C.byref(output_data))


def icer_compress_image_uint8(image: np.ndarray, image_w: int, image_h: int, stages: int, filt: int, segments: int,
                              output_data: icer_output_data_buf_typedef) -> int:
    _check_plane(image, image_w, image_h, np.uint8)
    return load_library().icer_compress_image_uint8(image.ctypes.data, image_w, image_h, stages, filt, segments,
                                                    C.byref(output_data))


def icer_compress_image_yuv_uint8(y: np.ndarray, u: np.ndarray, v: np.ndarray, image_w: int, image_h: int, stages: int,
                                  filt: int, segments: int, output_data: icer_output_data_buf_typedef) -> int:
    for p in (y, u, v):
        _check_plane(p, image_w, image_h, np.uint8)
    return load_library().icer_compress_image_yuv_uint8(y.ctypes.data, u.ctypes.data, v.ctypes.data, image_w, image_h,
                                                        stages, filt, segments, C.byref(output_data))


def compress_u8(planes, stages: int, filt: int, segments: int, byte_quota: int):
    """As compress(), through the uint8 twins (planes: (h, w) uint8 arrays, int8 storage)."""
    icer_init()
    work = [np.ascontiguousarray(p, dtype=np.uint8).copy() for p in planes]
    h, w = work[0].shape
    buf = np.zeros(2 * byte_quota + 64, dtype=np.uint8)
    out = icer_output_data_buf_typedef()
    rc = icer_init_output_struct(out, buf, buf.size, byte_quota)
    if rc != ICER_RESULT_OK:
        return rc, b"", work
    if len(work) == 1:
        rc = icer_compress_image_uint8(work[0], w, h, stages, filt, segments, out)
    else:
        rc = icer_compress_image_yuv_uint8(work[0], work[1], work[2], w, h, stages, filt, segments, out)
    return rc, bytes(buf[byte_quota: byte_quota + out.size_used]), work


def compress(planes, stages: int, filt: int, segments: int, byte_quota: int):
    """Convenience wrapper used by the tests: same call sequence as the reference's CLI
    (example/src/icer_util.c:186-227).  Returns (rc, stream bytes, planes as left by the call)."""
    icer_init()
    work = [np.ascontiguousarray(p, dtype=np.uint16).copy() for p in planes]
    h, w = work[0].shape
    buf = np.zeros(2 * byte_quota + 64, dtype=np.uint8)
    out = icer_output_data_buf_typedef()
    rc = icer_init_output_struct(out, buf, buf.size, byte_quota)
    if rc != ICER_RESULT_OK:
        return rc, b"", work
    if len(work) == 1:
        rc = icer_compress_image_uint16(work[0], w, h, stages, filt, segments, out)
    else:
        rc = icer_compress_image_yuv_uint16(work[0], work[1], work[2], w, h, stages, filt, segments, out)
    return rc, bytes(buf[byte_quota: byte_quota + out.size_used]), work


# ---- batched / device-resident extension ----------------------------------------------------------
class Encoder:
    """icerx_encoder: frames of one geometry, many per call, buffers resident on one GPU."""

    def __init__(self, w: int, h: int, channels: int = 1, stages: int = 4, filt: int = ICER_FILTER_A, segments: int = 10,
                 max_frames: int = 1, device: int = 0, sample_bits: int = 16):
        self.lib = load_library()
        self.w, self.h, self.channels, self.max_frames, self.device = w, h, channels, max_frames, device
        self.sample_bits = sample_bits
        self.handle = C.c_void_p()
        rc = self.lib.icerx_encoder_create_ex(C.byref(self.handle), device, w, h, channels, stages, filt, segments, max_frames,
                                              sample_bits)
        self.create_rc = rc
        if rc != 0:
            self.handle = C.c_void_p()
            if rc == ICER_FATAL_ERROR:
                raise IcerHipError(f"icerx_encoder_create failed: {self.lib.icerx_last_error().decode()}")

    def close(self):
        if getattr(self, "handle", None) and self.handle.value:
            self.lib.icerx_encoder_destroy(self.handle)
            self.handle = C.c_void_p()

    __del__ = close

    def encode_device_ptrs(self, d_frames: int, n_frames: int, byte_quota: int, d_out: int, out_stride: int, d_sizes: int,
                           d_rcs: int, stream: int = 0) -> None:
        rc = self.lib.icerx_encode_device(self.handle, d_frames, n_frames, byte_quota, d_out, out_stride, d_sizes, d_rcs,
                                          stream)
        if rc != 0:
            raise IcerHipError(f"icerx_encode_device rc={rc}: {self.lib.icerx_last_error().decode()}")

    def encode_device_async_ptrs(self, d_frames: int, n_frames: int, byte_quota: int, d_out: int, out_stride: int, d_sizes: int,
                                 d_rcs: int, stream: int = 0) -> None:
        """first half of encode_device_ptrs: returns once everything is enqueued; wait() completes it"""
        rc = self.lib.icerx_encode_device_async(self.handle, d_frames, n_frames, byte_quota, d_out, out_stride, d_sizes, d_rcs, stream)
        if rc != 0:
            raise IcerHipError(f"icerx_encode_device_async rc={rc}: {self.lib.icerx_last_error().decode()}")

    def wait(self) -> None:
        rc = self.lib.icerx_encoder_wait(self.handle)
        if rc != 0:
            raise IcerHipError(f"icerx_encoder_wait rc={rc}: {self.lib.icerx_last_error().decode()}")

    def encode_torch(self, frames, byte_quota: int, out, sizes, rcs) -> None:
        """frames: cuda int16/uint16 tensor (n, channels, h, w) or (n, h, w); out: cuda uint8 (n, stride);
        sizes: cuda int64 (n,); rcs: cuda int32 (n,).  Runs on torch's current stream."""
        import torch
        n = frames.shape[0]
        st = torch.cuda.current_stream(frames.device).cuda_stream
        self.encode_device_ptrs(frames.data_ptr(), n, byte_quota, out.data_ptr(), out.stride(0), sizes.data_ptr(),
                                rcs.data_ptr(), st)

    def encode_torch_s8(self, planes, byte_quota: int):
        """uint8 twins: planes = cuda uint8 tensor (n, h, w) or (n, channels, h, w), int8 storage; the encoder must have
        been created with sample_bits=8.  Returns [(rc, stream bytes)] per frame."""
        import torch
        n = planes.shape[0]
        stride = byte_quota + 64
        out = torch.empty((n, stride), dtype=torch.uint8, device=planes.device)
        sizes = torch.empty(n, dtype=torch.int64, device=planes.device)
        rcs = torch.empty(n, dtype=torch.int32, device=planes.device)
        st = torch.cuda.current_stream(planes.device).cuda_stream
        rc = self.lib.icerx_encode_device_s8(self.handle, planes.data_ptr(), n, byte_quota, out.data_ptr(), stride,
                                             sizes.data_ptr(), rcs.data_ptr(), st)
        if rc != 0:
            raise IcerHipError(f"icerx_encode_device_s8 rc={rc}: {self.lib.icerx_last_error().decode()}")
        sz, rr, host = sizes.cpu().numpy(), rcs.cpu().numpy(), out.cpu().numpy()
        return [(int(rr[k]), bytes(host[k, : int(sz[k])])) for k in range(n)]

    def encode_torch_frontend(self, raw, byte_quota: int, out, sizes, rcs) -> None:
        """raw: cuda uint8 tensor, (n, h, w) gray for a 1-channel encoder or (n, h, w, 3) packed RGB for a
        3-channel one; converted on the device (icerx_encode_device_u8 / _rgb8)."""
        import torch
        fn = self.lib.icerx_encode_device_u8 if self.channels == 1 else self.lib.icerx_encode_device_rgb8
        st = torch.cuda.current_stream(raw.device).cuda_stream
        rc = fn(self.handle, raw.data_ptr(), raw.shape[0], byte_quota, out.data_ptr(), out.stride(0), sizes.data_ptr(),
                rcs.data_ptr(), st)
        if rc != 0:
            raise IcerHipError(f"front-end encode rc={rc}: {self.lib.icerx_last_error().decode()}")

    def encode_host(self, frames: np.ndarray, byte_quota: int):
        """frames: uint16 array (n, channels, h, w) or (n, h, w).  Returns list of (rc, stream bytes)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint16)
        n = frames.shape[0]
        stride = byte_quota
        out = np.zeros((n, stride), dtype=np.uint8)
        sizes = np.zeros(n, dtype=np.uint64)
        rcs = np.zeros(n, dtype=np.int32)
        rc = self.lib.icerx_encode_host(self.handle, frames.ctypes.data, n, byte_quota, out.ctypes.data, stride,
                                        sizes.ctypes.data, rcs.ctypes.data)
        if rc != 0:
            raise IcerHipError(f"icerx_encode_host rc={rc}: {self.lib.icerx_last_error().decode()}")
        return [(int(rcs[i]), bytes(out[i, : int(sizes[i])])) for i in range(n)]

    def encode_host_into(self, frames: np.ndarray, byte_quota: int, out: np.ndarray, sizes: np.ndarray, rcs: np.ndarray) -> None:
        """the same into caller-owned arrays (out: uint8 (n, stride), sizes: uint64 (n,), rcs: int32 (n,)) -- nothing is
        allocated or copied on the Python side, so this is what a C caller of icerx_encode_host sees"""
        n = frames.shape[0]
        rc = self.lib.icerx_encode_host(self.handle, frames.ctypes.data, n, byte_quota, out.ctypes.data, out.shape[1],
                                        sizes.ctypes.data, rcs.ctypes.data)
        if rc != 0:
            raise IcerHipError(f"icerx_encode_host rc={rc}: {self.lib.icerx_last_error().decode()}")

    def coefficients(self, frame: int = 0, channel: int = 0) -> np.ndarray:
        dst = np.zeros((self.h, self.w), dtype=np.uint16)
        rc = self.lib.icerx_get_coefficients(self.handle, frame, channel, dst.ctypes.data)
        if rc != 0:
            raise IcerHipError(f"icerx_get_coefficients rc={rc}")
        return dst

    def timing_enable(self, on: bool = True) -> None:
        self.lib.icerx_timing_enable(self.handle, 1 if on else 0)

    def timing_read(self, reset: bool = True):
        ms = (C.c_double * ICERX_NUM_STAGES)()
        calls = C.c_uint64(0)
        rc = self.lib.icerx_timing_read(self.handle, ms, C.byref(calls), 1 if reset else 0)
        if rc != 0:
            raise IcerHipError(f"icerx_timing_read rc={rc}")
        return {n: ms[i] for i, n in enumerate(STAGE_NAMES)}, int(calls.value)

    def stats(self):
        out = (C.c_uint64 * 4)()
        self.lib.icerx_encoder_stats(self.handle, out)
        return {"unit_timeouts": out[0], "fallback_batches": out[1], "slot_retries": out[2], "coder_mode": out[3]}

    def routing(self):
        """coding units that went to the small workgroup coder beside the pipeline kernel, and the calls that routed"""
        out = (C.c_uint64 * 2)()
        self.lib.icerx_encoder_routing(self.handle, out)
        return {"routed_units": out[0], "routed_calls": out[1]}

    def launch_info(self):
        """shape of the last launch (icerx_encoder_launch_info)"""
        out = (C.c_uint32 * 4)()
        self.lib.icerx_encoder_launch_info(self.handle, out)
        return {"split": bool(out[0]), "sub_range_workgroups": int(out[1]), "pipeline_waves": int(out[2]), "window_coder_beside": bool(out[3])}

    def parts(self) -> int:
        """parts the last call was enqueued in (icerx_encoder_parts)"""
        return int(self.lib.icerx_encoder_parts(self.handle))

    def info(self):
        u, b, s = C.c_uint32(), C.c_uint32(), C.c_uint64()
        self.lib.icerx_info(self.handle, C.byref(u), C.byref(b), C.byref(s))
        return {"units_per_frame": u.value, "slot_bits_per_pixel": b.value, "slot_bytes_per_frame": s.value}


def compress_batch(frames: np.ndarray, stages: int, filt: int, segments: int, byte_quota: int, out: np.ndarray, sizes: np.ndarray,
                   rcs: np.ndarray, devices=None) -> int:
    """icerx_compress_batch_uint16[_devices]: frames uint16 (n, h, w) or (n, channels, h, w) in host memory (page-locked for
    DMA: pin_host), out uint8 (n, stride), sizes uint64 (n,), rcs int32 (n,); devices = list of HIP devices, None = all.
    Returns the call's return code (0 = every frame coded; per-frame codes in rcs)."""
    L = load_library()
    n = frames.shape[0]
    ch = 1 if frames.ndim == 3 else frames.shape[1]
    h, w = frames.shape[-2:]
    assert frames.dtype == np.uint16 and frames.flags["C_CONTIGUOUS"] and out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"]
    if devices is None:
        return L.icerx_compress_batch_uint16(frames.ctypes.data, n, w, h, ch, stages, filt, segments, byte_quota, out.ctypes.data,
                                             out.shape[1], sizes.ctypes.data, rcs.ctypes.data, 0)
    dv = (C.c_int * len(devices))(*devices)
    return L.icerx_compress_batch_uint16_devices(frames.ctypes.data, n, w, h, ch, stages, filt, segments, byte_quota, out.ctypes.data,
                                                 out.shape[1], sizes.ctypes.data, rcs.ctypes.data, dv, len(devices))


def process_stats():
    """unit time-outs / batches re-coded by the barrier-only coder / slot re-runs, summed over every encoder of the process"""
    out = (C.c_uint64 * 4)()
    load_library().icerx_process_stats(out)
    return {"unit_timeouts": out[0], "fallback_batches": out[1], "slot_retries": out[2]}


def pin_host(arr) -> bool:
    """page-lock a numpy array (icerx_pin_host); returns False if the runtime refuses"""
    return load_library().icerx_pin_host(arr.ctypes.data, arr.nbytes) == 0


def unpin_host(arr) -> None:
    load_library().icerx_unpin_host(arr.ctypes.data)
