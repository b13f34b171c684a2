"""Regenerates tests/golden/golden.json from the UNTOUCHED reference (oracle/_ref/libicer_ref.so, built by
oracle/Makefile from /root/reference).  Run in the authoring container:  python tests/golden/make_golden.py

Each entry pins (return code, stream length, zlib CRC-32, sha256[:16]) of the reference encoder's output for
one configuration of BASELINE.json / SURVEY.md 8(d), on inputs from icer_compression_amd.synth -- and, for frames
up to 4096 x 4096, the reference DECODER's verdict on that stream (return code, sha256[:16] of the decoded planes),
which pins oracle/icer_oracle.c's decoder restatement.
The reference's own repository holds no golden vectors, so these are the pinned vectors.
"""
import hashlib
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icer_compression_amd import synth  # noqa: E402
from oracle.binding import Reference  # noqa: E402

CASES = [
    # name, kind, w, h, stages, filt, segments, quota, seed, mode
    ("kat_512_m1", "gray", 512, 512, 3, 0, 10, 2 * 512 * 512, 12345, 1),
    ("kat_512_m0", "gray", 512, 512, 3, 0, 10, 2 * 512 * 512, 12345, 0),
    ("kat_512_quota30000", "gray", 512, 512, 4, 0, 6, 30000, 12345, 1),
    ("kat_odd_517x389_filtC", "gray", 517, 389, 4, 2, 7, 2 * 517 * 389, 777, 1),
    ("kat_1024_16seg", "gray", 1024, 1024, 4, 0, 16, 2 * 1024 * 1024, 12345, 1),
    ("kat_color_512_quota", "yuv", 512, 512, 4, 0, 10, 100000, 12345, 1),
    ("C2_4096_gray_5st_10seg", "gray", 4096, 4096, 5, 0, 10, 2 * 4096 * 4096, 12345, 1),
    ("C2_4096_gray_noise", "gray", 4096, 4096, 5, 0, 10, 2 * 4096 * 4096, 12345, 0),
    ("C3_4096_yuv_quota70000", "yuv", 4096, 4096, 5, 0, 10, 70000, 12345, 1),
    ("C4_2048_frame0", "gray", 2048, 2048, 4, 0, 16, 2 * 2048 * 2048, 12345, 1),
    ("C4_2048_frame1", "gray", 2048, 2048, 4, 0, 16, 2 * 2048 * 2048, 12346, 1),
    ("C5_8192_frame0", "gray", 8192, 8192, 6, 0, 32, 2 * 8192 * 8192, 12345, 1),
    # uint8 twins: the same synthetic frames >> 2, i.e. 6-bit data that stays inside int8
    ("u8_512_gray", "gray8", 512, 512, 3, 0, 10, 2 * 512 * 512, 12345, 1),
    ("u8_517x389_filtB_quota", "gray8", 517, 389, 4, 1, 7, 20000, 777, 1),
    ("u8_2048_gray_4st_16seg", "gray8", 2048, 2048, 4, 0, 16, 2 * 2048 * 2048, 12345, 1),
    ("u8_512_yuv_4st", "yuv8", 512, 512, 4, 0, 10, 2 * 512 * 512, 12345, 1),
    ("u8_512_yuv_quota", "yuv8", 512, 512, 3, 0, 8, 60000, 12345, 1),
    ("u8_256_yuv_5st_packet_table", "yuv8", 256, 256, 5, 0, 4, 1 << 20, 12345, 1),
    ("u8_512_gray_full_range_overflow", "gray8full", 512, 512, 3, 0, 10, 2 * 512 * 512, 12345, 0),
    # the PRODUCTION sub-range split (a lone 4096 x 4096 plane: level-1 units of 6 554 chunks are cut at 3 072 chunks,
    # csrc/api.hip split_chunks) through other tables than filter A / 8-bit content: the filter with the W3 quirk, 12-bit
    # content (more than 9 planes of magnitude: category 3, large coefficients) through filter B, and the uint8 twin
    # (7 planes, int8 storage, its own packet table)
    ("split_4096_filtC", "gray", 4096, 4096, 5, 2, 10, 2 * 4096 * 4096, 12345, 1),
    ("split_4096_12bit_filtB", "gray12", 4096, 4096, 5, 1, 10, 2 * 4096 * 4096, 12345, 1),
    ("split_4096_u8_gray", "gray8", 4096, 4096, 5, 0, 10, 2 * 4096 * 4096, 12345, 1),
]


def planes_of(kind, w, h, seed, mode):
    if kind == "gray":
        return [synth.gray_frame(w, h, seed, mode)]
    if kind == "yuv":
        return list(synth.color_frame_yuv(w, h, seed))
    if kind == "gray12":
        return [synth.gray_frame_12bit(w, h, seed, mode)]
    if kind == "gray8":
        return [synth.gray_frame_u8(w, h, seed, mode)]
    if kind == "gray8full":
        return [synth.gray_frame(w, h, seed, mode).astype("uint8")]
    return list(synth.color_frame_yuv_u8(w, h, seed))


def main():
    ref = Reference()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json")
    out = {}
    if "--only-missing" in sys.argv and os.path.exists(path):      # (the 8192 x 8192 entry takes minutes)
        with open(path) as fh:
            out = json.load(fh)
    for name, kind, w, h, st, f, sg, q, seed, mode in CASES:
        if name in out and ("decoded_rc" in out[name] or w * h > 4096 * 4096 or out[name]["size"] == 0):
            continue
        planes = planes_of(kind, w, h, seed, mode)
        t = time.time()
        rc, stream, _ = (ref.compress_u8 if kind.endswith(("8", "8full")) else ref.compress)(planes, st, f, sg, q)
        dt = time.time() - t
        dec = {}
        if stream and w * h <= 4096 * 4096:
            u8 = kind.endswith(("8", "8full"))
            drc, dw, dh, dplanes = ref.decompress_raw(stream, len(planes), st, f, sg, bits=8 if u8 else 16)
            hsh = hashlib.sha256()
            for p in dplanes:
                hsh.update(p.tobytes())
            dec = dict(decoded_rc=drc, decoded_w=dw, decoded_h=dh, decoded_sha256_16=hsh.hexdigest()[:16],
                       decoded_is_input=bool(all((d.reshape(h, w) == p).all() for d, p in zip(dplanes, planes))))
        out[name] = dict(**dec, kind=kind, w=w, h=h, stages=st, filt=f, segments=sg, quota=q, seed=seed, mode=mode, rc=rc,
                         size=len(stream), crc32="%08x" % zlib.crc32(stream), sha256_16=hashlib.sha256(stream).hexdigest()[:16],
                         ref_seconds=round(dt, 3))
        print(name, out[name], flush=True)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
