"""Real image content through the reference: the pixel planes of the reference repository's own fixtures and the
reference encoder's / decoder's answers on them.  Run in the authoring container (needs /root/reference and
oracle/_ref):  python tests/golden/make_fixture_goldens.py

Writes
  tests/golden/fixture_planes.npz   boat512_gray   (512, 512) uint8 -- boat.512.bmp as the reference's CLI sees it with
                                                   --grayscale: stb_image with one channel requested, (77 r + 150 g + 29 b) >> 8
                                    boatcolor512_rgb (512, 512, 3) uint8 -- boatcolor.512.bmp, the R, G, B the colour flow converts
  tests/golden/fixture_golden.json  per case: the geometry and (rc, size, crc32, sha256[:16]) of the stream of the UNTOUCHED
                                    reference build (oracle/_ref) + its decoder's verdict on that stream
Only pixel data is stored -- no reference source.  The cases are the reference's own invocations:
  cli_gray_<F>     icer_util compress boat.512.bmp X --grayscale -s 3 -g 10 -f <F>      (example/src/icer_util.c:123-246; quota = w*h)
  example_gray     ./compress        4 stages, filter A,  6 segments, quota  30 000      (example/src/example_encode.c)
  example_color    ./compress_color  4 stages, filter A, 10 segments, quota 100 000      (example/src/example_encode_color.c)
  cli_color        icer_util compress boatcolor.512.bmp X -s 4 -g 10 -f A               (colour default: quota = w*h*3)
The cli_gray_* sizes and hashes equal the ones recorded from the reference's CLI binary itself in SURVEY.md 8c
(189 566 B d12559536164296e ...).  The two example programs additionally pass the image through stb_image_resize
(third-party, not restated; not an identity even at the image's own size), so example_* are their PARAMETERS on the
unresized planes, not the bytes those binaries write.
"""
import hashlib
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import Reference  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def load_bmp_rgb(path):
    """24-bit uncompressed BMP -> (h, w, 3) uint8 R, G, B"""
    raw = open(path, "rb").read()
    off = int.from_bytes(raw[10:14], "little")
    w = int.from_bytes(raw[18:22], "little", signed=True)
    h = int.from_bytes(raw[22:26], "little", signed=True)
    assert int.from_bytes(raw[28:30], "little") == 24 and int.from_bytes(raw[30:34], "little") == 0
    stride = (3 * w + 3) // 4 * 4
    rows = np.frombuffer(raw, np.uint8, count=stride * abs(h), offset=off).reshape(abs(h), stride)[:, : 3 * w].reshape(abs(h), w, 3)
    if h > 0:
        rows = rows[::-1]
    return np.ascontiguousarray(rows[..., ::-1])                 # stored B, G, R


def rgb_to_ycbcr(rgb):
    """CRGB2Y / CRGB2Cb / CRGB2Cr of the reference's callers (example/inc/color_util.h:8,27-29), as uint16 planes"""
    r, g, b = (rgb[..., c].astype(np.int64) for c in range(3))
    clip = lambda v: np.clip(v, 0, 255)
    y = clip((19595 * r + 38470 * g + 7471 * b) >> 16)
    cb = clip(((36962 * (b - y)) >> 16) + 128)
    cr = clip(((46727 * (r - y)) >> 16) + 128)
    return [np.ascontiguousarray(p.astype(np.uint16)) for p in (y, cb, cr)]


def cases(gray, rgb):
    out = [(f"cli_gray_{'ABCDEFQ'[f]}", [gray.astype(np.uint16)], 3, f, 10, 512 * 512) for f in range(7)]
    out.append(("example_gray", [gray.astype(np.uint16)], 4, 0, 6, 30000))
    out.append(("example_color", rgb_to_ycbcr(rgb), 4, 0, 10, 100000))
    out.append(("cli_color", rgb_to_ycbcr(rgb), 4, 0, 10, 512 * 512 * 3))
    return out


def main():
    boat = load_bmp_rgb("/root/reference/boat.512.bmp").astype(np.uint32)
    gray = ((77 * boat[..., 0] + 150 * boat[..., 1] + 29 * boat[..., 2]) >> 8).astype(np.uint8)
    rgb = load_bmp_rgb("/root/reference/boatcolor.512.bmp")
    np.savez_compressed(os.path.join(HERE, "fixture_planes.npz"), boat512_gray=gray, boatcolor512_rgb=rgb)
    ref = Reference()
    gold = {}
    for name, planes, st, f, sg, q in cases(gray, rgb):
        h, w = planes[0].shape
        rc, stream, _ = ref.compress(planes, st, f, sg, q)
        drc, dw, dh, dplanes = ref.decompress_raw(stream, len(planes), st, f, sg, bits=16)
        hsh = hashlib.sha256()
        for p in dplanes:
            hsh.update(p.tobytes())
        gold[name] = dict(channels=len(planes), w=w, h=h, stages=st, filt=f, segments=sg, quota=q, rc=rc, size=len(stream),
                          crc32="%08x" % zlib.crc32(stream), sha256_16=hashlib.sha256(stream).hexdigest()[:16],
                          decoded_rc=drc, decoded_sha256_16=hsh.hexdigest()[:16],
                          decoded_max_abs_diff=int(max(np.abs(d.reshape(h, w).astype(np.int64) - p.astype(np.int64)).max() for d, p in zip(dplanes, planes))))
        print(name, gold[name], flush=True)
    with open(os.path.join(HERE, "fixture_golden.json"), "w") as fh:
        json.dump(gold, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
