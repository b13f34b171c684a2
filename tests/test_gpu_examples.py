"""The reference's OWN programs as the test (its CMake target `test_examples`, /root/reference/CMakeLists.txt:45-63):
`compress`, `decompress`, `compress_color`, `decompress_color` and `icer_util` -- example/src/*.c UNMODIFIED, compiled against
the reference's own header -- linked once with the reference build (oracle/_ref/ref_*) and once with the product libraries
libicer_hip.so / libicer_hip_dec.so (oracle/_ref/hip_*) by `make -C oracle examples`, run side by side on boat.512.bmp /
boatcolor.512.bmp, and every file they write compared byte for byte.

The binaries are prebuilt in the authoring container (the GPU box has no /root/reference) and travel with the snapshot;
the two bitmaps are re-made from the committed pixel planes of the reference's fixtures (tests/golden/fixture_planes.npz:
stb_image decodes a 24-bit BMP of those planes to exactly those pixels)."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from tests.test_reference_fixtures import FIXTURE_GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
PROGS = ("compress", "decompress", "compress_color", "decompress_color", "icer_util")


def _have(prefix):
    return all(os.path.exists(os.path.join(REFDIR, f"{prefix}_{p}")) for p in PROGS)


def _write_bmp24(path, rgb):
    h, w, _ = rgb.shape
    stride = (3 * w + 3) // 4 * 4
    rows = np.zeros((h, stride), np.uint8)
    rows[:, : 3 * w] = rgb[::-1, :, ::-1].reshape(h, 3 * w)                  # bottom-up, BGR
    hdr = b"BM" + (54 + stride * h).to_bytes(4, "little") + bytes(4) + (54).to_bytes(4, "little") + (40).to_bytes(4, "little") + \
        w.to_bytes(4, "little") + h.to_bytes(4, "little") + (1).to_bytes(2, "little") + (24).to_bytes(2, "little") + bytes(24)
    with open(path, "wb") as fh:
        fh.write(hdr + rows.tobytes())


def _fixtures(d):
    z = np.load(os.path.join(ROOT, "tests", "golden", "fixture_planes.npz"))
    g = z["boat512_gray"].astype(np.uint8)
    _write_bmp24(os.path.join(d, "boat.512.bmp"), np.repeat(g[:, :, None], 3, axis=2))
    _write_bmp24(os.path.join(d, "boatcolor.512.bmp"), z["boatcolor512_rgb"].astype(np.uint8))


def _run(prefix, prog, args, cwd):
    r = subprocess.run([os.path.join(REFDIR, f"{prefix}_{prog}")] + args, cwd=cwd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (prefix, prog, args, r.stdout[-2000:], r.stderr[-2000:])
    return r.stdout


def _sequence(prefix, d):
    """the command sequence of the reference's `test_examples` target; returns {file name: bytes} of everything written"""
    files = {}
    _run(prefix, "compress", [], d)
    files["compressed.bin (gray)"] = open(os.path.join(d, "compressed.bin"), "rb").read()
    _run(prefix, "decompress", [], d)
    files["decompress.bmp (gray)"] = open(os.path.join(d, "decompress.bmp"), "rb").read()
    _run(prefix, "compress_color", [], d)
    files["compressed.bin (color)"] = open(os.path.join(d, "compressed.bin"), "rb").read()
    _run(prefix, "decompress_color", [], d)
    files["decompress.bmp (color)"] = open(os.path.join(d, "decompress.bmp"), "rb").read()
    _run(prefix, "icer_util", ["compress", "boat.512.bmp", "test_compressed.bin", "--grayscale"], d)
    _run(prefix, "icer_util", ["decompress", "test_compressed.bin", "test_decompressed.bmp", "--grayscale"], d)
    _run(prefix, "icer_util", ["compress", "boatcolor.512.bmp", "test_compressed_color.bin", "--color"], d)
    _run(prefix, "icer_util", ["decompress", "test_compressed_color.bin", "test_decompressed_color.bmp", "--color"], d)
    # (beyond the CMake target: the CLI known answers of SURVEY 8c, one filter with the W3 quirk, and a byte quota)
    _run(prefix, "icer_util", ["compress", "boat.512.bmp", "kat_c.bin", "--grayscale", "-s", "3", "-g", "10", "-f", "C"], d)
    _run(prefix, "icer_util", ["compress", "boatcolor.512.bmp", "quota.bin", "--color", "-t", "50000", "-s", "5", "-g", "7"], d)
    for f in ("test_compressed.bin", "test_decompressed.bmp", "test_compressed_color.bin", "test_decompressed_color.bmp", "kat_c.bin", "quota.bin"):
        files[f] = open(os.path.join(d, f), "rb").read()
    return files


@pytest.mark.skipif(not _have("ref"), reason="oracle/_ref/ref_* not built (make -C oracle examples; needs /root/reference)")
def test_reference_programs_on_the_rebuilt_bitmaps_give_the_fixture_goldens(tmp_path):
    """CPU: the bitmaps re-made from the committed planes are the fixtures as far as the reference's programs can tell --
    the command-line tool's streams carry the digests recorded from the reference build on the original files (tests/golden/fixture_golden.json)"""
    _fixtures(str(tmp_path))
    for args, name in ((["boat.512.bmp", "k.bin", "--grayscale", "-s", "3", "-g", "10", "-f", "C"], "cli_gray_C"),
                       (["boat.512.bmp", "k.bin", "--grayscale", "-s", "3", "-g", "10", "-f", "Q"], "cli_gray_Q"),
                       (["boatcolor.512.bmp", "k.bin", "--color", "-s", "4", "-g", "10"], "cli_color")):
        _run("ref", "icer_util", ["compress"] + args, str(tmp_path))
        blob = open(tmp_path / "k.bin", "rb").read()
        g = FIXTURE_GOLDEN[name]
        assert (len(blob), hashlib.sha256(blob).hexdigest()[:16]) == (g["size"], g["sha256_16"]), name
    # (the two example programs resample the image through stb_image_resize before they encode it, so their streams are
    # not the fixture goldens of the un-resampled planes; they are compared between the two builds on the GPU box)


@pytest.mark.gpu
def test_reference_programs_linked_with_the_product_libraries(tmp_path):
    """GPU: the same unmodified programs linked with libicer_hip.so + libicer_hip_dec.so write the same files as the ones
    linked with the reference -- every .bin and every decoded .bmp of the `test_examples` sequence"""
    assert _have("hip") and _have("ref"), "oracle/_ref/hip_* / ref_* are missing: run `make -C oracle examples` in the authoring container"
    a, b = tmp_path / "ref", tmp_path / "hip"
    a.mkdir(); b.mkdir()
    _fixtures(str(a)); _fixtures(str(b))
    want, got = _sequence("ref", str(a)), _sequence("hip", str(b))
    assert sorted(want) == sorted(got)
    for name in want:
        assert len(want[name]) > 0 and got[name] == want[name], name
    g = FIXTURE_GOLDEN["cli_gray_C"]
    assert (len(got["kat_c.bin"]), hashlib.sha256(got["kat_c.bin"]).hexdigest()[:16]) == (g["size"], g["sha256_16"])
