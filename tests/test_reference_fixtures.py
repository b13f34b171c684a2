"""Known answers recorded from the reference's own CLI on its own fixture:
`icer_util compress boat.512.bmp X --grayscale -s 3 -g 10 -f {A..Q}` (quota = w*h = 262 144).  The image is read from
/root/reference (never copied into this repository); the test is skipped where that directory does not exist."""
import hashlib
import os

import numpy as np
import pytest

BOAT = "/root/reference/boat.512.bmp"
KAT = {0: (189566, "d12559536164296e"), 1: (188996, "851229b79048d200"), 2: (190218, "b815bf03c3ff04bc"),
       3: (188910, "0d1c9032698d7d28"), 4: (190578, "cc59339cfcbe07d1"), 5: (192612, "2d7fd5067babdf76"),
       6: (190383, "cb63818012be1eec")}


def load_bmp_gray(path):
    """24-bit uncompressed BMP -> (h, w) uint16.  The reference loads it through stb_image with one channel
    requested, i.e. (77 r + 150 g + 29 b) >> 8."""
    raw = open(path, "rb").read()
    off = int.from_bytes(raw[10:14], "little")
    w = int.from_bytes(raw[18:22], "little", signed=True)
    h = int.from_bytes(raw[22:26], "little", signed=True)
    bpp = int.from_bytes(raw[28:30], "little")
    assert bpp == 24 and int.from_bytes(raw[30:34], "little") == 0
    stride = (3 * w + 3) // 4 * 4
    rows = np.frombuffer(raw, np.uint8, count=stride * abs(h), offset=off).reshape(abs(h), stride)[:, : 3 * w].reshape(abs(h), w, 3)
    if h > 0:
        rows = rows[::-1]                                    # bottom-up storage
    b, g, r = (rows[..., c].astype(np.uint32) for c in range(3))
    return ((77 * r + 150 * g + 29 * b) >> 8).astype(np.uint16)


@pytest.mark.skipif(not os.path.exists(BOAT), reason="reference fixture not mounted")
@pytest.mark.parametrize("filt", sorted(KAT))
def test_boat_512_cli_known_answers(oracle, filt):
    img = load_bmp_gray(BOAT)
    assert img.shape == (512, 512)
    rc, stream, _ = oracle.compress([img], 3, filt, 10, 512 * 512)
    assert rc == 0
    assert (len(stream), hashlib.sha256(stream).hexdigest()[:16]) == KAT[filt]


# ---- the fixtures' pixel planes, committed (tests/golden/fixture_planes.npz, made by tests/golden/make_fixture_goldens.py):
# the same known answers -- and the reference's parameters of its two example programs and of the colour CLI flow -- wherever
# the tests run, with the reference build's encoder AND decoder verdicts (tests/golden/fixture_golden.json)
import json  # noqa: E402

GOLD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLD_DIR, "fixture_golden.json")) as _fh:
    FIXTURE_GOLDEN = json.load(_fh)


def fixture_planes(name):
    """uint16 planes of a case of fixture_golden.json"""
    z = np.load(os.path.join(GOLD_DIR, "fixture_planes.npz"))
    if FIXTURE_GOLDEN[name]["channels"] == 1:
        return [z["boat512_gray"].astype(np.uint16)]
    rgb = z["boatcolor512_rgb"]
    r, g, b = (rgb[..., c].astype(np.int64) for c in range(3))
    clip = lambda v: np.clip(v, 0, 255)
    y = clip((19595 * r + 38470 * g + 7471 * b) >> 16)                       # CRGB2Y/Cb/Cr, example/inc/color_util.h:27-29
    return [np.ascontiguousarray(p.astype(np.uint16)) for p in (y, clip(((36962 * (b - y)) >> 16) + 128), clip(((46727 * (r - y)) >> 16) + 128))]


def test_committed_gray_plane_is_the_fixture():
    if not os.path.exists(BOAT):
        pytest.skip("reference fixture not mounted")
    assert np.array_equal(load_bmp_gray(BOAT), fixture_planes("cli_gray_A")[0])


@pytest.mark.parametrize("name", sorted(FIXTURE_GOLDEN))
def test_fixture_known_answers_oracle(oracle, name):
    g = FIXTURE_GOLDEN[name]
    planes = fixture_planes(name)
    rc, stream, _ = oracle.compress(planes, g["stages"], g["filt"], g["segments"], g["quota"])
    assert (rc, len(stream), hashlib.sha256(stream).hexdigest()[:16]) == (g["rc"], g["size"], g["sha256_16"])
    if name.startswith("cli_gray"):
        assert (g["size"], g["sha256_16"]) == KAT[g["filt"]]                 # the reference CLI binary's own output
    drc, w, h, back = oracle.decompress(stream, g["channels"], g["stages"], g["filt"], g["segments"])
    hsh = hashlib.sha256()
    for p in back:
        hsh.update(p.tobytes())
    assert (drc, w, h, hsh.hexdigest()[:16]) == (g["decoded_rc"], g["w"], g["h"], g["decoded_sha256_16"])
