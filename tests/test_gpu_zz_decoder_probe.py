"""The plain-C decode example (tests/c_abi/dropin_decode_example.c) against libicer_hip_dec.so, every decode kernel, on the
configurations of the decoder's first GPU run (profiles/archive/r01_decoder_first_gpu_run.log): 16-bit gray frames, lossless streams.
The example runs in a child process, so a fault in the decoder cannot take the test session with it.  (The wide decoder suite
-- YUV, uint8, damaged streams, goldens, the batch object -- is tests/test_gpu_decoder.py, also in the default `-m gpu` run.)
"""
import os
import subprocess

import numpy as np
import pytest

from icer_compression_amd import synth
from oracle.binding import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    from icer_compression_amd.build import build_decoder_library
    out = str(tmp_path_factory.mktemp("dec") / "dropin_decode_example")
    libdir = os.path.dirname(build_decoder_library())
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "dropin_decode_example.c"),
                           "-L", libdir, "-licer_hip_dec", "-Wl,-rpath," + libdir, "-o", out])
    return out


@pytest.mark.parametrize("w,h,stages,segments,seed", [(320, 200, 3, 7, 5), (1024, 1024, 4, 16, 12345)])
@pytest.mark.parametrize("kernel", ["0", "1", "2"], ids=["thread-per-chain", "wave-per-chain", "wave-per-plane"])
def test_decoder_on_the_gpu(exe, tmp_path, w, h, stages, segments, seed, kernel):
    orc = Oracle()
    img = synth.gray_frame(w, h, seed, 1)
    rc, stream, _ = orc.compress([img], stages, 0, segments, 2 * w * h)
    assert rc == 0
    want = orc.decompress(stream, 1, stages, 0, segments)
    (tmp_path / "in.bin").write_bytes(stream)
    env = dict(os.environ, ICER_DEC_WAVE=kernel)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), "1", str(stages), "0", str(segments), str(tmp_path / "out.raw")],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and f"rc=0 w={w} h={h}" in r.stdout, r.stdout + r.stderr
    got = np.fromfile(tmp_path / "out.raw", "<u2")
    assert np.array_equal(got, want[3][0]) and np.array_equal(got.reshape(h, w), img)
