"""A bounded slice of tests/stress_gpu.py in the GPU gate: randomised geometries / filters / quotas / content through the C
ABI against the oracle, once per coding-unit kernel (the eight-wave pipeline and the barrier-only workgroup coder) and
once with the automatic choice.  Fails on a mismatch and on any coding-unit time-out (icerx_process_stats)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(180)
@pytest.mark.parametrize("coder,seconds,seed", [("pipe", 18, 101), ("wg", 18, 202), ("auto", 12, 303)])
def test_randomised_encodes(coder, seconds, seed):
    env = dict(os.environ, ICER_STRESS_BIG="0.08")
    env.pop("ICER_HIP_CODER", None)
    if coder != "auto":
        env["ICER_HIP_CODER"] = coder
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stress_gpu.py"), str(seconds), str(seed)], env=env,
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "0 mismatches" in r.stdout and "unit time-outs 0" in r.stdout, r.stdout[-2000:]
