"""A bounded slice of tests/stress_gpu.py in the GPU gate: randomised geometries / filters / quotas / content through the C
ABI against the oracle, once per coding-unit kernel (the wave pipeline in both workgroup shapes, the barrier-only workgroup coder)
and once with the automatic choice.  Fails on a mismatch and on any coding-unit time-out (icerx_process_stats)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(180)
@pytest.mark.parametrize("coder,seconds,seed", [("pipe8", 12, 101), ("pipe11", 12, 111), ("wg", 14, 202), ("auto", 10, 303), ("hybrid", 14, 404),
                                                ("hybrid2", 8, 414), ("hybrid4", 10, 424), ("split", 14, 505), ("batch", 14, 606)])
def test_randomised_encodes(coder, seconds, seed):
    """pipe8 / pipe11: the wave pipeline with its workgroup shape pinned (8 waves: batches, 11 waves: single frames --
    the drop-in entry points would always pick 11); wg: the barrier-only workgroup coder; auto: the library's own choice; hybrid: both coders in one launch (what a batch
    gets: the all-but-blank units go to the small workgroup coder beside the pipeline) forced on for single frames too, with the
    one-wave instance a batch uses; hybrid2 / hybrid4: the same with the two- and the four-wave instance (ICER_HIP_LIST_WAVES; four
    is what the launch of a lone large frame uses);
    split: single gray frames with their coding units cut into sub-ranges of 128 chunks, one workgroup each, spliced where the
    workgroups' coder states meet (what the headline frame gets with pieces of 3072 chunks); batch: every gray case as a batch of
    six frames through the pipelined host batch (the batch build of the pipeline kernel: four workgroups per compute unit)"""
    env = dict(os.environ, ICER_STRESS_BIG="0.08")
    env.pop("ICER_HIP_CODER", None)
    env.pop("ICER_HIP_PIPE_WAVES", None)
    env.pop("ICER_HIP_HYBRID", None)
    env.pop("ICER_HIP_HYBRID_FRAMES", None)
    env.pop("ICER_HIP_SPLIT", None)
    env.pop("ICER_HIP_LIST_WAVES", None)
    env.pop("ICER_STRESS_BATCH", None)
    if coder.startswith("pipe"):
        env["ICER_HIP_CODER"] = "pipe"
        env["ICER_HIP_PIPE_WAVES"] = coder[4:]
    elif coder.startswith("hybrid"):
        env["ICER_HIP_HYBRID"] = "90"
        env["ICER_HIP_HYBRID_FRAMES"] = "1"
        if coder[6:]:
            env["ICER_HIP_LIST_WAVES"] = coder[6:]
    elif coder == "split":
        env["ICER_HIP_SPLIT"] = "128"
        env["ICER_STRESS_BIG"] = "0.3"
    elif coder == "batch":
        env["ICER_STRESS_BATCH"] = "6"
    elif coder != "auto":
        env["ICER_HIP_CODER"] = coder
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stress_gpu.py"), str(seconds), str(seed)], env=env,
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "0 mismatches" in r.stdout and "unit time-outs 0" in r.stdout, r.stdout[-2000:]
