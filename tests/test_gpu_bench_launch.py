"""`python bench.py --gpus 2` on the one-GPU box: bench.py launches its own two ranks (WORLD_SIZE unset), the ranks share the
physical device through the library's logical devices + gloo (the dry-run path), the headline is the whole C4 batch split over
the ranks (strong scaling), every one of the 256 frames checked against its reference golden before and after the timed step."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_gpus_2_starts_itself_and_reports_the_c4_batch():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "ICER_HIP_VIRTUAL_DEVICES")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--no-batch-configs", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-1500:]
    line = lines[0]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["parity_after_timing"] is True
    assert line["physical_gpus"] == 1 and "DRY RUN" in line["note"]
    assert line["config"]["frames_per_gpu_per_step"] == 128 and "BASELINE configs[3]" in line["config"]["workload"]
    assert line["value"] > 0 and line["coder_events"]["unit_timeouts"] == 0
    assert line["c2_per_rank"]["parity"] is True and line["c2_per_rank"]["n_gpus"] == 2
    # the same split fed from page-locked host memory (icerx_compress_batch_uint16_devices on every rank's device), every frame checked
    assert line["host_fed"]["parity"] is True and line["host_fed"]["n_gpus"] == 2 and line["host_fed"]["value"] > 0


@pytest.mark.parametrize("name,frames", [("C4", 256), ("C5", 64)])
def test_every_frame_of_the_batch_configurations(name, frames):
    """bench.py --sweep: this one GPU plays every rank of the 8-GPU job in turn -- ALL 256 / 64 frames of BASELINE configs[3] / [4],
    each against the reference CPU encoder's length and CRC-32 (tests/golden/batch_golden.json), after the first encode and again
    after two timed ones.  (The gate used to see ranks 0, 3 and 5 only.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "ICER_HIP_VIRTUAL_DEVICES")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sweep", "--config", name], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith('{"sweep"')][-1])
    assert line["frames_checked"] == frames and line["frames_not_bit_exact"] == [] and line["parity"] is True
