"""Build libicer_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libicer_hip.so")
SOURCES = ["api.hip"]
# the decoder's own sources do not go into libicer_hip.so; every other file under csrc/ is a dependency
DEC_ONLY = {"decoder.hip", "decoder_core.hpp", "decoder_plan.hpp", "decoder_wave.hpp"}


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f not in DEC_ONLY] + [os.path.join(PKG, "..", "include", "icer_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _tuning_flags():
    """experiments only: ICER_WG_WAVES=<n> builds the workgroup coder with n wavefronts per coding unit (default 16),
    ICER_QUEUE_DEPTH=<n> the pipeline with queues of n chunks between its waves (default 4)"""
    flags = []
    for name in ("ICER_WG_WAVES", "ICER_QUEUE_DEPTH"):
        v = os.environ.get(name)
        if v:
            flags.append(f"-D{name}={int(v)}")
    return flags


def build_profiling_library(verbose: bool = False) -> str:
    """Separate build with per-phase s_memtime counters (-DICER_PHASE_TIMERS) for tools/phase_profile.py."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = os.path.join(PKG, "libicer_hip_prof.so")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-DICER_PHASE_TIMERS", "-o", out] + _tuning_flags() + \
        [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wall",
           "-o", LIB] + _tuning_flags() + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


DEC_LIB = os.path.join(PKG, "libicer_hip_dec.so")
DEC_DEPS = ["decoder.hip", "decoder_core.hpp", "decoder_plan.hpp", "decoder_wave.hpp", "wave.hpp", "plan.hpp", "icer_tables.hpp"]


def build_decoder_library(force: bool = False, verbose: bool = False) -> str:
    """libicer_hip_dec.so: the decoder, a separate library -- see include/icer_hip_dec.h."""
    deps = [os.path.join(CSRC, f) for f in DEC_DEPS] + [os.path.join(PKG, "..", "include", "icer_hip_dec.h")]
    if not force and os.path.exists(DEC_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(DEC_LIB) for d in deps):
        return DEC_LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wall", "-o", DEC_LIB,
           os.path.join(CSRC, "decoder.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return DEC_LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
    print(build_decoder_library(force=True, verbose=True))
