"""Build libicer_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
from __future__ import annotations

import os
import re
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libicer_hip.so")
SOURCES = ["api.hip"]
INCLUDE = os.path.join(PKG, "..", "include")


def _deps(source: str):
    """every file a translation unit pulls in, transitively, by following its quoted #include lines (csrc/ and include/):
    the dependency list of a library is read from the sources, never kept by hand"""
    seen, todo = [], [os.path.join(CSRC, source)]
    while todo:
        f = os.path.normpath(todo.pop())
        if f in seen or not os.path.exists(f):
            continue
        seen.append(f)
        with open(f, errors="replace") as fh:
            for m in re.finditer(r'^\s*#\s*include\s*"([^"]+)"', fh.read(), re.M):
                for base in (os.path.dirname(f), CSRC, INCLUDE):
                    cand = os.path.join(base, m.group(1))
                    if os.path.exists(cand):
                        todo.append(cand)
                        break
    return seen


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for s in SOURCES for d in _deps(s))


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


def build_decoder_library(force: bool = False, verbose: bool = False) -> str:
    """libicer_hip_dec.so: the decoder, a separate library -- see include/icer_hip_dec.h."""
    deps = _deps("decoder.hip")
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
