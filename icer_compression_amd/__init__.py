"""icer_compression_amd -- MI355X (gfx950) native ICER image encoder behind lib_icer's C API.

Only what the encode hot path needs lives here:
  csrc/      HIP kernels + the C-ABI shared library (include/icer_hip.h)
  api.py     host-side mirror of the reference interface over that ABI (ctypes)
  build.py   in-tree hipcc build
  synth.py   deterministic synthetic frames for tests and bench
"""
from . import api, synth  # noqa: F401

# (Importing the package does not touch the environment.  The host-fed batch wants GPU_MAX_HW_QUEUES=8 -- a process-wide
# HIP runtime setting that has to be in place before the runtime starts: bench.py, tools/ and tests/conftest.py set it for
# themselves, a host program does the same, INTEGRATION.md "Hardware queues".)
