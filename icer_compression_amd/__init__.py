"""icer_compression_amd -- MI355X (gfx950) native ICER image encoder behind lib_icer's C API.

Only what the encode hot path needs lives here:
  csrc/      HIP kernels + the C-ABI shared library (include/icer_hip.h)
  api.py     host-side mirror of the reference interface over that ABI (ctypes)
  build.py   in-tree hipcc build
  synth.py   deterministic synthetic frames for tests and bench
"""
import os as _os

# more hardware queues than the HIP runtime's default of 4 (read once, when the runtime starts): the host-fed batch keeps
# several encoder streams busy at a time, csrc/api.hip RuntimeDefaults
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import api, synth  # noqa: F401,E402
