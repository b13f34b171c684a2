"""The torch (device-side) frame generator bench.py uses for the batch configurations equals the numpy one."""
import numpy as np

from icer_compression_amd import synth


def test_torch_generator_equals_numpy():
    import torch
    for (n, w, h, seed, mode) in [(3, 97, 61, 12345, 1), (2, 256, 128, 777, 0), (1, 640, 333, 12345 + 255, 1)]:
        a = synth.gray_batch(n, w, h, seed, mode)
        b = synth.gray_frames_torch(n, w, h, seed, torch.device("cpu"), mode).numpy().view(np.uint16)
        assert np.array_equal(a, b)
