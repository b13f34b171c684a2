import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np
from icer_compression_amd import api, synth
res = {}
for (w, h, st, sg, n, seed) in [(512, 384, 3, 8, 4, 7), (2048, 2048, 4, 16, 4, 12345), (1024, 768, 2, 2, 3, 11), (640, 480, 4, 5, 6, 3)]:
    frames = synth.gray_batch(n, w, h, seed, 1)
    enc = api.Encoder(w, h, 1, st, 0, sg, max_frames=n)
    enc.encode_host(frames, 2 * w * h)
    res[f"{w}x{h}_{st}_{sg}_{n}"] = enc.routing()["routed_units"]
    enc.close()
print(json.dumps(res))
