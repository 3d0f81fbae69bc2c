"""What does the closing `barrier + torch.cuda.synchronize()` of bench.py's timed region cost (it is INSIDE the region by the contract), and does it
depend on how many streams the process holds (a handle with lanes = 3 owns three streams + graphs)?  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests import helpers
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights
W = synthetic_weights()
torch.cuda.set_device(0)
torch.cuda.synchronize()
for lanes in (1, 3, 1, 3):
    h = _native.Handle([1.0, 0.8, 0.6], lanes=lanes, num_frame_slots=8)
    h.set_weights(W); h.finalize()
    for k in range(8):
        h.upload_frame(k, helpers.synth_frame(1234 + k))
    t = 10.0
    ds, first = [], []
    for rep in range(30):
        for i in range(20):
            t += 1 / 30
            h.infer_resident(i % 8, t, t + 1e-3)
        a = time.perf_counter()
        torch.cuda.synchronize()
        b = time.perf_counter()
        t += 1 / 30
        h.infer_resident(0, t, t + 1e-3)
        c = time.perf_counter()
        ds.append((b - a) * 1e6); first.append((c - b) * 1e3)
    steady = []
    for i in range(50):
        t += 1 / 30
        a = time.perf_counter(); h.infer_resident(i % 8, t, t + 1e-3); steady.append((time.perf_counter() - a) * 1e3)
    print("lanes %d: torch.cuda.synchronize() behind 20 synchronous frames: median %.1f us (min %.1f max %.1f); first frame behind it %.4f ms against %.4f steady" % (
        lanes, np.median(ds), min(ds), max(ds), np.median(first), np.median(steady)), flush=True)
    h.close()
