"""Pipelined frame rate of ONE stream against the number of lanes (frames in flight); runs on the GPU box.
Usage: python tools/lanes_rate.py [fp32|bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights

prec = _native.BF16 if len(sys.argv) > 1 and sys.argv[1] == "bf16" else _native.FP32
w = synthetic_weights()
for lanes in (1, 2, 3, 4, 5):
    h = _native.Handle([1.0, 0.8, 0.6], precision=prec, lanes=lanes, num_frame_slots=8)
    h.set_weights(w); h.finalize()
    for k in range(8):
        h.upload_frame(k, helpers.synth_frame(1234 + k))
    depth = max(lanes, 1)
    best = 0.0
    for rep in range(3):
        n = 600
        t0 = time.perf_counter()
        for i in range(n):
            if i >= depth:
                h.collect()
            h.submit_resident(i % 8, 10.0 + rep * 100 + i / 30, 10.0 + rep * 100 + i / 30 + 1e-3)
        for _ in range(min(depth, n)):
            h.collect()
        best = max(best, n / (time.perf_counter() - t0))
    print("lanes %d: %.1f frames/s" % (lanes, best), flush=True)
    h.close()
