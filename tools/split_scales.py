"""Tuning aid: one frame's scales split over two concurrent handles (timing only: each handle post-processes its own scales)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights

w = synthetic_weights()
prec = _native.BF16 if os.environ.get("LT_BF16") == "1" else _native.FP32
def mk(scales):
    h = _native.Handle(scales, precision=prec)
    h.set_weights(w); h.finalize()
    for k in range(4):
        h.upload_frame(k, helpers.synth_frame(1234 + k))
    return h
steps = 300
def run(h, base):
    for i in range(steps):
        h.infer_resident(i % 4, base + i / 30, base + i / 30 + 1e-3)
for split in ([[1.0, 0.8, 0.6]], [[1.0], [0.8, 0.6]], [[1.0, 0.6], [0.8]], [[1.0], [0.8], [0.6]]):
    hs = [mk(s) for s in split]
    for h in hs: run(h, 1.0)
    # lock-step: both halves of frame i start together (a barrier per frame), as one synchronous frame would
    bar = threading.Barrier(len(hs))
    def lock(h, base):
        for i in range(steps):
            bar.wait()
            h.infer_resident(i % 4, base + i / 30, base + i / 30 + 1e-3)
    ths = [threading.Thread(target=lock, args=(h, 100.0)) for h in hs]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print("scales split %-28s: %.3f ms per frame (%.0f frames/s)" % (split, dt / steps * 1e3, steps / dt))
    for h in hs: h.close()
