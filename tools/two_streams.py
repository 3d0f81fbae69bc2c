"""Tuning aid: throughput of K independent streams (handles) sharing ONE GPU, each driven by its own host thread."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights

w = synthetic_weights()
prec = _native.BF16 if os.environ.get("LT_BF16") == "1" else _native.FP32
for nstreams in (1, 2, 3):
    hs = []
    for s in range(nstreams):
        h = _native.Handle([1.0, 0.8, 0.6], precision=prec)
        h.set_weights(w); h.finalize()
        for k in range(4):
            h.upload_frame(k, helpers.synth_frame(1234 + 1000 * s + k))
        hs.append(h)
    steps = 300
    def run(h, base):
        for i in range(steps):
            h.infer_resident(i % 4, base + i / 30, base + i / 30 + 1e-3)
    for h in hs: run.__call__(h, 1.0)  # warm-up
    ths = [threading.Thread(target=run, args=(h, 100.0)) for h in hs]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print("%d stream(s) on one GPU: %.1f frames/s total, %.3f ms per frame per stream" % (nstreams, nstreams * steps / dt, dt / steps * 1e3))
    for h in hs: h.close()
