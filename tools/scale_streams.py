"""Tuning aid: one frame's 3 scales as 3 concurrent single-image pipelines (one handle + host thread each) vs the batched S=3 handle."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights

w = synthetic_weights()
prec = _native.BF16 if os.environ.get("LT_BF16") == "1" else _native.FP32
steps = 300
def mk(scales):
    h = _native.Handle(scales, precision=prec)
    h.set_weights(w); h.finalize()
    for k in range(4):
        h.upload_frame(k, helpers.synth_frame(1234 + k))
    return h
def run(h, base):
    for i in range(steps):
        h.infer_resident(i % 4, base + i / 30, base + i / 30 + 1e-3)
full = mk([1.0, 0.8, 0.6])
run(full, 1.0)
t0 = time.perf_counter(); run(full, 100.0); dt = time.perf_counter() - t0
print("batched S=3 handle: %.1f frames/s (%.3f ms)" % (steps / dt, dt / steps * 1e3))
full.close()
hs = [mk([s]) for s in (1.0, 0.8, 0.6)]
for h in hs: run(h, 1.0)
ths = [threading.Thread(target=run, args=(h, 100.0)) for h in hs]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
dt = time.perf_counter() - t0
print("3 concurrent S=1 pipelines (one per scale): %.1f frame-equivalents/s (%.3f ms per frame of 3 scales)" % (steps / dt, dt / steps * 1e3))
# lock-step variant: a frame is done when all three scales are done (barrier per frame)
bar = threading.Barrier(3)
def run_lock(h, base):
    for i in range(steps):
        h.infer_resident(i % 4, base + i / 30, base + i / 30 + 1e-3)
        bar.wait()
ths = [threading.Thread(target=run_lock, args=(h, 200.0)) for h in hs]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
dt = time.perf_counter() - t0
print("same, joined after every frame: %.1f frames/s (%.3f ms)" % (steps / dt, dt / steps * 1e3))
