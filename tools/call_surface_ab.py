"""The call-surface rate, measured properly: VNectEstimator.__call__ takes HOST memory (/root/reference/src/estimator.py:97-99),
so what a caller sees is vnect_infer, not vnect_infer_resident.  One handle, one process, variants INTERLEAVED in rounds so that box drift
cancels; >= 200 frames per variant; medians of the per-frame latency.
  resident   vnect_infer_resident (the bench's `value`: frame already in HBM)
  pageable   vnect_infer from ordinary numpy memory (CPU memcpy into the internal pinned staging buffer, then the copy kernel)
  pinned     vnect_infer from a vnect_frame_buffer, whole 368x368 frame (16-byte aligned: the contiguous form of the copy kernel)
  crop       vnect_infer from a 368x368 crop of a 480x640 frame inside a vnect_frame_buffer at x0 = 101 (byte offset 303: NOT dword-aligned,
             three crops in four of a tracking loop are like this -- the any-alignment form of the copy kernel)
  crop_al    the same crop at x0 = 100 (byte offset 300: dword-aligned rows)
Usage: python3 tools/call_surface_ab.py [fp32|bf16] [rounds] [frames per round]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import helpers
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights

prec_name = sys.argv[1] if len(sys.argv) > 1 else "fp32"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
per = int(sys.argv[3]) if len(sys.argv) > 3 else 40
h = _native.Handle([1.0, 0.8, 0.6], precision=_native.BF16 if prec_name == "bf16" else _native.FP32, num_frame_slots=8, lanes=3)
h.set_weights(synthetic_weights()); h.finalize()
frames = [helpers.synth_frame(1234 + k) for k in range(8)]
big = [helpers.synth_frame(4321 + k, 480, 640) for k in range(2)]
pin = [h.frame_buffer(i, 480, 640) for i in range(2)]
clk = [1.7e9]


def tick():
    clk[0] += 1 / 30
    return clk[0], clk[0] + 1e-3


def prep(variant):
    if variant == "resident":
        for k in range(8):
            h.upload_frame(k, frames[k])
    elif variant == "pinned":
        for i in range(2):
            pin[i].reshape(-1)[:368 * 368 * 3] = frames[i].reshape(-1)
    elif variant in ("crop", "crop_al"):
        for i in range(2):
            pin[i][...] = big[i]


def one(variant, i):
    t = tick()
    if variant == "resident":
        return h.infer_resident(i % 8, *t)
    if variant == "pageable":
        return h.infer(frames[i % 8], *t)
    if variant == "pinned":
        return h.infer(pin[i % 2].reshape(-1)[:368 * 368 * 3].reshape(368, 368, 3), *t)
    x0 = 101 if variant == "crop" else 100
    return h.infer(pin[i % 2][56:56 + 368, x0:x0 + 368, :], *t)


variants = ["resident", "pageable", "pinned", "crop", "crop_al"]
lat = {v: [] for v in variants}
for v in variants:     # warm every path once
    prep(v)
    for i in range(10):
        one(v, i)
for r in range(rounds):
    for v in (variants if r % 2 == 0 else variants[::-1]):
        prep(v)
        for i in range(3):
            one(v, i)
        for i in range(per):
            t0 = time.perf_counter()
            one(v, i)
            lat[v].append((time.perf_counter() - t0) * 1e3)
base = float(np.median(lat["resident"]))
print("%s, %d rounds x %d frames per variant, interleaved; per-frame latency [ms] and the rate of the median frame:" % (prec_name, rounds, per))
for v in variants:
    a = np.array(lat[v])
    print("  %-9s median %.4f  mean %.4f  p95 %.4f   %7.1f frames/s   %+5.2f %% vs resident  (+%.1f us)" % (
        v, np.median(a), a.mean(), np.percentile(a, 95), 1e3 / np.median(a), 100 * (base / np.median(a) - 1), (np.median(a) - base) * 1e3))
h.close()
