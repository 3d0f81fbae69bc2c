"""Print the per-layer launch plan and HIP-event times (tuning aid; runs on the GPU box)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights

prec = _native.BF16 if os.environ.get("LT_BF16") == "1" else (_native.FP32_SPLIT if os.environ.get("LT_SPLIT") == "1" else _native.FP32)
scales = [float(x) for x in os.environ.get("LT_SCALES", "1.0,0.8,0.6").split(",")]  # LT_SCALES=1.0: the plan of a pyramid rank
h = _native.Handle(scales, precision=prec)
h.set_weights(synthetic_weights()); h.finalize()
h.upload_frame(0, helpers.synth_frame(1234))
for i in range(5):
    h.infer_resident(0, 1.0 + i, 1.0 + i)
h.set_profiling(True)
acc = None
N = 20
for i in range(N):
    h.infer_resident(0, 10.0 + i, 10.0 + i)
    ls = h.layers()
    if acc is None:
        acc = ls
    else:
        for a, l in zip(acc, ls):
            a["last_ms"] += l["last_ms"]
tot = 0; totf = 0
print("%-22s %6s %5s %5s %4s %4s %2s %5s %8s %7s %6s" % ("layer", "M", "N", "K", "BM", "BN", "ks", "WGs", "us", "TF/s", "ideal"))
for a in acc:
    us = a["last_ms"] / N * 1e3
    tf = a["flops"] / (us * 1e-6) / 1e12 if us > 0 and a["flops"] else 0
    ideal = a["flops"] / 157.3e12 * 1e6
    tot += us; totf += a["flops"]
    print("%-22s %6d %5d %5d %4d %4d %2d %5d %8.1f %7.1f %6.1f" % (a["name"], a["M"], a["N"], a["K"], a["tile_m"], a["tile_n"], a["split_k"], a["workgroups"], us, tf, ideal))
print("total %.1f us, %.1f TF/s" % (tot, totf / (tot * 1e-6) / 1e12))
print(json.dumps(h.timings()))
