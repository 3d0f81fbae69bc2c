"""Probe (no new kernels): what would TWO frames per launch buy?  A handle with scales [1.0, 0.8, 0.6, 1.0, 0.8, 0.6]
already runs the conv stack of two 3-scale frames as ONE batch of six images (M doubles, half the launches per frame); its merge is a
6-scale merge, so its joints are not two frames' joints -- a TIMING proxy only.  Compared in one process, interleaved rounds:
  (a) the 3-scale handle, synchronous frames            -> frames/s
  (b) the 6-image handle, synchronous "double frames"   -> 2 x double-frames/s
  (c) the 3-scale handle three frames deep (three lanes) -> frames/s     (what the product offers today for throughput)
  (d) the 6-image handle three deep                      -> 2 x
and the per-layer table of the 6-image plan beside 2 x the 3-image plan (S >= 6 falls back to the unfused tail / bone plans: read the
table, not only the total).      python3 tools/two_frames_per_launch.py [fp32|bf16] > profiles/r06_two_frames_probe_<prec>.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import helpers
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights

prec_name = sys.argv[1] if len(sys.argv) > 1 else "fp32"
prec = _native.BF16 if prec_name == "bf16" else _native.FP32
W = synthetic_weights()
S3, S6 = [1.0, 0.8, 0.6], [1.0, 0.8, 0.6, 1.0, 0.8, 0.6]
hs = {}
for name, sc in (("3", S3), ("6", S6)):
    h = _native.Handle(sc, precision=prec, lanes=3, num_frame_slots=8)
    h.set_weights(W); h.finalize()
    for k in range(8):
        h.upload_frame(k, helpers.synth_frame(1234 + k))
    hs[name] = h
clk = [100.0]


def sync(h, n):
    t0 = time.perf_counter()
    for i in range(n):
        clk[0] += 1 / 30
        h.infer_resident(i % 8, clk[0], clk[0] + 1e-3)
    return n / (time.perf_counter() - t0)


def deep(h, n, depth=3):
    t0 = time.perf_counter()
    for i in range(n):
        if i >= depth:
            h.collect()
        clk[0] += 1 / 30
        h.submit_resident(i % 8, clk[0], clk[0] + 1e-3)
    for _ in range(depth):
        h.collect()
    return n / (time.perf_counter() - t0)


for h in hs.values():
    sync(h, 50)
res = {"a": [], "b": [], "c": [], "d": []}
for rep in range(5):
    res["a"].append(sync(hs["3"], 300))
    res["b"].append(2 * sync(hs["6"], 150))
    res["c"].append(deep(hs["3"], 300))
    res["d"].append(2 * deep(hs["6"], 150))
med = {k: float(np.median(v)) for k, v in res.items()}
print("%s, frames/s (median of 5 interleaved rounds; min-max):" % prec_name)
for k, what in (("a", "3-scale handle, synchronous"), ("b", "6-image handle (two frames per launch), synchronous x2"),
                ("c", "3-scale handle, three frames deep on three lanes"), ("d", "6-image handle, three deep x2")):
    print("  (%s) %-58s %8.1f   (%.1f - %.1f)" % (k, what, med[k], min(res[k]), max(res[k])))
print("  two frames per launch vs synchronous 3-scale: %+.1f %%;  vs three lanes: %+.1f %%;  three-deep 6-image vs three lanes: %+.1f %%"
      % (100 * (med["b"] / med["a"] - 1), 100 * (med["b"] / med["c"] - 1), 100 * (med["d"] / med["c"] - 1)))

# per-layer: the 6-image plan beside 2 x the 3-image plan
tabs = {}
for name, h in hs.items():
    h.set_profiling(True)
    acc = None
    N = 20
    for i in range(N):
        clk[0] += 1 / 30
        h.infer_resident(i % 8, clk[0], clk[0] + 1e-3)
        ls = h.layers()
        if acc is None:
            acc = ls
        else:
            for a, l in zip(acc, ls):
                a["last_ms"] += l["last_ms"]
    for a in acc:
        a["us"] = a["last_ms"] / N * 1e3
    tabs[name] = acc
    t = h.timings()
    print("%s-image plan: %d conv launches, conv slots %.1f us per launch-set, frame (HIP events) %.1f us" % (
        name, t["conv_launches"], t["conv_slot_ms"] / t["frames"] * 1e3, t["total_ms"] / t["frames"] * 1e3))
    h.set_profiling(False)
print("\n%-44s %5s %5s %9s | %-44s %5s %5s %9s" % ("3-image plan layer", "tile", "WGs", "2 x us", "6-image plan layer", "tile", "WGs", "us"))
a3 = [a for a in tabs["3"] if a["us"] > 0]
a6 = [a for a in tabs["6"] if a["us"] > 0]
for i in range(max(len(a3), len(a6))):
    l = a3[i] if i < len(a3) else None
    r = a6[i] if i < len(a6) else None
    ls = "%-44s %5s %5d %9.1f" % (l["name"][:44], "%dx%d" % (l["tile_m"], l["tile_n"]), l["workgroups"], 2 * l["us"]) if l else " " * 66
    rs = "%-44s %5s %5d %9.1f" % (r["name"][:44], "%dx%d" % (r["tile_m"], r["tile_n"]), r["workgroups"], r["us"]) if r else ""
    print(ls + " | " + rs)
print("sum: 2 x 3-image %.1f us, 6-image %.1f us" % (2 * sum(a["us"] for a in a3), sum(a["us"] for a in a6)))
for h in hs.values():
    h.close()
