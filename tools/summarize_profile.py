#!/usr/bin/env python3
"""Summarise the rocprofv3 output of tools/profile_round.sh into small files for profiles/.

Writes <out>/summary/<round>_kernel_stats.csv (rocprofv3's own --stats table), <round>_conv_roofline.json and
<round>_traffic.json.  HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE is reported in KiB-like units of 1 KB and
counts 64 B per 128-B request on gfx950 for wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE is exact.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, rnd = sys.argv[1], sys.argv[2]
sumdir = os.path.join(out, "summary")
os.makedirs(sumdir, exist_ok=True)
FLOPS_PER_FRAME = 71.49e9


def is_conv(name):
    """launches of the conv stack: conv_stream_kernel (all shapes) and, since round 3, stem_kernel (conv1 + pool1 [+ gen_input_batch])"""
    return "conv_stream" in name or "stem_kernel" in name


def find(sub, pat):
    g = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return g[0] if g else None


res = {}
st = find("stats", "*kernel_stats.csv")
if st:
    rows = list(csv.DictReader(open(st)))
    with open(os.path.join(sumdir, rnd + "_kernel_stats.csv"), "w") as f:
        f.write(open(st).read())
    conv = [r for r in rows if is_conv(r["Name"])]
    calls = sum(int(r["Calls"]) for r in conv)
    tot_ns = sum(float(r["TotalDurationNs"]) for r in conv)
    allk = sum(float(r["TotalDurationNs"]) for r in rows)
    sq = [r for r in rows if "post_kernel" in r["Name"] or "joints_kernel" in r["Name"]]  # one per frame (round 3: the pyramid kernel is gone)
    frames = int(sq[0]["Calls"]) if sq else 0
    res["frames_profiled"] = frames
    res["conv_kernel"] = conv[0]["Name"] if conv else None
    res["conv_calls_per_frame"] = calls / max(frames, 1)
    res["conv_avg_us_per_launch"] = tot_ns / max(calls, 1) / 1e3
    res["conv_ms_per_frame"] = tot_ns / max(frames, 1) / 1e6
    res["all_kernels_ms_per_frame"] = allk / max(frames, 1) / 1e6
    res["conv_tflops"] = FLOPS_PER_FRAME / (tot_ns / max(frames, 1) * 1e-9) / 1e12
    res["conv_frac_of_fp32_mfma_peak"] = res["conv_tflops"] / 157.3


def counter_sum(sub, name):
    f = find(sub, "*counter_collection.csv")
    if not f:
        return None, 0
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != name:
            continue
        k = r["Kernel_Name"].split("(")[0]
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
    return tot, n


ft, fn = counter_sum("fetch", "FETCH_SIZE")
wt, wn = counter_sum("write", "WRITE_SIZE")
if ft and wt and res.get("frames_profiled"):
    fr = res["frames_profiled"]
    per_kernel = {}
    for k in sorted(set(ft) | set(wt)):
        per_kernel[k] = {"fetch_KB_raw_per_frame": ft.get(k, 0) / fr, "write_KB_per_frame": wt.get(k, 0) / fr}
    conv_f = sum(v for k, v in ft.items() if is_conv(k)) / fr
    conv_w = sum(v for k, v in wt.items() if is_conv(k)) / fr
    traffic = {
        "note": "per frame, conv_stream_kernel + stem_kernel launches only (all tile shapes, profiling twins included); FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request), WRITE_SIZE as is; units KB=1024 B",
        "fetch_bytes_per_frame": conv_f * 2 * 1024, "write_bytes_per_frame": conv_w * 1024,
        "hbm_bytes_per_frame": (conv_f * 2 + conv_w) * 1024, "per_kernel": per_kernel}
    json.dump(traffic, open(os.path.join(sumdir, rnd + "_traffic.json"), "w"), indent=1)
    res["conv_hbm_bytes_per_frame"] = traffic["hbm_bytes_per_frame"]
mt, mn = counter_sum("mfma", "SQ_VALU_MFMA_BUSY_CYCLES")
if mt and res.get("frames_profiled") and res.get("conv_ms_per_frame"):
    fr = res["frames_profiled"]
    busy = sum(v for k, v in mt.items() if is_conv(k)) / fr  # SIMD-cycles per frame, all 1024 SIMDs
    res["mfma_busy_simd_cycles_per_frame"] = busy
    # utilisation over the time the conv kernels occupy the stream (conv_ms_per_frame, from the --stats pass), against the
    # 2.4 GHz of the peak figure: busy / (1024 SIMDs x 2.4e9 x seconds)
    res["mfma_util_vs_2p4GHz"] = busy / (1024 * 2.4e9 * res["conv_ms_per_frame"] * 1e-3)
    res["mfma_note"] = "SQ_VALU_MFMA_BUSY_CYCLES summed over the conv_stream_kernel + stem_kernel launches of a frame (64 cycles per v_mfma_f32_32x32x2_f32, padded tiles included)"
for tag in ("stats", "fetch", "write", "mfma"):
    b = os.path.join(out, "bench_%s.json" % tag)
    if os.path.exists(b) and os.path.getsize(b):
        try:
            res["bench_under_" + tag] = json.loads(open(b).read().strip().splitlines()[-1])
        except Exception as e:  # noqa
            res["bench_under_" + tag] = "unparsed: %s" % e
json.dump(res, open(os.path.join(sumdir, rnd + "_conv_roofline.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if not k.startswith("bench_")}, indent=1))
