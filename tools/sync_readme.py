"""Fill README.md's round-6 placeholders from the committed bench lines (CPU; run once after tools/r6_final.sh's lines are copied to profiles/)."""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_line.json")))
s = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_line_short_run.json")))
def f(x, nd=0):
    t = ("%." + str(nd) + "f") % x
    w, _, fr = t.partition(".")
    if len(w) > 3: w = w[:-3] + " " + w[-3:]
    return w + ("." + fr if fr else "")
cs = d["call_surface"]
rep = {"FP32_DEF": f(d["value"]), "FP32_SHORT": f(s["value"]), "FP32_MED": f(s["latency_ms"]["value_from_median"]),
       "BF16_DEF": f(d["bf16"]["value"]), "BF16_SHORT": f(s["bf16"]["value"]), "BF16_P50": "%.3f" % s["bf16"]["latency_ms"]["p50"], "BF16_P95": "%.3f" % s["bf16"]["latency_ms"]["p95"],
       "BF16_PIPE": f(d["bf16"]["pipelined_frames_per_s_per_gpu"]), "PIPE": f(d["pipelined_frames_per_s_per_gpu"]), "THREE": f(d["three_streams_on_one_handle_frames_per_s"]),
       "SPLIT_DEF": f(d["fp32_split"]["value"]), "CS_PINNED_PCT": "%+.1f" % cs["vs_resident_percent"]["pinned"], "CS_PAGEABLE_PCT": "%+.1f" % cs["vs_resident_percent"]["pageable"],
       "CS_PINNED": f(cs["frames_per_s"]["pinned"]), "CS_PAGEABLE": f(cs["frames_per_s"]["pageable"]), "CS_RESIDENT": f(cs["frames_per_s"]["resident"]),
       "CPU_FW": "%.1f" % d["cpu_baseline"]["value"], "CPU_CORES": str(d["cpu_baseline"]["cores"]), "CPU_PORT": "%.1f" % d["cpu_baseline_port"]["value"]}
p = os.path.join(ROOT, "README.md")
t = open(p).read()
for k in sorted(rep, key=len, reverse=True):
    t = t.replace(k, rep[k])
open(p, "w").write(t)
print(rep)
