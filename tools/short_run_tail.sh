#!/bin/bash
# Where do the slow frames of a driver-length run (--steps 20 --warmup 5) sit?  N repeats of the driver's own command in ONE gpurun call; per
# repeat: rate, median-based rate, p50 / p95 and the 20 per-frame latencies of the fp32, bf16 and split legs (bench.py prints them for runs of
# <= 64 steps).    gpurun -- tools/short_run_tail.sh 10 [extra bench.py flags]
cd "$(dirname "$0")/.."
N=${1:-10}; shift
O=gpurun_out/short_run; mkdir -p $O
for rep in $(seq 1 $N); do
  python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 "$@" > $O/line_$rep.json 2> $O/err_$rep.txt || { echo "repeat $rep failed"; tail -5 $O/err_$rep.txt; exit 1; }
  python3 - $O/line_$rep.json $rep <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for leg, o in (("fp32", d), ("bf16", d.get("bf16")), ("split", d.get("fp32_split"))):
    if not o:
        continue
    l = o["latency_ms"]
    print("rep %2s %-5s mean %7.1f  median-rate %7.1f  p50 %.4f p95 %.4f (x%.3f)  frames: %s" % (
        sys.argv[2], leg, o["value"], l["value_from_median"], l["p50"], l["p95"], l["p95"] / l["p50"],
        " ".join("%.3f" % x for x in l.get("frames_ms", []))))
PY
done
