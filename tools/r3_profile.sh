#!/bin/bash
# One GPU call: kernel stats of the synchronous fp32 and bf16 loops (rocprofv3 --kernel-trace --stats) + a plan A/B.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3e
rm -rf $OUT; mkdir -p $OUT
for prec in fp32 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$prec -o s -- python3 bench.py --steps 100 --warmup 10 --cpu-seconds 0 --no-aux --precision $prec > $OUT/bench_$prec.json 2> $OUT/stats_$prec.log
  f=$(find $OUT/stats_$prec -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/kernel_stats_$prec.csv
  find $OUT/stats_$prec -name "*kernel_trace.csv" -delete
done
python3 - <<'PY'
import csv, os
out = os.environ.get("OUT", "gpurun_out/r3e")
for prec in ("fp32", "bf16"):
    print("==", prec)
    for r in csv.DictReader(open("%s/kernel_stats_%s.csv" % (out, prec))):
        n = r["Name"]
        short = n.split("(")[0][-70:]
        print("%-72s calls %5s avg %8.1f us  %5s %%" % (short, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
