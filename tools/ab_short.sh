#!/bin/bash
# the driver's run length (--steps 20 --warmup 5), several environment settings, interleaved, in ONE gpurun call
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for v in "$@"; do
    r=$(env $v python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 %.1f  bf16 %.1f  split %.1f  p50 %.4f' % (d['value'], d['bf16']['value'], d['fp32_split']['value'], d['latency_ms']['p50']))")
    echo "[${v:-default}] $r"
  done
done
