#!/bin/bash
# pipelined / multi-stream legs under several environment settings, interleaved, ONE gpurun call
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in "$@"; do
    r=$(env $v python bench.py --steps 300 --warmup 30 --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sync %.1f  pipelined %.1f  streams3 %.1f  split-pipelined %.1f' % (d['value'], d['pipelined_frames_per_s_per_gpu'], d['three_streams_on_one_handle_frames_per_s'], d['fp32_split']['pipelined_frames_per_s_per_gpu']))")
    echo "[${v:-default}] $r"
  done
done
