#!/bin/bash
# A/B/C... timing of several environment settings in ONE gpurun call (boxes differ by a few per cent, so never compare across calls):
#   gpurun -- ./tools/ab_multi.sh "VNECT_NO_STEM=1" "VNECT_STEM=batch" ""      (an empty string = the defaults)
# Prints fp32 and bf16 frames/s and the conv stack's kernel time per frame, two rounds each, interleaved.
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in "$@"; do
    for prec in ${PRECS:-fp32 bf16}; do
      r=$(env $v python bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-aux --precision $prec | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f fps  conv %.1f us/frame  p50 %.4f ms' % (d['value'], d['roofline'].get('kernel_ms_per_frame', 0)*1e3, d['latency_ms']['p50']))")
      echo "[${v:-default}] $prec $r"
    done
  done
done
