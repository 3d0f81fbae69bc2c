#!/bin/bash
# A/B timing of one environment switch in ONE gpurun call:  gpurun -- ./tools/ab_env.sh VNECT_SPLITK_KERNEL=1 [bench args]
cd "$(dirname "$0")/.."
sw="$1"; shift
for rep in 1 2; do
  for v in "" "$sw"; do
    r=$(env $v python bench.py --steps 300 --warmup 30 --cpu-seconds 0 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f fps  conv %.1f us/frame' % (d['value'], d['roofline'].get('kernel_ms_per_frame', 0)*1e3))")
    echo "[${v:-default}] $r"
  done
done
