#!/bin/bash
# A/B timing of two builds of the library in ONE gpurun call (boxes differ by ~10 %, so never compare across calls).
# Variant a = vnect_amd/lib/libvnect_hip.so, variant b = vnect_amd/lib/libvnect_hip_b.so (e.g. the previous commit's sources).
# Run on GPU:  gpurun -- ./tools/ab.sh [bench args]
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in "" _b; do
    r=$(VNECT_LIB=$PWD/vnect_amd/lib/libvnect_hip$v.so python bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-aux "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f fps  conv %.1f us/frame' % (d['value'], d['roofline'].get('kernel_ms_per_frame', 0)*1e3))")
    echo "variant '${v:-a}': $r"
  done
done
