#!/bin/bash
# tuning aid: per-layer kernel times (rocprof) for forced tile / split-K / ring-depth choices
export TMPDIR=/tmp
mkdir -p gpurun_out/sweep
for cfg in ${CFGS:-"64,64,1,1" "64,64,1,2" "64,64,1,5" "64,32,2,1" "32,32,4,1"}; do   # BM,BN,KG,ks (VNECT_FORCE_TILE)
  for deep in 0; do
    tag=$(echo $cfg | tr , x)_d$deep
    VNECT_FORCE_TILE=$cfg timeout 200 rocprofv3 --kernel-trace --output-format csv -d $PWD/gpurun_out/sweep/$tag -o lt -- python3 tools/layer_table.py > /dev/null 2>&1
    python tools/trace_layers.py gpurun_out/sweep/$tag/lt_kernel_trace.csv > gpurun_out/sweep/$tag.txt
    rm -rf gpurun_out/sweep/$tag
  done
done
