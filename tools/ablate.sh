#!/bin/bash
# tuning aid: kernel-trace the layer table under each VNECT_ABLATE setting (results are garbage, timings valid)
export TMPDIR=/tmp
mkdir -p gpurun_out/abl
for ab in 0 1 2 4 8 12 14; do
  VNECT_ABLATE=$ab timeout 200 rocprofv3 --kernel-trace --output-format csv -d $PWD/gpurun_out/abl/a$ab -o lt -- python3 tools/layer_table.py > /dev/null 2>&1
  python tools/trace_layers.py gpurun_out/abl/a$ab/lt_kernel_trace.csv > gpurun_out/abl/a$ab.txt
  echo "== ablate $ab"; head -6 gpurun_out/abl/a$ab.txt | tail -5
  rm -rf gpurun_out/abl/a$ab
done
