#!/bin/bash
# Where the fused stem's time goes (tuning): layer_table's conv1 row with parts of the kernel switched off (VNECT_STEM_DBG, wrong results).
cd "$(dirname "$0")/.."
for mode in frame batch; do
  for dbg in 0 1 2 4 5 7; do
    echo "mode=$mode dbg=$dbg: $(VNECT_STEM=$mode VNECT_STEM_DBG=$dbg python tools/layer_table.py 2>/dev/null | grep '^conv1')"
  done
done
echo "no stem: $(VNECT_NO_STEM=1 python tools/layer_table.py 2>/dev/null | grep -E '^conv1|^pool1')"
