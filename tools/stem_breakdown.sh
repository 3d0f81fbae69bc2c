#!/bin/bash
# Where the fused stem's time goes (tuning): layer_table's conv1 row (= the whole stem launch) with parts of the kernel switched off
# (VNECT_STEM_DBG: 1 = no conv blocks, 2 = no pooling, 4 = no patch, 16 = no pair GEMM; wrong results).  LT_BF16=1 for the bf16 form; VNECT_LIB for a variant.
cd "$(dirname "$0")/.."
for mode in frame batch; do
  for dbg in ${DBGS:-0 1 2 4 5 7 21}; do
    echo "mode=$mode dbg=$dbg: $(VNECT_STEM=$mode VNECT_STEM_DBG=$dbg timeout -k 10 120 python tools/layer_table.py 2>/dev/null | grep '^conv1')"
  done
done
