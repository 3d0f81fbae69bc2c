#!/bin/bash
# Where the wide tail's time goes: the fused res3 / head launches with parts of the tail switched off (WT_DBG builds, wrong results).
cd "$(dirname "$0")/.."
for v in "" _d1 _d2 _d3; do
  for prec in "" "LT_BF16=1"; do
    echo "== lib$v ${prec:-fp32}"
    env $prec VNECT_LIB=vnect_amd/lib/libvnect_hip$v.so python tools/layer_table.py 2>/dev/null | grep -E "res3b_branch2b|res5c_branch2b|^total"
  done
done
echo "== no wide tail"
VNECT_NO_WIDE_TAIL=1 python tools/layer_table.py 2>/dev/null | grep -E "res3b_branch2|res5c_branch2|^total"
