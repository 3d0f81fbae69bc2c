#!/bin/bash
# Probe (round 3): the 200-tile 3x3 layers of the 46x46 stage as 32x128 tiles (one workgroup owns all 128 channels of its 32 pixels --
# what a tail GEMM behind them would need) against the 64x64 tiles in use.  One gpurun call, per-layer first-to-last-wave times.
cd "$(dirname "$0")/.."
P="res3a_branch2b=32,128,1,1;res3b_branch2b=32,128,1,1;res3c_branch2b=32,128,1,1;res3d_branch2b=32,128,1,1;res5c_branch2b=32,128,1,1"
for prec in "" "LT_BF16=1"; do
  echo "== ${prec:-fp32}: plan in use"
  env $prec python tools/layer_table.py | grep -E "res3._branch2b |res5c_branch2b |^total"
  echo "== ${prec:-fp32}: 32x128"
  env $prec VNECT_PLAN="$P" python tools/layer_table.py | grep -E "res3._branch2b |res5c_branch2b |^total"
done
