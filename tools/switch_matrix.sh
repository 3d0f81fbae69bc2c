#!/bin/bash
# Oracle parity of the ALTERNATIVE launch plans: the end-to-end, pre / post-processing and every-layer tests under each plan switch (the
# defaults are what `pytest -m gpu` covers).  One gpurun call, ~3 minutes:   gpurun -- tools/switch_matrix.sh
cd "$(dirname "$0")/.."
for v in VNECT_NO_ONE=1 VNECT_NO_HEAD_SPLIT=1 VNECT_NO_STEM=1 VNECT_NO_TAIL=1 VNECT_NO_BIG_GRID=1 VNECT_NO_DECONV96=1 ${EXTRA_SWITCHES}; do
  echo "== $v"
  env $v timeout -k 10 300 python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_prepost.py "tests/test_gpu_conv.py::test_conv_stack_every_layer" -x -q 2>&1 | tail -2
done
