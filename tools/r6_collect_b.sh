#!/bin/bash
# Round-6 evidence, part B (one gpurun call): the vendor libraries on the same layers (MIOpen, rocBLAS / hipBLASLt; fp32 and bf16) in the SAME
# call as this repo's own layer tables (same box, same day: VERDICT r5 item 7), the switch matrix, the soak.
O=gpurun_out/r06d; mkdir -p $O
python3 tools/layer_table.py > $O/lt_fp32.txt 2>/dev/null
LT_BF16=1 python3 tools/layer_table.py > $O/lt_bf16.txt 2>/dev/null
{ echo "# tools/vendor_ref.py, fp32 then bf16, on the same box and in the same gpurun call as this repo's layer tables below"; timeout -k 10 420 python3 tools/vendor_ref.py --budget 300; echo; timeout -k 10 420 python3 tools/vendor_ref.py --bf16 --budget 300; echo; echo "# this repo, same call: fp32"; tail -3 $O/lt_fp32.txt; echo "# this repo, same call: bf16"; tail -3 $O/lt_bf16.txt; } > $O/r06_vendor_ref.txt 2>$O/vendor.err
echo "vendor ref done"; tail -4 $O/r06_vendor_ref.txt | cut -c1-200
timeout -k 10 500 tools/switch_matrix.sh > $O/r06_switch_matrix.txt 2>&1; echo "switch matrix rc $?"; tail -8 $O/r06_switch_matrix.txt
