#!/bin/bash
# Round-5 evidence in one gpurun call: rocprofv3 rounds (fp32, bf16, fp32_split), layer tables, phase tables, the default bench line, the
# driver-length run.  Everything lands under gpurun_out/r05c/ ; copy what is to be judged into profiles/.
O=gpurun_out/r05c; mkdir -p $O
tools/profile_round.sh r05 > $O/prof_fp32.log 2>&1; echo "fp32 profiled"
tools/profile_round.sh r05_bf16 --precision bf16 > $O/prof_bf16.log 2>&1; echo "bf16 profiled"
tools/profile_round.sh r05_split --precision fp32_split > $O/prof_split.log 2>&1; echo "split profiled"
for r in r05 r05_bf16 r05_split; do cp gpurun_out/prof_$r/summary/* $O/ 2>/dev/null; done
python3 tools/layer_table.py > $O/r05_layer_table.txt 2>/dev/null
LT_BF16=1 python3 tools/layer_table.py > $O/r05_bf16_layer_table.txt 2>/dev/null
VNECT_PROF_DETAIL=1 PT_PROD=1 python3 tools/phase_table.py > $O/r05_phase_table.txt 2>/dev/null
VNECT_PROF_DETAIL=1 PT_PROD=1 LT_BF16=1 python3 tools/phase_table.py > $O/r05_bf16_phase_table.txt 2>/dev/null; echo "tables done"
python3 bench.py > $O/r05_bench_line.json 2> $O/bench.err; echo "bench done"
python3 bench.py --steps 20 --warmup 5 > $O/r05_bench_line_short_run.json 2>> $O/bench.err
python3 tools/one_scale_rate.py $O/r05_one_scale_rate.json > $O/r05_one_scale_rate.txt 2>&1
ls -la $O | head -40
