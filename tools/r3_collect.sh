#!/bin/bash
# Round-3 evidence in one gpurun call: rocprofv3 rounds (fp32, bf16, fp32_split), layer tables, the default bench line, short vs long run,
# the bf16 weights-from-global microbenchmark.  Everything lands under gpurun_out/r03/ ; copy what is to be judged into profiles/.
O=gpurun_out/r03; mkdir -p $O
tools/profile_round.sh r03 > $O/prof_fp32.log 2>&1
tools/profile_round.sh r03_bf16 --precision bf16 > $O/prof_bf16.log 2>&1
tools/profile_round.sh r03_split --precision fp32_split > $O/prof_split.log 2>&1
for r in r03 r03_bf16 r03_split; do cp gpurun_out/prof_$r/summary/* $O/ 2>/dev/null; done
python tools/layer_table.py > $O/r03_layer_table.txt 2>/dev/null
LT_BF16=1 python tools/layer_table.py > $O/r03_bf16_layer_table.txt 2>/dev/null
LT_SPLIT=1 python tools/layer_table.py > $O/r03_split_layer_table.txt 2>/dev/null
python bench.py > $O/r03_bench_line.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/r03_bench_line_short_run.json 2>> $O/bench.err
timeout -k 10 120 tools/bf16_bglobal > $O/r03_bf16_bglobal_microbench.txt 2>&1
VNECT_BENCH_BACKEND=gloo VNECT_BENCH_DEVICE=0 python bench.py --gpus 2 --steps 100 --warmup 10 > $O/r03_rehearsal_spawn2_one_gpu.json 2>> $O/bench.err
VNECT_BENCH_BACKEND=gloo VNECT_BENCH_DEVICE=0 python bench.py --gpus 3 --pyramid-both --steps 100 --warmup 10 > $O/r03_rehearsal_pyramid_both_one_gpu.json 2>> $O/bench.err
ls -la $O | head -40
