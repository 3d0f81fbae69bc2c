#!/bin/bash
# Round-4 evidence in one gpurun call: rocprofv3 rounds (fp32, bf16, fp32_split), layer tables, the default bench line, the driver-length
# run, the one-rank pyramid rehearsal under backend nccl, the one-scale rate, the microbenchmarks of the round.  Everything lands under
# gpurun_out/r04/ ; copy what is to be judged into profiles/.
O=gpurun_out/r04; mkdir -p $O
tools/profile_round.sh r04 > $O/prof_fp32.log 2>&1
tools/profile_round.sh r04_bf16 --precision bf16 > $O/prof_bf16.log 2>&1
tools/profile_round.sh r04_split --precision fp32_split > $O/prof_split.log 2>&1
for r in r04 r04_bf16 r04_split; do cp gpurun_out/prof_$r/summary/* $O/ 2>/dev/null; done
python3 tools/layer_table.py > $O/r04_layer_table.txt 2>/dev/null
LT_BF16=1 python3 tools/layer_table.py > $O/r04_bf16_layer_table.txt 2>/dev/null
python3 bench.py > $O/r04_bench_line.json 2> $O/bench.err
python3 bench.py --steps 20 --warmup 5 > $O/r04_bench_line_short_run.json 2>> $O/bench.err
python3 bench.py --pyramid --scales 1.0 --gpus 1 --steps 100 --warmup 10 --cpu-seconds 0 > $O/r04_rehearsal_pyramid_rccl_one_rank.json 2>> $O/bench.err
python3 tools/one_scale_rate.py $O/r04_one_scale_rate.json > $O/r04_one_scale_rate.txt 2>&1
timeout -k 10 120 tools/mfma_valu_coexec > $O/r04_mfma_valu_coexec_microbench.txt 2>&1
python3 tools/bf16_margin_probe.py > $O/r04_bf16_margin_probe.txt 2>&1
ls -la $O | head -40
