#!/bin/bash
# Round-6 evidence, part A (one gpurun call, ~12 min): rocprofv3 rounds (fp32, bf16, fp32_split: kernel stats + separate FETCH / WRITE / MFMA-busy
# passes), layer and phase tables, the default bench line, the driver-length line, the single-scale rate.  Everything lands under
# gpurun_out/r06c/ ; copy what is to be judged into profiles/.
O=gpurun_out/r06c; mkdir -p $O
tools/profile_round.sh r06 > $O/prof_fp32.log 2>&1; echo "fp32 profiled"
tools/profile_round.sh r06_bf16 --precision bf16 > $O/prof_bf16.log 2>&1; echo "bf16 profiled"
tools/profile_round.sh r06_split --precision fp32_split > $O/prof_split.log 2>&1; echo "split profiled"
for r in r06 r06_bf16 r06_split; do cp gpurun_out/prof_$r/summary/* $O/ 2>/dev/null; done
python3 tools/layer_table.py > $O/r06_layer_table.txt 2>/dev/null
LT_BF16=1 python3 tools/layer_table.py > $O/r06_bf16_layer_table.txt 2>/dev/null
VNECT_PROF_DETAIL=1 PT_PROD=1 python3 tools/phase_table.py > $O/r06_phase_table.txt 2>/dev/null
VNECT_PROF_DETAIL=1 PT_PROD=1 LT_BF16=1 python3 tools/phase_table.py > $O/r06_bf16_phase_table.txt 2>/dev/null; echo "tables done"
python3 bench.py > $O/r06_bench_line.json 2> $O/bench.err; echo "bench done rc $?"
python3 bench.py --steps 20 --warmup 5 > $O/r06_bench_line_short_run.json 2>> $O/bench.err; echo "short bench done rc $?"
python3 tools/one_scale_rate.py $O/r06_one_scale_rate.json > $O/r06_one_scale_rate.txt 2>&1
ls -la $O | head -50
python3 -c "
import json
for f in ('r06_bench_line.json','r06_bench_line_short_run.json'):
    d=json.load(open('$O/'+f)); print(f, d['value'], d['latency_ms']['p50'], d['latency_ms']['value_from_median'], 'bf16', d['bf16']['value'], d['bf16']['latency_ms']['p50'], d['bf16']['latency_ms']['p95'], 'split', d['fp32_split']['value'], 'frac', d['roofline']['frac'], 'pcie', d['pcie_inclusive_frames_per_s_per_gpu'], d['pcie_inclusive_from_pinned_capture_buffer_frames_per_s_per_gpu'], 'pipe', d['pipelined_frames_per_s_per_gpu'], 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'])
"
