#!/bin/bash
# Round-6 final check in one gpurun call: the whole `-m gpu` suite on the final library, smoke(), then the bench lines to be committed (default run and
# the driver's length) and the two one-GPU rehearsals of the N > 1 lines.
O=gpurun_out/r06f; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "pytest rc $?"; tail -4 $O/gputests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.txt
python3 bench.py > $O/r06_bench_line.json 2> $O/bench.err; echo "bench rc $?"
python3 bench.py --steps 20 --warmup 5 > $O/r06_bench_line_short_run.json 2>> $O/bench.err; echo "short bench rc $?"
VNECT_BENCH_BACKEND=gloo VNECT_BENCH_DEVICE=0 python3 bench.py --gpus 2 --steps 100 --warmup 10 --cpu-seconds 0 --no-aux > $O/r06_rehearsal_replicas_one_gpu.json 2>$O/err_reh2.txt; echo "rehearsal replicas rc $?"
VNECT_BENCH_BACKEND=gloo VNECT_BENCH_DEVICE=0 python3 bench.py --gpus 3 --pyramid-both --steps 100 --warmup 10 --cpu-seconds 0 > $O/r06_rehearsal_pyramid_one_gpu.json 2>$O/err_reh3.txt; echo "rehearsal pyramid rc $?"
python3 tools/explain_scale.py $O/r06_rehearsal_replicas_one_gpu.json $O/r06_rehearsal_pyramid_one_gpu.json > $O/r06_explain_scale_rehearsals.txt 2>&1; cat $O/r06_explain_scale_rehearsals.txt
python3 -c "
import json
for f in ('r06_bench_line.json','r06_bench_line_short_run.json'):
    d=json.load(open('$O/'+f)); print(f, d['value'], d['latency_ms']['value_from_median'], 'bf16', d['bf16']['value'], d['bf16']['latency_ms']['p50'], d['bf16']['latency_ms']['p95'], 'split', d['fp32_split']['value'], 'frac', d['roofline']['frac'], d['roofline']['traffic_source'], 'cs', d['call_surface']['frames_per_s'], d['call_surface']['vs_resident_percent'])
"
