#!/bin/bash
# second final check (one gpurun call): the `-m gpu` suite with the last additions, the soak on the final library, one driver-length line
O=gpurun_out/r06g; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "pytest rc $?"; tail -4 $O/gputests.txt
timeout -k 10 600 python3 tools/soak.py > $O/r06_soak.txt 2>&1; echo "soak rc $?"; tail -6 $O/r06_soak.txt
python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/short.json 2>$O/short.err; echo "short rc $?"
python3 -c "
import json
d=json.load(open('$O/short.json')); print(d['value'], d['latency_ms']['value_from_median'], d['per_rank'][0]['closing_barrier_us'], d['per_rank'][0]['own_elapsed_s'], d['bf16']['value'], d['ranks'][0]['host_binding'])"
