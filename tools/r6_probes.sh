#!/bin/bash
# Round-6 probes in one gpurun call: the call surface measured interleaved (item 5), two frames per launch (item 8), host binding at N = 1
# (A/B), what sysfs shows on the box, and the two one-GPU rehearsals of the N > 1 lines with their new per-rank fields.
cd "$(dirname "$0")/.."
O=gpurun_out/r06p; mkdir -p $O
python3 tools/call_surface_ab.py fp32 8 40 > $O/r06_call_surface_fp32.txt 2>$O/err_cs_fp32.txt; echo "call surface fp32 rc $?"
python3 tools/call_surface_ab.py bf16 8 40 > $O/r06_call_surface_bf16.txt 2>$O/err_cs_bf16.txt; echo "call surface bf16 rc $?"
python3 tools/two_frames_per_launch.py fp32 > $O/r06_two_frames_probe_fp32.txt 2>$O/err_tf_fp32.txt; echo "two frames fp32 rc $?"
python3 tools/two_frames_per_launch.py bf16 > $O/r06_two_frames_probe_bf16.txt 2>$O/err_tf_bf16.txt; echo "two frames bf16 rc $?"
{ echo "# KFD nodes / GPU local cpulist / this process's affinity on the GPU box"; ls /sys/class/kfd/kfd/topology/nodes 2>&1 | tr '\n' ' '; echo
  python3 -c "
import os,sys
sys.path.insert(0,'.')
from vnect_amd import parallel as P
print('gpu_bdfs', P.gpu_bdfs()); print('binding(0)', {k:(v if k!='cpus' else (P.format_cpulist(v) if v else v)) for k,v in P.rank_binding(0, allowed=sorted(os.sched_getaffinity(0))).items()})
print('affinity', P.format_cpulist(sorted(os.sched_getaffinity(0))), 'n', len(os.sched_getaffinity(0)))
try: print('cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e: print('cpu.max ?', e)
"; } > $O/r06_box_topology.txt 2>&1
for rep in 1 2 3; do
  for v in "" "--bind"; do
    r=$(python3 bench.py --steps 300 --warmup 30 --cpu-seconds 0 --no-aux $v 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f fps  p50 %.4f  clock %s MHz  binding %s' % (d['value'], d['latency_ms']['p50'], d['per_rank'][0]['shader_clock_mhz'], d['ranks'][0]['host_binding']))")
    echo "[N=1 ${v:-unbound}] $r"
  done
done > $O/r06_ab_bind_n1.txt 2>&1; echo "bind A/B done"
VNECT_BENCH_BACKEND=gloo VNECT_BENCH_DEVICE=0 python3 bench.py --gpus 2 --steps 100 --warmup 10 --cpu-seconds 0 --no-aux > $O/r06_rehearsal_replicas_one_gpu.json 2>$O/err_reh2.txt; echo "rehearsal replicas rc $?"
VNECT_BENCH_BACKEND=gloo VNECT_BENCH_DEVICE=0 python3 bench.py --gpus 3 --pyramid-both --steps 100 --warmup 10 --cpu-seconds 0 > $O/r06_rehearsal_pyramid_one_gpu.json 2>$O/err_reh3.txt; echo "rehearsal pyramid rc $?"
tail -n 30 $O/r06_call_surface_fp32.txt $O/r06_call_surface_bf16.txt $O/r06_ab_bind_n1.txt $O/r06_box_topology.txt
head -8 $O/r06_two_frames_probe_fp32.txt $O/r06_two_frames_probe_bf16.txt
