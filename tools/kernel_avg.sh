#!/bin/bash
# In-frame average duration of the kernels matching PATTERN for several builds of the library, one rocprofv3 --kernel-trace --stats run of the
# synchronous bench loop each, in ONE gpurun call:   tools/kernel_avg.sh post_kernel "" _s8 _s24     (variants = suffixes of libvnect_hip*.so)
# PREC=bf16 profiles the bf16 handle.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp VNECT_PRIME_FRAMES=0
PAT=$1; shift
O=$PWD/gpurun_out/kernel_avg; rm -rf $O; mkdir -p $O
for v in "$@"; do
  export VNECT_LIB=$PWD/vnect_amd/lib/libvnect_hip$v.so
  if [ ! -f "$VNECT_LIB" ]; then echo "variant [$v]: not built"; continue; fi
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v$v -o p -- python3 bench.py --steps 100 --warmup 10 --cpu-seconds 0 --no-aux --precision ${PREC:-fp32} > $O/bench$v.json 2> $O/log$v.txt
  f=$(find $O/v$v -name '*kernel_stats.csv' 2>/dev/null | head -1)
  if [ -z "$f" ]; then echo "variant [$v]: no statistics (see $O/log$v.txt)"; continue; fi
  echo "variant [$v]: $(python3 -c "import json,sys; d=json.load(open('$O/bench$v.json')); print('%.1f fps' % d['value'])" 2>/dev/null)"
  grep -h "$PAT" "$f" | awk -F'",' '{n=split($2,a,","); printf "    %-60s calls %s avg %.2f us min %.2f max %.2f\n", substr($1,2,60), a[1], a[3]/1000, a[5]/1000, a[6]/1000}'
  find $O/v$v -name '*kernel_trace.csv' -delete
done
