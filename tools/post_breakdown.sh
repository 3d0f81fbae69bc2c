#!/bin/bash
# Where post_kernel's time goes: timing builds of post.hip with parts switched off (-DPOST_DBG=n, wrong results), each run back to back
# under rocprofv3 (tools/post_warm.py).  Builds here (CPU box): tools/post_breakdown.sh build ; on the GPU box: tools/post_breakdown.sh run
cd "$(dirname "$0")/.."
L=vnect_amd/lib
if [ "$1" = build ]; then
  for n in 1 2 3 4; do
    mkdir -p $L/obj_pd$n
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -DPOST_DBG=$n -c vnect_amd/csrc/post.hip -o $L/obj_pd$n/post.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libvnect_hip_pd$n.so $L/obj/conv.o $L/obj_pd$n/post.o $L/obj/stem.o $L/obj/runtime.o -ldl
  done
  exit 0
fi
export TMPDIR=/tmp
O=$PWD/gpurun_out/post_breakdown; rm -rf $O; mkdir -p $O
for v in "" _pd1 _pd2 _pd3 _pd4 $EXTRA_VARIANTS; do
  export VNECT_LIB=$PWD/$L/libvnect_hip$v.so
  if [ ! -f "$VNECT_LIB" ]; then echo "variant [$v]: not built"; continue; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v$v -o p -- python3 tools/post_warm.py > $O/log$v.txt 2>&1
  f=$(find $O/v$v -name '*kernel_stats.csv' 2>/dev/null | head -1)
  if [ -z "$f" ]; then echo "variant [$v]: no statistics (see $O/log$v.txt)"; continue; fi
  echo "variant [$v]: $(grep -h post_kernel "$f" | awk -F, '{print "calls", $(NF-6), "avg ns", $(NF-4), "min", $(NF-2), "max", $(NF-1)}')"
done
