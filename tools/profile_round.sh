#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   1. --kernel-trace --stats of the default bench.py command   -> per-kernel time
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes  -> HBM traffic (MI355X_MICROARCH.md, HBM section)
#   3. --pmc SQ_VALU_MFMA_BUSY_CYCLES in its own pass              -> MFMA utilisation of the conv kernels
# Usage: tools/profile_round.sh r01
R=${1:-r01}
export TMPDIR=/tmp
# the statistics describe the measured loop: without the warm-start frames of vnect_finalize (24 + 6 per handle), a profiled run is the
# 10 warm-up + 100 timed frames of the product kernels + 25 frames of the profiling twin, as in rounds 1 and 2
export VNECT_PRIME_FRAMES=0
OUT=$PWD/gpurun_out/prof_$R
rm -rf $OUT; mkdir -p $OUT
# --no-aux: the auxiliary legs of bench.py overlap frames on purpose; the statistics here describe the synchronous headline loop
STEPS="--steps 100 --warmup 10 --cpu-seconds 0 --no-aux ${@:2}"   # e.g. tools/profile_round.sh r01_bf16 --precision bf16
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $R -- python3 bench.py $STEPS > $OUT/bench_stats.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o $R -- python3 bench.py $STEPS > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o $R -- python3 bench.py $STEPS > $OUT/bench_write.json 2> $OUT/write.log
# matrix-pipe busy cycles (summed over the 1024 SIMDs; 64 per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_32x32x16_bf16), its own pass
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/mfma -o $R -- python3 bench.py $STEPS > $OUT/bench_mfma.json 2> $OUT/mfma.log
find $OUT -name "*.csv" | head -20
python3 tools/summarize_profile.py $OUT $R
# keep the merge-back small: drop the per-dispatch traces after summarising
find $OUT -name "*kernel_trace.csv" -size +8M -delete
