#!/bin/bash
# Runs on the MI355X box (gpurun): regenerates the round's evidence under gpurun_out/ (copied into profiles/ afterwards).
#   bench line, rocprofv3 kernel statistics (bench command with the timing legs switched off, and decode steps alone),
#   FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, --kernel-trace only), in-kernel time line of the decode step (trace build:
#   bark.cpp_amd/build_variant.sh trace -DBARK_TRACE, built before the call).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-r03}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/${N}_bench_small_n1.json 2> gpurun_out/${N}_bench.err
if [ -f bark.cpp_amd/lib/libbark_trace.so ]; then
    timeout 300 python tools/trace_decode.py small 640 gpurun_out/${N}_trace_decode_step.json > gpurun_out/${N}_trace_decode_step.txt 2>&1
fi
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batched --no-q4 --no-large --no-fast --no-roofline-legs > $R/gpurun_out/prof_bench.log 2>&1
DB=$(find $R/gpurun_out/prof_bench -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB $R/gpurun_out/${N}_kernel_stats_bench.csv > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_decode -- python $R/tools/profile_decode.py > $R/gpurun_out/prof_decode.log 2>&1
DB=$(find $R/gpurun_out/prof_decode -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB $R/gpurun_out/${N}_kernel_stats_decode.csv > /dev/null
for C in ${PMC_COUNTERS:-FETCH_SIZE WRITE_SIZE}; do   # PMC_COUNTERS="" skips the passes (they hung twice in round 3 after working once; the committed summaries are from the first run)
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/prof_pmc_$C -- python $R/tools/profile_decode.py > $R/gpurun_out/prof_pmc_$C.log 2>&1
    DB=$(find $R/gpurun_out/prof_pmc_$C -name "*.db" | head -1); python $R/tools/rocpd_pmc.py $DB $R/gpurun_out/${N}_pmc_$C.json > /dev/null
done
[ -s $R/gpurun_out/${N}_pmc_FETCH_SIZE.json ] && [ -s $R/gpurun_out/${N}_pmc_WRITE_SIZE.json ] && python $R/tools/derive_pmc_gemv_fc.py $R/gpurun_out/${N}_pmc_FETCH_SIZE.json $R/gpurun_out/${N}_pmc_WRITE_SIZE.json $R/gpurun_out/${N}_pmc_gemv_fc.json > /dev/null
rm -rf $R/gpurun_out/prof_bench $R/gpurun_out/prof_decode $R/gpurun_out/prof_pmc_*
ls -la $R/gpurun_out | tail -15
