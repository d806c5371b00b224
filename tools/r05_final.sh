#!/bin/bash
# round 5, final GPU call at HEAD: the whole GPU suite, the bench line, rocprofv3 kernel statistics of the same bench command and of the decode step,
# FETCH_SIZE / WRITE_SIZE passes at the roofline's own context (385), in-kernel time lines of the f16 and the q4_0 decode step, a soak of cloned contexts
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out
N=r05
( timeout 100 python tools/state_race_demo.py lag_old; timeout 100 python tools/state_race_demo.py lag_fix ) 2>&1 | grep -v "^bark-mi355x" > gpurun_out/${N}_state_race_demo.txt; cat gpurun_out/${N}_state_race_demo.txt | cut -c1-200
timeout 1100 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/${N}_gpu_suite_at_head.log 2>&1; echo "pytest rc $?" >> gpurun_out/${N}_gpu_suite_at_head.log; tail -22 gpurun_out/${N}_gpu_suite_at_head.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/${N}_bench_small_n1.json 2> gpurun_out/${N}_bench.err; tail -c 700 gpurun_out/${N}_bench_small_n1.json; echo
( timeout 130 python tools/clone_stress.py 100000 mini 8 3 85 ) > gpurun_out/${N}_clone_stress_final_8_threads.txt 2>&1; grep "clone_stress:\| coarse: \| semantic: \|errors" gpurun_out/${N}_clone_stress_final_8_threads.txt | cut -c1-250 | tail -5
if [ -f bark.cpp_amd/lib/libbark_trace.so ]; then
    timeout 150 python tools/trace_decode.py small 640 gpurun_out/${N}_trace_decode_step.json > gpurun_out/${N}_trace_decode_step.txt 2>&1
    timeout 150 python tools/trace_decode.py small 640 gpurun_out/${N}_trace_q4_decode_step.json q4_0 > gpurun_out/${N}_trace_q4_decode_step.txt 2>&1
    head -3 gpurun_out/${N}_trace_q4_decode_step.txt | cut -c1-300
fi
timeout 200 python tools/lock_step_timeline.py small 640 > gpurun_out/${N}_lock_step_timeline.txt 2>&1; cp gpurun_out/lock_step_timeline.json gpurun_out/${N}_lock_step_timeline.json; tail -8 gpurun_out/${N}_lock_step_timeline.txt | cut -c1-260
cd /tmp && export TMPDIR=/tmp
stats() {
    local name=$1; shift
    timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$name -- "$@" > $R/gpurun_out/prof_$name.log 2>&1
    local DB=$(find $R/gpurun_out/prof_$name -name "*.db" | head -1)
    [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB $R/gpurun_out/${N}_kernel_stats_$name.csv > /dev/null || echo "rocprofv3 produced no database for $name (see prof_$name.log)" > $R/gpurun_out/${N}_kernel_stats_$name.csv
    rm -rf $R/gpurun_out/prof_$name
}
stats bench python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batched --no-q4 --no-large --no-fast --no-roofline-legs
stats decode python $R/tools/profile_decode.py f16 200 385
FINE_WINDOWS=8 stats fine_8_windows python $R/tools/profile_fine.py
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/prof_pmc_$C -- python $R/tools/profile_decode.py f16 48 385 > $R/gpurun_out/prof_pmc_$C.log 2>&1
    DB=$(find $R/gpurun_out/prof_pmc_$C -name "*.db" | head -1); [ -n "$DB" ] && python $R/tools/rocpd_pmc.py $DB $R/gpurun_out/${N}_pmc_$C.json > /dev/null
done
[ -s $R/gpurun_out/${N}_pmc_FETCH_SIZE.json ] && [ -s $R/gpurun_out/${N}_pmc_WRITE_SIZE.json ] && python $R/tools/derive_pmc_decode_step.py $R/gpurun_out/${N}_pmc_FETCH_SIZE.json $R/gpurun_out/${N}_pmc_WRITE_SIZE.json $R/gpurun_out/${N}_pmc_decode_step.json 385
rm -rf $R/gpurun_out/prof_pmc_*
head -12 $R/gpurun_out/${N}_kernel_stats_decode.csv | cut -c1-200; cat $R/gpurun_out/${N}_pmc_decode_step.json 2>/dev/null | head -20
