#!/bin/bash
# Runs on the MI355X box (gpurun): regenerates the round's evidence under gpurun_out/ (copied into profiles/ afterwards).
#   bench line; rocprofv3 kernel statistics (bench command with the timing legs switched off, decode steps alone, fine passes of one and
#   eight windows, ONE lock step of 8 / 64 slots through bark_hip_profile_lock_step); FETCH_SIZE / WRITE_SIZE PMC passes (separate runs,
#   --kernel-trace only) and the per-step / per-kernel traffic derived from them; the lock step's per-launch time line; whole-job
#   throughput by slot count; in-kernel time line of the decode step when the trace build exists (bark.cpp_amd/build_variant.sh trace -DBARK_TRACE).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-r06}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/${N}_bench_small_n1.json 2> gpurun_out/${N}_bench.err
timeout 300 python tools/lock_step_timeline.py small 640 > gpurun_out/${N}_lock_step_timeline.txt 2>&1; cp gpurun_out/lock_step_timeline.json gpurun_out/${N}_lock_step_timeline.json
timeout 600 python tools/batch_throughput.py 256 small > gpurun_out/${N}_batch_throughput.txt 2>&1
timeout 120 python tools/profile_codec.py > gpurun_out/${N}_codec_time.txt 2>&1
timeout 300 python tools/fine_ab.py c1_default c1m:BARK_HIP_FINE_ORDER=c1m c1_rows_attn:BARK_HIP_CROSSCHECK=512 c1m_rows_attn:BARK_HIP_CROSSCHECK=512,BARK_HIP_FINE_ORDER=c1m c1m_x8_default:FINE_WINDOWS=8 c1_x8:BARK_HIP_FINE_ORDER=c1,FINE_WINDOWS=8 c1m_x8_rows_attn:BARK_HIP_CROSSCHECK=512,FINE_WINDOWS=8 > gpurun_out/${N}_fine_ab.txt 2>&1
if [ -f bark.cpp_amd/lib/libbark_attnw.so ]; then
    timeout 120 python tools/attnw_phases.py 1 > gpurun_out/${N}_attnw_phases_1_window.txt 2>&1
    timeout 120 python tools/attnw_phases.py 8 > gpurun_out/${N}_attnw_phases_8_windows.txt 2>&1
fi
if [ -f bark.cpp_amd/lib/libbark_trace.so ]; then
    timeout 300 python tools/trace_decode.py small 640 gpurun_out/${N}_trace_decode_step.json > gpurun_out/${N}_trace_decode_step.txt 2>&1
fi
cd /tmp && export TMPDIR=/tmp
stats() {   # stats <name> <command...>: rocprofv3 kernel statistics of a command -> gpurun_out/${N}_kernel_stats_<name>.csv
    local name=$1; shift
    timeout 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$name -- "$@" > $R/gpurun_out/prof_$name.log 2>&1
    local DB=$(find $R/gpurun_out/prof_$name -name "*.db" | head -1)
    [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB $R/gpurun_out/${N}_kernel_stats_$name.csv > /dev/null || echo "rocprofv3 produced no database for $name (see prof_$name.log)" > $R/gpurun_out/${N}_kernel_stats_$name.csv
    rm -rf $R/gpurun_out/prof_$name
}
stats bench python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batched --no-q4 --no-large --no-fast --no-roofline-legs
stats decode python $R/tools/profile_decode.py
stats fine_1_window_c1 python $R/tools/profile_fine.py
BARK_HIP_FINE_ORDER=c1m stats fine_1_window_c1m python $R/tools/profile_fine.py
FINE_WINDOWS=8 stats fine_8_windows python $R/tools/profile_fine.py
stats lock_step python $R/tools/lock_step_timeline.py small 640
for C in ${PMC_COUNTERS:-FETCH_SIZE WRITE_SIZE}; do   # PMC_COUNTERS="" skips the passes
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/prof_pmc_$C -- python $R/tools/profile_decode.py f16 72 385 > $R/gpurun_out/prof_pmc_$C.log 2>&1      # context 385 = the mean context of the bench workload: the roofline's own
    DB=$(find $R/gpurun_out/prof_pmc_$C -name "*.db" | head -1); python $R/tools/rocpd_pmc.py $DB $R/gpurun_out/${N}_pmc_$C.json > /dev/null
done
[ -s $R/gpurun_out/${N}_pmc_FETCH_SIZE.json ] && [ -s $R/gpurun_out/${N}_pmc_WRITE_SIZE.json ] && python $R/tools/derive_pmc_decode_step.py $R/gpurun_out/${N}_pmc_FETCH_SIZE.json $R/gpurun_out/${N}_pmc_WRITE_SIZE.json $R/gpurun_out/${N}_pmc_decode_step.json 385 > /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES -d $R/gpurun_out/prof_pmc_mfma -- env FINE_WINDOWS=8 python $R/tools/profile_fine.py > $R/gpurun_out/prof_pmc_mfma.log 2>&1
DB=$(find $R/gpurun_out/prof_pmc_mfma -name "*.db" | head -1); [ -n "$DB" ] && python $R/tools/rocpd_pmc.py $DB $R/gpurun_out/${N}_pmc_mfma_fine_8_windows.json > /dev/null
rm -rf $R/gpurun_out/prof_pmc_*
ls -la $R/gpurun_out | grep ${N}_ | tail -30
