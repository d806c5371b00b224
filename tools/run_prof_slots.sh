# rocprofv3 of the lock-step product kernels (tools/time_slots.py): kernel durations, then SQ counters in a separate pass
R=$GRAFT_REPO_ROOT; K=${1:-4}; B=${2:-32}
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_slots -- python $R/tools/time_slots.py small $B $K 640 > $R/gpurun_out/prof_slots.log 2>&1
DB=$(find $R/gpurun_out/prof_slots -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB $R/gpurun_out/slots_kernel_stats.csv | grep -i "slots\|gemv_batch\|ln_rows\|attn_fused\|TOTAL" | cut -c1-200
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_slots_pmc -- python $R/tools/time_slots.py small $B $K 640 > $R/gpurun_out/prof_slots_pmc.log 2>&1
DB=$(find $R/gpurun_out/prof_slots_pmc -name "*.db" | head -1); python $R/tools/rocpd_pmc.py $DB $R/gpurun_out/slots_pmc.json | grep -i "slots" | cut -c1-220
rm -rf $R/gpurun_out/prof_slots $R/gpurun_out/prof_slots_pmc
