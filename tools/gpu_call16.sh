#!/bin/bash
# round 3, call 16: PMC counters of the exact GEMM in the eight-window fine pass (matrix-core occupancy, stalls, LDS conflicts)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
FINE_WINDOWS=8 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $R/gpurun_out/prof_pmc_gemm -- python $R/tools/profile_fine.py > $R/gpurun_out/prof_pmc_gemm.log 2>&1
DB=$(find $R/gpurun_out/prof_pmc_gemm -name "*.db" | head -1); python $R/tools/rocpd_pmc.py $DB $R/gpurun_out/c16_pmc_gemm.json | cut -c1-170 | head -60
rm -rf $R/gpurun_out/prof_pmc_gemm
