#!/bin/bash
# round 3, call 23: PMC counters of the tolerance route in the eight-window fine pass
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
BARK_HIP_FAST_GEMM=1 FINE_WINDOWS=8 timeout 100 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $R/gpurun_out/prof_pmc_fast -- python $R/tools/profile_fine.py > $R/gpurun_out/prof_pmc_fast.log 2>&1
DB=$(find $R/gpurun_out/prof_pmc_fast -name "*.db" | head -1); python $R/tools/rocpd_pmc.py $DB $R/gpurun_out/c23_pmc_fast.json | grep -E "gemm_f16|flash" | cut -c1-170
rm -rf $R/gpurun_out/prof_pmc_fast
