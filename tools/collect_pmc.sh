#!/bin/bash
# FETCH_SIZE / WRITE_SIZE PMC passes over a short run of decode steps (separate passes, --kernel-trace only), and the per-step / per-launch
# HBM traffic derived from them.  collect_pmc.sh [rNN] [steps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-r04}; STEPS=${2:-48}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 170 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/prof_pmc_$C -- python $R/tools/profile_decode.py f16 $STEPS > $R/gpurun_out/prof_pmc_$C.log 2>&1
    echo "$C pass rc $?"
    DB=$(find $R/gpurun_out/prof_pmc_$C -name "*.db" 2>/dev/null | head -1)
    [ -n "$DB" ] && python $R/tools/rocpd_pmc.py $DB $R/gpurun_out/${N}_pmc_$C.json > /dev/null 2>&1
    tail -3 $R/gpurun_out/prof_pmc_$C.log
done
[ -s $R/gpurun_out/${N}_pmc_FETCH_SIZE.json ] && [ -s $R/gpurun_out/${N}_pmc_WRITE_SIZE.json ] && python $R/tools/derive_pmc_decode_step.py $R/gpurun_out/${N}_pmc_FETCH_SIZE.json $R/gpurun_out/${N}_pmc_WRITE_SIZE.json $R/gpurun_out/${N}_pmc_decode_step.json
rm -rf $R/gpurun_out/prof_pmc_*/
