#!/bin/bash
# rocprofv3 kernel statistics of the fine forward pass on the GPU box: fine_kernel_stats.sh <tag> [env assignments...]  -> gpurun_out/<tag>_kernel_stats_fine_{1,8}_windows.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for Z in 1 8; do
    rm -rf $R/gpurun_out/prof_fine_$Z
    env FINE_WINDOWS=$Z "$@" timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fine_$Z -- python $R/tools/profile_fine.py > $R/gpurun_out/prof_fine_$Z.log 2>&1
    DB=$(find $R/gpurun_out/prof_fine_$Z -name "*.db" | head -1)
    [ -n "$DB" ] && python $R/tools/rocpd_stats.py $DB $R/gpurun_out/${TAG}_kernel_stats_fine_${Z}_windows.csv | head -12
    rm -rf $R/gpurun_out/prof_fine_$Z
done
