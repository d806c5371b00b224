# rocprofv3 kernel statistics of the fine forward pass (tools/profile_fine.py) under an environment variant: per-kernel average durations
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  NAME=${V%%:*}; KV=${V#*:}; [ "$KV" = "$V" ] && KV=""
  env $(echo $KV | tr ',' ' ') timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fine_$NAME -- python $R/tools/profile_fine.py > $R/gpurun_out/prof_fine_$NAME.log 2>&1
  DB=$(find $R/gpurun_out/prof_fine_$NAME -name "*.db" | head -1); echo "== $NAME"; python $R/tools/rocpd_stats.py $DB $R/gpurun_out/fine_kernel_stats_$NAME.csv | cut -c1-160 | head -14
  rm -rf $R/gpurun_out/prof_fine_$NAME
done
