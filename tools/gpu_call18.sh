#!/bin/bash
# round 3, call 18: final build - whole GPU suite, whole-batch throughput of both routes, the round's evidence
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c18_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c18_pytest.log
tail -4 gpurun_out/c18_pytest.log
timeout 400 python tools/batch_ab.py exact fast:BARK_HIP_FAST_GEMM=1 > gpurun_out/c18_batch_ab.txt 2>&1; cat gpurun_out/c18_batch_ab.txt
BATCH_AB_SIZES=16 timeout 200 python tools/batch_ab.py exact16 >> gpurun_out/c18_batch_ab.txt 2>&1; tail -1 gpurun_out/c18_batch_ab.txt
bash tools/collect_profiles.sh r03 > gpurun_out/c18_collect.log 2>&1
ls -la gpurun_out | grep r03_
