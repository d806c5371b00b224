#!/bin/bash
# round 3, call 1: full GPU test suite on the pruned build, exact-GEMM A/B, evidence collection
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/c1_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c1_pytest.log
tail -3 gpurun_out/c1_pytest.log
timeout 200 python tools/fine_ab.py base:BARK_HIP_GEMM16=0 g16:BARK_HIP_GEMM16=1 base2:BARK_HIP_GEMM16=0 > gpurun_out/c1_fine_ab.txt 2>&1
cat gpurun_out/c1_fine_ab.txt
bash tools/collect_profiles.sh r03 > gpurun_out/c1_collect.log 2>&1
tail -20 gpurun_out/c1_collect.log
head -c 600 gpurun_out/r03_bench_small_n1.json
