#!/bin/bash
# round 3, call 2: first run of the tolerance route (f16 tile GEMM + flash attention)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tolerance_route or fine_eval or mfma_gemm" > gpurun_out/c2_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c2_pytest.log
tail -15 gpurun_out/c2_pytest.log
timeout 300 python tools/check_routes.py fast small > gpurun_out/c2_check_fast_small.txt 2>&1; tail -40 gpurun_out/c2_check_fast_small.txt
timeout 200 python tools/fine_ab.py exact fast:BARK_HIP_FAST_GEMM=1 ks1:BARK_HIP_FAST_GEMM=1,BARK_HIP_FLASH_KS=1 ks4:BARK_HIP_FAST_GEMM=1,BARK_HIP_FLASH_KS=4 > gpurun_out/c2_fine_ab.txt 2>&1; cat gpurun_out/c2_fine_ab.txt
bash tools/run_prof_fine.sh fast:BARK_HIP_FAST_GEMM=1 > gpurun_out/c2_prof_fine.txt 2>&1; cat gpurun_out/c2_prof_fine.txt
