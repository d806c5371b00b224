#!/bin/bash
# round 3, call 14: exact GEMM with LDS-staged operands (gemm_lds_kernel): parity + timing against gemm_kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
BARK_HIP_GEMM_LDS=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fine_eval or prefill or mfma_gemm or stage_loops or small_model or in_engine_batch" > gpurun_out/c14_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c14_pytest.log
tail -5 gpurun_out/c14_pytest.log
timeout 300 python tools/fine_ab.py z1 lds1:BARK_HIP_GEMM_LDS=1 z8:FINE_WINDOWS=8 lds8:BARK_HIP_GEMM_LDS=1,FINE_WINDOWS=8 z1b > gpurun_out/c14_fine_ab.txt 2>&1; cat gpurun_out/c14_fine_ab.txt
bash tools/run_prof_fine.sh lds1:BARK_HIP_GEMM_LDS=1 lds8:BARK_HIP_GEMM_LDS=1,FINE_WINDOWS=8 > gpurun_out/c14_prof_fine.txt 2>&1; grep -E "==|gemm" gpurun_out/c14_prof_fine.txt
