#!/bin/bash
# round 3, call 4: XCD-aware tile orders (exact GEMM, exact prefill attention, tolerance route)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tolerance_route or fine_eval or mfma_gemm or stage_loops or prefill or small_model or fine_stage" > gpurun_out/c4_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c4_pytest.log
tail -5 gpurun_out/c4_pytest.log
F=BARK_HIP_FAST_GEMM=1
timeout 400 python tools/fine_ab.py exact fast:$F d2:$F,BARK_HIP_FAST_DEPTH=2 nb2:$F,BARK_HIP_FLASH_NB=2 ks4:$F,BARK_HIP_FLASH_KS=4 ks1:$F,BARK_HIP_FLASH_KS=1,BARK_HIP_FLASH_NB=4 > gpurun_out/c4_fine_ab.txt 2>&1; cat gpurun_out/c4_fine_ab.txt
bash tools/run_prof_fine.sh exact fast:$F > gpurun_out/c4_prof_fine.txt 2>&1; cat gpurun_out/c4_prof_fine.txt
