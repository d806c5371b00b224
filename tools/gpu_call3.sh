#!/bin/bash
# round 3, call 3: tolerance route v2 (register rings, lazy rescale): correctness + A/B of the pipeline depths
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tolerance_route or fine_eval or mfma_gemm or stage_loops_toy" > gpurun_out/c3_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c3_pytest.log
tail -5 gpurun_out/c3_pytest.log
F=BARK_HIP_FAST_GEMM=1
timeout 400 python tools/fine_ab.py exact d2:$F,BARK_HIP_FAST_DEPTH=2 d3:$F,BARK_HIP_FAST_DEPTH=3 d4:$F,BARK_HIP_FAST_DEPTH=4 \
   nb2:$F,BARK_HIP_FLASH_NB=2 nb4:$F,BARK_HIP_FLASH_NB=4 ks1nb4:$F,BARK_HIP_FLASH_KS=1,BARK_HIP_FLASH_NB=4 ks4nb3:$F,BARK_HIP_FLASH_KS=4,BARK_HIP_FLASH_NB=3 > gpurun_out/c3_fine_ab.txt 2>&1; cat gpurun_out/c3_fine_ab.txt
bash tools/run_prof_fine.sh fast:$F fastd4:$F,BARK_HIP_FAST_DEPTH=4 > gpurun_out/c3_prof_fine.txt 2>&1; cat gpurun_out/c3_prof_fine.txt
timeout 300 python tools/check_routes.py fast small > gpurun_out/c3_check_fast_small.txt 2>&1; tail -3 gpurun_out/c3_check_fast_small.txt
