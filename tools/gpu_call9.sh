#!/bin/bash
# round 3, call 9: exact GEMM / prefill attention with branch-free prefetch loops: parity + timing
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fine or prefill or mfma_gemm or stage_loops or small_model or bench_workload or config5 or large" > gpurun_out/c9_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c9_pytest.log
tail -5 gpurun_out/c9_pytest.log
F=BARK_HIP_FAST_GEMM=1
timeout 300 python tools/fine_ab.py z1 z8:FINE_WINDOWS=8 f1:$F f8:$F,FINE_WINDOWS=8 > gpurun_out/c9_fine_ab.txt 2>&1; cat gpurun_out/c9_fine_ab.txt
bash tools/run_prof_fine.sh exact1 exact8:FINE_WINDOWS=8 > gpurun_out/c9_prof_fine.txt 2>&1; cat gpurun_out/c9_prof_fine.txt
timeout 300 python tools/batch_ab.py exact fast:$F > gpurun_out/c9_batch_ab.txt 2>&1; cat gpurun_out/c9_batch_ab.txt
