#!/bin/bash
# round 3, call 5: fine windows of a lock-step batch side by side (engine_fine_many): parity + throughput by chunk size, both routes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch or fine or tolerance or config5" > gpurun_out/c5_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c5_pytest.log
tail -8 gpurun_out/c5_pytest.log
F=BARK_HIP_FAST_GEMM=1
timeout 600 python tools/batch_ab.py one:BARK_HIP_FINE_BATCH=1 z4:BARK_HIP_FINE_BATCH=4 z8 z16:BARK_HIP_FINE_BATCH=16 z32:BARK_HIP_FINE_BATCH=32 \
    fast1:$F,BARK_HIP_FINE_BATCH=1 fast8:$F fast32:$F,BARK_HIP_FINE_BATCH=32 > gpurun_out/c5_batch_ab.txt 2>&1; cat gpurun_out/c5_batch_ab.txt
