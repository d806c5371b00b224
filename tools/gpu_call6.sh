#!/bin/bash
# round 3, call 6: window prompts of a lock-step batch in one pass (batch_prefill_many): parity + throughput, both routes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch or cross_check or config5 or two_ranks or q4_0_generate" > gpurun_out/c6_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c6_pytest.log
tail -8 gpurun_out/c6_pytest.log
F=BARK_HIP_FAST_GEMM=1
timeout 600 python tools/batch_ab.py slotwise:BARK_HIP_CROSSCHECK=16 many fast_slotwise:$F,BARK_HIP_CROSSCHECK=16 fast_many:$F > gpurun_out/c6_batch_ab.txt 2>&1; cat gpurun_out/c6_batch_ab.txt
