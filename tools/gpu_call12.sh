#!/bin/bash
# round 3, call 12: LayerNorm fused into the lock-step products: parity + per-launch times + whole batches
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch or cross_check or config5 or two_ranks or request_batcher or large" > gpurun_out/c12_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c12_pytest.log
tail -6 gpurun_out/c12_pytest.log
for B in 8 32; do timeout 120 python tools/time_slots.py small $B 4 640 2>&1 | tail -1; done > gpurun_out/c12_time_slots.txt 2>&1; cat gpurun_out/c12_time_slots.txt
F=BARK_HIP_FAST_GEMM=1
timeout 600 python tools/batch_ab.py separate_ln:BARK_HIP_CROSSCHECK=128 fused_ln fast_fused_ln:$F > gpurun_out/c12_batch_ab.txt 2>&1; cat gpurun_out/c12_batch_ab.txt
