#!/bin/bash
# round 3, call 10: lock-step decode attention as scores + mix launches: parity + per-launch times + whole batches
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch or cross_check or config5 or two_ranks or q4_0_generate or request_batcher" > gpurun_out/c10_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c10_pytest.log
tail -6 gpurun_out/c10_pytest.log
for B in 8 16 32; do for C in 385 640 900; do timeout 120 python tools/time_slots.py small $B 0 $C 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); u=d['us_per_launch']; print('B', d['B'], 'ctx', d['ctx'], 'fused', u['attention_all_slots_one_workgroup_per_head_and_slot'], 'split', u['attention_all_slots'])"; done; done > gpurun_out/c10_attn_times.txt 2>&1; cat gpurun_out/c10_attn_times.txt
F=BARK_HIP_FAST_GEMM=1
timeout 600 python tools/batch_ab.py fusedattn:BARK_HIP_CROSSCHECK=32 split fast_split:$F > gpurun_out/c10_batch_ab.txt 2>&1; cat gpurun_out/c10_batch_ab.txt
