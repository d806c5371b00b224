#!/bin/bash
# Runs on the MI355X box (one gpurun call): the device probe of v_mfma_f32_4x4x1_16b_f32, then every opt-in route of this build against
# the default one (tools/check_routes.py), their timings (tools/batch_ab.py), and the parity tests that touch the changed kernels.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma4x4_probe tools/probes/mfma4x4_probe.hip 2>/dev/null && timeout 60 /tmp/mfma4x4_probe > gpurun_out/mfma4x4_probe.txt 2>&1
cat gpurun_out/mfma4x4_probe.txt
if grep -q rows_in_lanes gpurun_out/mfma4x4_probe.txt; then export BARK_HIP_MFMA4_ROWS_IN_LANES=1; echo "operands swapped"; fi
timeout 200 python tools/check_routes.py batch toy 9 40 valu:BARK_HIP_BATCH_MFMA=0 mfma4:BARK_HIP_BATCH_MFMA=2 > gpurun_out/check_batch_toy.txt 2>&1; tail -c 600 gpurun_out/check_batch_toy.txt; echo
timeout 400 python tools/check_routes.py batch small 32 64 valu:BARK_HIP_BATCH_MFMA=0 mfma4:BARK_HIP_BATCH_MFMA=2 plainln:BARK_HIP_LN_ROWS_PLAIN=1 > gpurun_out/check_batch_small.txt 2>&1; tail -c 800 gpurun_out/check_batch_small.txt; echo
timeout 300 python tools/batch_ab.py valu:BARK_HIP_BATCH_MFMA=0 mfma4:BARK_HIP_BATCH_MFMA=2 > gpurun_out/batch_ab.txt 2>&1; cat gpurun_out/batch_ab.txt
timeout 300 python tools/check_routes.py fast small > gpurun_out/check_fast_small.txt 2>&1; tail -c 2500 gpurun_out/check_fast_small.txt; echo
BARK_HIP_BATCH_MFMA=2 timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "larger_lock_step or unequal_lengths or lock_step_batch_with_temp or lock_step_products or stage_loops_mini or small_model_decode or fine_eval" 2>&1 | tail -4
