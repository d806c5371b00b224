#!/bin/bash
# Opt-in variants written at the end of round 2 without device time left to run them.  One gpurun call (about 3 minutes) decides each:
#   1. lock-step products, route 3 (4x4x1 MFMA kernel with the conversions hoisted out of the MFMA runs) and route 5 (16x16x4 MFMA,
#      16-row x 16-slot tiles, chains over the waves of a workgroup) against routes 0 / 2:
#      device us per launch, then bit-equality of a whole 32-slot batch against the VALU route
#   2. exact GEMM with hoisted conversions (BARK_HIP_FAST_GEMM=2) and prefill / fine attention with the score MFMAs round-robin over
#      the four C2 accumulators (BARK_HIP_ATTN_DBG=8) against the defaults: fine pass time, then the parity tests with both switched on
# Adopt a variant only if it is faster AND its equality / parity line says so.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out
# 0. can the kernel boundaries of the decode chain be hidden? (overlap_probe.hip said no with fences: 26.7 us against 2.99 us per kernel;
#    overlap_probe2.hip uses the fence-free granule hand-off of the programming guide, three variants, bounded waits)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/overlap_probe2 tools/probes/overlap_probe2.hip 2>/dev/null && timeout 30 /tmp/overlap_probe2 | tee gpurun_out/pending_overlap_probe2.txt
timeout 100 python tools/time_slots.py small 32 0,2,3,5 640 2>&1 | tail -1 | tee gpurun_out/pending_time_slots_b32.txt
timeout 100 python tools/time_slots.py small 8 0,2,3,5 640 2>&1 | tail -1 | tee gpurun_out/pending_time_slots_b8.txt
timeout 60 python tools/check_routes.py batch toy 17 32 valu:BARK_HIP_BATCH_MFMA=0 route3:BARK_HIP_BATCH_MFMA=3 route5:BARK_HIP_BATCH_MFMA=5 2>&1 | tail -1 | tee gpurun_out/pending_check_batch_toy.txt
timeout 240 python tools/check_routes.py batch small 32 48 valu:BARK_HIP_BATCH_MFMA=0 route2:BARK_HIP_BATCH_MFMA=2 route3:BARK_HIP_BATCH_MFMA=3 route5:BARK_HIP_BATCH_MFMA=5 2>&1 | tail -1 | tee gpurun_out/pending_check_batch.txt
timeout 150 python tools/fine_ab.py base hoist:BARK_HIP_FAST_GEMM=2 ilv:BARK_HIP_ATTN_DBG=8 both:BARK_HIP_FAST_GEMM=2,BARK_HIP_ATTN_DBG=8 base2 2>&1 | tee gpurun_out/pending_fine_ab.txt
BARK_HIP_FAST_GEMM=2 BARK_HIP_ATTN_DBG=8 timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mfma_gemm_matches or fine_eval or semantic_eval_prefill or coarse_prefill_ragged or small_model_decode" 2>&1 | tail -2 | tee gpurun_out/pending_parity_hoist.txt
