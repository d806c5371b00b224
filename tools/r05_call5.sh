#!/bin/bash
# round 5, GPU call 5: the state race of the sampler forced on both diagnostic builds; the restored per-slot few-slot route: parity subset + lock-step times
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r05_c5
( timeout 120 python tools/state_race_demo.py lag_old; timeout 120 python tools/state_race_demo.py lag_fix ) > ${O}_state_race_demo.txt 2>&1; grep -v "^bark-mi355x" ${O}_state_race_demo.txt | tail -12
timeout 300 python -m pytest tests -m gpu -x -q -k "test_few_slot_route or test_in_engine_batch or test_larger_lock_step_batches or test_batch_with_unequal_lengths or test_small_ragged_job or test_cross_check_routes or test_job_larger_than_the_slots or test_stage_loops_toy or test_cloned_contexts_serve" > ${O}_pytest.log 2>&1; echo "pytest rc $?" >> ${O}_pytest.log; tail -6 ${O}_pytest.log
timeout 300 python tools/r05_sweep.py part2 > ${O}_sweep_part2.txt 2>&1; cat ${O}_sweep_part2.txt | cut -c1-900
