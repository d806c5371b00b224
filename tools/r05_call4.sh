#!/bin/bash
# round 5, GPU call 4: the few-slot route (slot-group kernels) - parity, lock-step times per route, soaks of the default build under concurrent contexts
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r05_c4
timeout 400 python -m pytest tests -m gpu -x -q -k "test_few_slot_route or test_in_engine_batch or test_larger_lock_step_batches or test_batch_with_unequal_lengths or test_small_ragged_job or test_cross_check_routes or test_job_larger_than_the_slots or test_lock_step_batch_whose_first_slot or test_q4_0_generate_and_lock_step or test_lock_step_batch_with_temperature" > ${O}_pytest.log 2>&1; echo "pytest rc $?" >> ${O}_pytest.log; tail -15 ${O}_pytest.log
timeout 300 python tools/r05_sweep.py part2 > ${O}_sweep_part2.txt 2>&1; cat ${O}_sweep_part2.txt
run() { name=$1; g=$2; secs=$3; shift 3; ( env "$@" timeout $((secs + 45)) python tools/clone_stress.py 100000 mini $g 3 $secs ) > ${O}_stress_$name.txt 2>&1; echo "rc $?" >> ${O}_stress_$name.txt; echo "== $name"; grep -c " coarse: \| semantic: " ${O}_stress_$name.txt; grep "clone_stress:\|errors\| coarse: \| semantic: " ${O}_stress_$name.txt | cut -c1-300 | head -8; }
run default_8_threads 8 150 BARK_HIP_GUARD=0
run old_routes_8_threads 8 100 BARK_HIP_FEW_SLOTS=0
