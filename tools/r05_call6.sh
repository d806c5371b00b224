#!/bin/bash
# round 5, GPU call 6 (no csrc change since the suite run of the final call): the forced-race GPU test added afterwards, the bench line with the round's own
# PMC summary in place, one more soak of cloned contexts
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -x -q -k "test_concurrent_dispatch_race or test_cloned_contexts_serve_jobs or test_few_slot_route" > gpurun_out/r05_c6_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05_c6_pytest.log; tail -5 gpurun_out/r05_c6_pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r05_bench_small_n1.json 2> gpurun_out/r05_bench.err; tail -c 400 gpurun_out/r05_bench_small_n1.json; echo
( timeout 200 python tools/clone_stress.py 100000 mini 8 3 150 ) > gpurun_out/r05_clone_stress_final_8_threads_b.txt 2>&1; grep "clone_stress:\| coarse: \| semantic: \|errors" gpurun_out/r05_clone_stress_final_8_threads_b.txt | cut -c1-250 | tail -5
