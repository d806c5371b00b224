#!/bin/bash
# round 5, GPU call 2: long stress of the read-back arms (the r04 failure did not show in 12 iterations per arm), few-slot route sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r05_c2
( BARK_HIP_READBACK=legacy BARK_HIP_READBACK_CHECK=1 timeout 200 python tools/clone_stress.py 400 ) > ${O}_stress_legacy_check.txt 2>&1; echo "rc $?" >> ${O}_stress_legacy_check.txt
( BARK_HIP_READBACK=legacy timeout 200 python tools/clone_stress.py 400 mini 8 3 ) > ${O}_stress_legacy_8threads.txt 2>&1; echo "rc $?" >> ${O}_stress_legacy_8threads.txt
( timeout 200 python tools/clone_stress.py 400 mini 8 3 ) > ${O}_stress_default_8threads.txt 2>&1; echo "rc $?" >> ${O}_stress_default_8threads.txt
( BARK_HIP_READBACK_CHECK=1 timeout 200 python tools/clone_stress.py 400 ) > ${O}_stress_default_check.txt 2>&1; echo "rc $?" >> ${O}_stress_default_check.txt
for f in ${O}_stress_*.txt; do echo "== $f"; grep -c "MISMATCH" $f; grep "differ at\|rc \|errors\|BASE\|clone_stress:" $f | head -12; tail -2 $f; done
timeout 400 python tools/r05_sweep.py part2 > ${O}_sweep_part2.txt 2>&1; cat ${O}_sweep_part2.txt
