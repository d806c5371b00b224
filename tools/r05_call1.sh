#!/bin/bash
# round 5, GPU call 1: root cause of round 4's red suite (read-back arms), then the whole suite in its new order (no -x), plain and poisoned
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r05_c1
( BARK_HIP_READBACK=legacy timeout 240 python tools/clone_stress.py 12 ) > ${O}_stress_legacy.txt 2>&1; echo "rc $?" >> ${O}_stress_legacy.txt
( BARK_HIP_READBACK=legacy BARK_HIP_READBACK_CHECK=1 timeout 240 python tools/clone_stress.py 12 ) > ${O}_stress_legacy_check.txt 2>&1; echo "rc $?" >> ${O}_stress_legacy_check.txt
( BARK_HIP_READBACK=pinned BARK_HIP_READBACK_CHECK=1 timeout 240 python tools/clone_stress.py 12 ) > ${O}_stress_pinned_check.txt 2>&1; echo "rc $?" >> ${O}_stress_pinned_check.txt
( timeout 240 python tools/clone_stress.py 12 ) > ${O}_stress_default.txt 2>&1; echo "rc $?" >> ${O}_stress_default.txt
( BARK_HIP_READBACK_CHECK=1 timeout 240 python tools/clone_stress.py 12 ) > ${O}_stress_default_check.txt 2>&1; echo "rc $?" >> ${O}_stress_default_check.txt
for f in ${O}_stress_*.txt; do echo "== $f"; grep -c "MISMATCH" $f; grep "differ\|rc \|errors" $f | head -12; done
timeout 1200 python -m pytest tests -m gpu -q --durations=40 > ${O}_pytest.log 2>&1; echo "pytest rc $?" >> ${O}_pytest.log
tail -60 ${O}_pytest.log
BARK_HIP_POISON=1 timeout 900 python -m pytest tests -m gpu -q > ${O}_pytest_poison.log 2>&1; echo "pytest rc $?" >> ${O}_pytest_poison.log
tail -30 ${O}_pytest_poison.log
timeout 600 python bench.py --steps 3 --warmup 1 > ${O}_bench.json 2> ${O}_bench.err; tail -c 600 ${O}_bench.json
