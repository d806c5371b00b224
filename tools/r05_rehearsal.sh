#!/bin/bash
# round 5, last GPU call: the driver's own round-end sequence at HEAD (smoke, then the whole GPU suite with -x)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke_at_head.txt 2>&1; tail -1 gpurun_out/r05_smoke_at_head.txt | cut -c1-200
timeout 700 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r05_gpu_suite_at_head.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05_gpu_suite_at_head.log; tail -9 gpurun_out/r05_gpu_suite_at_head.log
