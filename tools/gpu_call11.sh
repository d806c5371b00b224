#!/bin/bash
# round 3, call 11: whole GPU suite + the round's evidence on the current build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c11_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c11_pytest.log
tail -4 gpurun_out/c11_pytest.log
bash tools/collect_profiles.sh r03 > gpurun_out/c11_collect.log 2>&1
tail -12 gpurun_out/c11_collect.log
