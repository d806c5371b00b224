#!/bin/bash
# round 3, call 7: kernel statistics of the fine pass with 8 windows side by side, both routes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
F=BARK_HIP_FAST_GEMM=1
timeout 300 python tools/fine_ab.py z1 z8:FINE_WINDOWS=8 z32:FINE_WINDOWS=32 f1:$F f8:$F,FINE_WINDOWS=8 f32:$F,FINE_WINDOWS=32 > gpurun_out/c7_fine_ab.txt 2>&1; cat gpurun_out/c7_fine_ab.txt
bash tools/run_prof_fine.sh exact8:FINE_WINDOWS=8 fast8:$F,FINE_WINDOWS=8 > gpurun_out/c7_prof_fine.txt 2>&1; cat gpurun_out/c7_prof_fine.txt
