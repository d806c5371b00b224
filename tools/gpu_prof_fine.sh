#!/bin/bash
# rocprofv3 kernel statistics of the fine forward pass on the canonical route (products on the f16 matrix cores), one window and eight side by side
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
bash tools/run_prof_fine.sh c1m c1m_x8:FINE_WINDOWS=8 > gpurun_out/${1:-r04}_fine_kernel_stats.txt 2>&1
cat gpurun_out/${1:-r04}_fine_kernel_stats.txt
python tools/time_slots.py small 8 0 640 | tail -1
BARK_HIP_CROSSCHECK=32 python tools/time_slots.py small 8 0 640 | tail -1
