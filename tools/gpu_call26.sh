#!/bin/bash
# round 3, call 26: last check of the routes touched after the final full-suite run (call 18) and the bench line of the final build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 240 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tolerance or cross_check or config5 or server or batcher or first_slot or unequal or in_engine_batch" > gpurun_out/c26_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c26_pytest.log
tail -3 gpurun_out/c26_pytest.log
timeout 400 python bench.py --steps 3 --warmup 1 > gpurun_out/r03_bench_small_n1.json 2> gpurun_out/r03_bench.err
python -c "import json; d=json.load(open('gpurun_out/r03_bench_small_n1.json')); print(d['value'], d['config5_64_prompts']['prompts_per_s'], d['tolerance_route']['fine_pass']['us_per_pass'], d['tolerance_route']['fine_pass']['eight_windows_side_by_side'], d['tolerance_route']['config5_64_prompts']['prompts_per_s'], d['tolerance_route']['rtf'])"
