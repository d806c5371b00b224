#!/bin/bash
# round 3, call 8: whole GPU suite on the current build + the bench line with the new legs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c8_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c8_pytest.log
tail -5 gpurun_out/c8_pytest.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/c8_bench.json 2> gpurun_out/c8_bench.err; tail -3 gpurun_out/c8_bench.err
python - <<'P'
import json
d = json.load(open("gpurun_out/c8_bench.json"))
for k in ("value", "roofline_fine_pass", "config5_64_prompts", "tolerance_route", "bark_large", "q4_0"):
    print(k, json.dumps(d.get(k))[:900])
P
