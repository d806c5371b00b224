#!/usr/bin/env python3
"""A/B of decode-step variants selected by environment switches / alternate library builds: every arm runs in a fresh process,
checks parity on a few GPU tests and times the decode step at several context lengths.

  python tools/decode_ab.py NAME[:ENV=VAL,ENV=VAL...] ...      (first arm should be the baseline)
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json
sys.path.insert(0, %r)
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
pkg = load_package()
ctx = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=32), 0)
r = {}
for cl in (300, 640, 1000):
    r["step@%%d" %% cl] = round(ctx.time_decode_step(0, cl, 300)[0], 2)
for op, nm in enumerate(("ln_qkv", "proj", "ln_fc", "mproj")):
    r[nm] = round(ctx.time_gemv(0, op, 960)[0], 2)
print("RESULT", json.dumps(r))
ctx.free()
''' % ROOT
TESTS = "test_q4_0_stage_loops_toy or test_stage_loops_toy or test_exact_sampling_path or test_semantic_eval_prefill_and_decode or test_coarse_eval_prefill_and_decode or test_stage_loops_mini or test_graph_and_eager_agree or test_small_model_decode_and_stages"
out = {}
for arm in sys.argv[1:]:
    name, _, envs = arm.partition(":")
    env = dict(os.environ)
    for kv in filter(None, envs.split(",")):
        k, v = kv.split("=", 1)
        env[k] = v
    t = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-x", "-q", "-m", "gpu", "-k", TESTS], env=env, cwd=ROOT, capture_output=True, text=True)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    out[name] = {"parity": t.stdout.strip().splitlines()[-1] if t.stdout.strip() else t.stderr[-300:], "times_us": json.loads(line[0][7:]) if line else {"error": p.stderr[-400:]}}
    print(name, json.dumps(out[name]), flush=True)
    if t.returncode != 0:
        print(t.stdout[-1500:], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "decode_ab.json"), "w"), indent=1)
