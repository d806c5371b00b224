#!/usr/bin/env python3
"""A/B of process-level runtime switches on the decode step (each arm in a fresh process: the HIP runtime reads them once)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json
sys.path.insert(0, %r)
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
pkg = load_package()
ctx = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=32), 0)
r = {"decode_step_us": ctx.time_decode_step(0, 640, 400)[0]}
for op, nm in enumerate(("ln_qkv", "proj", "ln_fc", "mproj")):
    r[nm] = ctx.time_gemv(0, op, 960)[0]
r["attn"] = ctx.time_gemv(0, 12, 960)[0]
print("RESULT", json.dumps(r))
ctx.free()
''' % ROOT
arms = [("default", {}), ("HIP_FORCE_DEV_KERNARG=1", {"HIP_FORCE_DEV_KERNARG": "1"}), ("HIP_FORCE_DEV_KERNARG=0", {"HIP_FORCE_DEV_KERNARG": "0"}),
        ("DEBUG_HIP_GRAPH_DOT_PRINT=0 GPU_MAX_HW_QUEUES=1", {"GPU_MAX_HW_QUEUES": "1"})]
extra = sys.argv[1:]
for a in extra:                                      # further arms: NAME=VALUE[,NAME=VALUE]
    arms.append((a, dict(kv.split("=") for kv in a.split(","))))
out = {}
for name, env in arms:
    e = dict(os.environ); e.update(env)
    p = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    out[name] = json.loads(line[0][7:]) if line else {"error": p.stderr[-400:]}
    print(name, out[name], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "kernarg_ab.json"), "w"), indent=1)
