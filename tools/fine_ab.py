#!/usr/bin/env python3
"""Fine forward pass (N = 1024) under environment variants, each in a fresh process: fine_ab.py NAME[:K=V,...] ...  -> us per pass (FINE_WINDOWS=Z: Z windows side by side, us per window).
BARK_HIP_ATTN_DBG bits skip phases of attn_rows_kernel (1 scores, 2 exp, 4 mix; results are wrong, timing only)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
pkg = load_package()
ctx = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0), 0)
import os
Z = int(os.environ.get("FINE_WINDOWS", "1"))
us, flops = ctx.time_fine_pass(12 if Z == 1 else 6, Z)
us /= Z
print("RESULT", round(us, 1))
ctx.free()
''' % ROOT
out = {}
for a in sys.argv[1:] or ["base"]:
    name, _, kv = a.partition(":")
    e = dict(os.environ); e.update(dict(x.split("=") for x in kv.split(",") if x))
    p = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    out[name] = float(line[0].split()[1]) if line else p.stderr[-300:]
    print(name, out[name], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fine_ab.json"), "w"), indent=1)
