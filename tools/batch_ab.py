#!/usr/bin/env python3
"""A/B of the lock-step batch path (BARK_HIP_CROSSCHECK=2 VALU products, default matrix cores): prompts/s at B = 8 and 32, each arm in a fresh process."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time, json
sys.path.insert(0, %r)
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
import bench
pkg = load_package()
prompts = bench.synth_prompts(64)
out = {}
for B in [int(v) for v in os.environ.get("BATCH_AB_SIZES", "8,32").split(",")]:
    ctx = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=256), 0)
    ctx.generate_batch(prompts[:B])
    t0 = time.perf_counter(); res = ctx.generate_batch(prompts[B:2 * B]); dt = time.perf_counter() - t0
    st = ctx.stats()
    out["B%%d" %% B] = {"prompts_per_s": round(B / dt, 2), "semantic_ms": st["t_semantic_us"] // 1000, "coarse_ms": st["t_coarse_us"] // 1000, "fine_ms": st["t_fine_us"] // 1000, "codec_ms": st["t_codec_us"] // 1000}
    ctx.free()
print("RESULT", json.dumps(out))
''' % ROOT
# arms: NAME[:K=V,K=V...] on the command line; default: the VALU route against the two MFMA routes
arms = []
for a in sys.argv[1:] or ["valu:BARK_HIP_CROSSCHECK=2", "mfma"]:
    name, _, kv = a.partition(":")
    arms.append((name, dict(x.split("=") for x in kv.split(",") if x)))
for name, env in arms:
    e = dict(os.environ); e.update(env)
    p = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    print(name, line[0][7:] if line else p.stderr[-600:], flush=True)
