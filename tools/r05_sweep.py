#!/usr/bin/env python3
"""Round 5: one measurement of every opt-in switch that round 4 left unmeasured, so that each can be adopted or deleted.

  part 1 (single-utterance decode step): the BARK_HIP_WPREFETCH arms of round 4 - measured in round 5 (every arm 10 - 17 % SLOWER than the default,
          profiles/r05_gpu_suite_reordered_first_run.log) and deleted with the experiment; the part is kept as the decode-step timer of the default build.
  part 2 (lock steps at few slots): BARK_HIP_FEW_SLOTS arms - graph-replayed lock step (bark_hip_profile_lock_step) at 2 .. 32 live slots for both
          causal models, per-site times at 4 / 8 / 16 slots.
Every arm is a process of its own (the switches are read once per process).   python tools/r05_sweep.py [part1|part2|all]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD1 = r'''
import sys, json, hashlib
sys.path.insert(0, %r)
import numpy as np
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
pkg = load_package()
ctx = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=48), 0)
r = {}
for which, nm in ((0, "sem"), (1, "coarse")):
    for cl in (300, 640, 1000):
        r["%%s@%%d" %% (nm, cl)] = round(ctx.time_decode_step(which, cl, 400)[0], 2)
h = hashlib.sha256()
assert ctx.generate_audio("the quick brown fox jumps over the lazy dog")
for a in (ctx.semantic_tokens(), ctx.coarse_tokens(), ctx.fine_tokens(), ctx.audio_data()):
    h.update(np.ascontiguousarray(a).tobytes())
r["sha"] = h.hexdigest()[:16]
print("RESULT", json.dumps(r))
ctx.free()
''' % ROOT
CHILD2 = r'''
import sys, json
sys.path.insert(0, %r)
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
pkg = load_package()
ctx = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0), 0)
ctx.reserve_batch(32)
r = {}
for which, nm in ((1, "coarse"), (0, "sem")):
    for B in (2, 4, 6, 8, 12, 16, 24, 32):
        tl = ctx.profile_lock_step(which, B, 640, 12)
        r["%%s_B%%d" %% (nm, B)] = round(tl[-1]["us"], 1)
        if B in (4, 8, 16) and which == 1:
            sites = {}
            for e in tl[:-1]:
                sites[e["site"]] = sites.get(e["site"], 0.0) + e["us"]
            r["sites_B%%d" %% B] = {k: round(v, 1) for k, v in sites.items()}
print("RESULT", json.dumps(r))
ctx.free()
''' % ROOT


def run(child, env_add):
    env = dict(os.environ); env.update(env_add)
    p = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=240)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    return json.loads(line[0][7:]) if (p.returncode == 0 and line) else {"error": "rc %d: %s" % (p.returncode, p.stderr[-400:])}


what = sys.argv[1] if len(sys.argv) > 1 else "all"
out = {}
if what in ("part1", "all"):
    arms = {"default": {}, "default_again": {}}
    for name, env in arms.items():
        out[name] = run(CHILD1, env)
        print("decode", name, json.dumps(out[name]), flush=True)
    ref = out["default"].get("sha")
    print("bits equal to the default arm:", {k: v.get("sha") == ref for k, v in out.items()}, flush=True)
if what in ("part2", "all"):
    arms = {"matrix_core_route": {"BARK_HIP_FEW_SLOTS": "0"}, "default_products_16_scores_8": {}, "products_16_scores_16": {"BARK_HIP_FEW_SLOTS": "16,16"},
            "products_32_scores_8": {"BARK_HIP_FEW_SLOTS": "32,8"}}
    for name, env in arms.items():
        out["slots_" + name] = run(CHILD2, env)
        print("lock step", name, json.dumps(out["slots_" + name]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_sweep_%s.json" % what), "w"), indent=1)
