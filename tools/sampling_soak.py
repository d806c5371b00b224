#!/usr/bin/env python3
"""Soak of the device multinomial picks: N sampled utterances (temp / fine_temp > 0, own seeds) as lock-step jobs with device sampling against
the same utterances with BARK_HIP_HOST_SAMPLING=1 (std::discrete_distribution on fetched logits, one utterance at a time) - every id of all three
stages must agree.  usage: sampling_soak.py [preset] [N] [cap]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
import bench
pkg = load_package()
preset, N, cap, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
ctx = pkg.BarkContext.load_model(ensure_model(preset, 0), pkg.default_params(), 0)
rng = np.random.default_rng(99)
words = " ".join(bench.synth_prompts(16)).split()
texts = [" ".join(rng.choice(words, size=int(rng.integers(2, 30)))) for _ in range(N)]
reqs = [ctx.request_params(temp=float(rng.choice([0.7, 1.0])), fine_temp=0.5, min_eos_p=0.2, n_steps_text_encoder=int(rng.integers(cap // 2, cap + 1)), seed=1000 + i) for i in range(N)]
d = {}
for k0 in range(0, N, 64):
    res = ctx.generate_batch(texts[k0:k0 + 64], params=reqs[k0:k0 + 64])
    for i, r in enumerate(res):
        for k in ("semantic", "coarse", "fine"):
            d["%%s%%d" %% (k, k0 + i)] = np.asarray(r[k]) if r is not None else np.zeros(0, np.int32)
d["near_tie"] = np.int64(ctx.stats()["n_near_tie"])
np.savez(out, **d)
ctx.free()
''' % ROOT
import numpy as np
preset = sys.argv[1] if len(sys.argv) > 1 else "toy"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 60
outs = []
for host in ("0", "1"):
    with tempfile.NamedTemporaryFile(suffix=".npz", delete=False) as f:
        path = f.name
    env = dict(os.environ); env.pop("BARK_HIP_HOST_SAMPLING", None)
    if host == "1": env["BARK_HIP_HOST_SAMPLING"] = "1"
    r = subprocess.run([sys.executable, "-c", CHILD, preset, str(N), str(cap), path], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    outs.append(np.load(path))
bad = 0; picks = 0
for k in outs[0].files:
    if k == "near_tie": continue
    a, b = outs[0][k], outs[1][k]
    picks += a.size
    if a.shape != b.shape or not np.array_equal(a, b):
        bad += 1; print("MISMATCH", k, a.shape, b.shape, int((a.ravel()[:min(a.size, b.size)] != b.ravel()[:min(a.size, b.size)]).sum()) if a.size and b.size else -1)
print(f"{preset}: {N} sampled utterances, {picks} ids, {bad} arrays differ between device and host sampling; picks of the last job settled by the exact path: {int(outs[0]['near_tie'])}")
sys.exit(1 if bad else 0)
