#!/usr/bin/env python3
"""Seeded lock-step batch at the default parameters (temp 0.7 / 0.5, 768 steps: more than 1024 frames, truncated coarse history) against the oracle,
stage by stage; arms = environment variants in fresh processes.   dbg_long_temp_batch.py NAME[:K=V,...] ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
from oracle.pyoracle import Oracle
pkg = load_package()
path = ensure_model("toy", 0)
texts = ["request number %%d about water and time" %% i for i in range(6)]
seeds = [40 + i for i in range(6)]
n = int(sys.argv[1])
c = pkg.BarkContext.load_model(path, pkg.default_params(), seed=1)
res = c.generate_batch(texts[:n], seeds=seeds[:n])
orc = Oracle(path, n_threads=8)
out = []
for i in range(n):
    orc.seed(seeds[i]); ref = orc.generate(texts[i], orc.params(temp=0.7, fine_temp=0.5))
    r = res[i]; row = {}
    for k in ("semantic", "coarse", "fine"):
        a, b = np.asarray(r[k]).ravel(), np.asarray(ref[k]).ravel()
        m = min(len(a), len(b)); d = np.flatnonzero(a[:m] != b[:m])
        row[k] = "ok %%d" %% len(b) if len(a) == len(b) and not len(d) else "len %%d/%%d first diff %%s" %% (len(a), len(b), d[0] if len(d) else None)
    out.append(row)
print("RESULT", out)
''' % ROOT
for a in sys.argv[1:] or ["default"]:
    name, _, kv = a.partition(":")
    e = dict(os.environ); e.update(dict(x.split("=") for x in kv.split(",") if x and not x.startswith("N")))
    n = [x[1:] for x in kv.split(",") if x.startswith("N")]
    p = subprocess.run([sys.executable, "-c", CHILD, n[0] if n else "6"], env=e, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    print(name, line[0][7:] if line else p.stderr[-800:], flush=True)
