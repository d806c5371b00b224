#!/usr/bin/env python3
"""Config 5's per-GPU share at N = 8 (8 prompts per rank): ONE lock-step job on 8 slots against TWO concurrent jobs of 4 on clones of the context (two host
threads, two streams) - a lock step at few slots is a latency chain that leaves most of the chip idle, so two chains may overlap.
usage: python tools/two_jobs_per_gpu.py [prompts_per_rank=8]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
pkg = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
path = ensure_model("small", 0)
params = pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=256)
prompts = bench.synth_prompts(64)
idx = bench.shard_prompts(prompts, 0, 64 // n)          # rank 0's shard at N = 64 / n GPUs
texts = [prompts[i] for i in idx]

def timed(fn, reps=2):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps, out

one = pkg.BarkContext.load_model(path, params, 0)
dt1, res1 = timed(lambda: one.generate_batch(texts))
print(f"one job of {n} on {n} slots: {dt1 * 1e3:.0f} ms = {n / dt1:.1f} prompts/s", flush=True)
one.free()

for parts in (2, 4):
    if n % parts:
        continue
    base = pkg.BarkContext.load_model(path, params, 0)
    ctxs = [base] + [base.clone(k) for k in range(1, parts)]
    shares = [texts[k::parts] for k in range(parts)]
    def run():
        out = [None] * parts
        def work(k):
            out[k] = ctxs[k].generate_batch(shares[k])
        th = [threading.Thread(target=work, args=(k,)) for k in range(parts)]
        for t in th: t.start()
        for t in th: t.join()
        return out
    dt, res = timed(run)
    ok = all(np.array_equal(res[k][j]["pcm"], res1[k + parts * j]["pcm"]) for k in range(parts) for j in range(len(shares[k])))
    print(f"{parts} concurrent jobs of {n // parts} (clones, one host thread each): {dt * 1e3:.0f} ms = {n / dt:.1f} prompts/s; PCM equal to the one-job run: {ok}", flush=True)
    for c in ctxs[1:]:
        c.free()
    base.free()
