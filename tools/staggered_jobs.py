#!/usr/bin/env python3
"""Lock-step jobs of S slots on ONE GPU from G host threads (cloned contexts sharing the weight slab, one stream each), thread g starting
g * stagger ms late, J jobs per thread back to back: does the matrix-core-heavy tail of one job (fine passes, codec) hide under the
latency-bound decode chain of another one?  Against the same prompts as jobs of G * S slots from one thread.
  python tools/staggered_jobs.py S:G:J:stagger_ms [...]   -> prompts/s per spec (the first pass of every thread is a warm-up)"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
import bench
pkg = load_package()
base = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=256), 0)
prompts = bench.synth_prompts(64)
out = {}
for spec in sys.argv[1:] or ["64:1:2:0", "32:2:2:600", "32:2:2:0"]:
    S, G, J, stagger = [int(v) for v in spec.split(":")]
    ctxs = [base.clone(i + 1) for i in range(G)]
    jobs = [[prompts[(g * S + i) % 64] for i in range(S)] for g in range(G)]
    def run(g, reps):
        time.sleep(g * stagger / 1000.0)
        for _ in range(reps):
            ctxs[g].generate_batch(jobs[g])
    for reps in (1, J):                                       # first pass: warm-up (graph capture, allocations)
        th = [threading.Thread(target=run, args=(g, reps)) for g in range(G)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
    out[spec] = round(S * G * J / dt, 2)
    print("slots", S, "threads", G, "jobs per thread", J, "stagger ms", stagger, "wall s", round(dt, 3), "prompts/s", out[spec], flush=True)
    for c in ctxs: c.free()
print(json.dumps(out))
