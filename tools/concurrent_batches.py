#!/usr/bin/env python3
"""N prompts on ONE GPU as G concurrent lock-step batches of N / G slots (cloned contexts sharing the weight slab, one host thread and one
stream each) against one batch of N: a lock step is a chain of ~100 small dependent kernels, so two chains can share the chip.
  python tools/concurrent_batches.py [N ...]   -> prompts/s per (N, G)"""
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
for N in [int(v) for v in sys.argv[1:]] or [8, 16, 32]:
    for G in (1, 2, 4):
        if N // G < 2:
            continue
        ctxs = [base.clone(i + 1) for i in range(G)]
        idx = sorted(range(N), key=lambda i: (len(prompts[i]), i))
        groups = [[prompts[i] for i in idx[g::G]] for g in range(G)]
        def run(g):
            ctxs[g].generate_batch(groups[g])
        for rep in range(2):                                  # first pass: warm-up (graph capture, allocations)
            th = [threading.Thread(target=run, args=(g,)) for g in range(G)]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            dt = time.perf_counter() - t0
        out["N%d_G%d" % (N, G)] = round(N / dt, 2)
        print("N", N, "groups", G, "slots per group", N // G, "prompts/s", round(N / dt, 2), flush=True)
        for c in ctxs: c.free()
print(json.dumps(out))
