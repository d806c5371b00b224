#!/usr/bin/env python3
"""The request collector under load: R requests (config 5's prompts, 256 semantic steps, greedy) submitted at once from 16 host threads,
served by 1 or 2 job streams (bark_hip_batcher_create_ex) in jobs of up to 64: requests / s from the first submit to the last result.
  python tools/batcher_load.py [R] [streams ...]"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
import bench
pkg = load_package()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prompts = bench.synth_prompts(64)
out = {}
for streams in [int(v) for v in sys.argv[2:]] or [1, 2]:
    ctx = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=256), 0)
    b = pkg.Batcher(ctx, max_batch=64, max_wait_ms=20, streams=streams)
    def client(k, n):
        ts = [b.submit(prompts[i % 64]) for i in range(k, n, 16)]
        for t in ts: b.wait(t)
    for n in (64 * streams, R):                               # first pass: warm-up (graph captures, clones)
        th = [threading.Thread(target=client, args=(k, n)) for k in range(16)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
    st = b.stats()
    out[streams] = round(R / dt, 2)
    print("streams", streams, "requests", R, "wall s", round(dt, 3), "requests/s", out[streams], st, flush=True)
    b.free(); ctx.free()
print(json.dumps(out))
