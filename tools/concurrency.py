"""Throughput of B utterances in flight on ONE GPU (cloned contexts sharing the weight slab, one stream + host thread each)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
import bench
pkg = load_package()
n_sem = int(sys.argv[1]) if len(sys.argv) > 1 else 256
base = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=n_sem), 0)
prompts = bench.synth_prompts(64)
ctxs = [base]
for B in (1, 2, 4, 8, 16):
    while len(ctxs) < B:
        ctxs.append(base.clone(len(ctxs)))
    pkg.BarkContext.generate_audio_batch(ctxs[:B], prompts[:B])          # warm-up (graph capture)
    t0 = time.perf_counter()
    ok = pkg.BarkContext.generate_audio_batch(ctxs[:B], prompts[B:2 * B])
    dt = time.perf_counter() - t0
    audio = sum(c.stats()["n_samples"] for c in ctxs[:B]) / 24000.0
    print(f"B={B:2d} ok={ok} wall={dt * 1e3:8.1f} ms  prompts/s={B / dt:6.2f}  audio-s/s={audio / dt:7.2f}", flush=True)
