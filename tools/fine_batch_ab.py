import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
import bench
pkg = load_package()
prompts = bench.synth_prompts(64)
ctx = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=256), 0)
ctx.generate_batch(prompts)
t0 = time.perf_counter(); ctx.generate_batch(prompts); dt = time.perf_counter() - t0
st = ctx.stats()
print("BARK_HIP_FINE_BATCH=%s: 64-prompt job %.1f ms, fine %.1f ms" % (os.environ.get("BARK_HIP_FINE_BATCH", "8 (default)"), dt * 1e3, st["t_fine_us"] / 1e3), flush=True)
ctx.free()
