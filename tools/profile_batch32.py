"""rocprofv3 driver: one lock-step batch of 32 utterances, 32 semantic steps (kernel-trace of the batched decode step)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
import bench
pkg = load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=32), 0)
ctx.generate_batch(bench.synth_prompts(64)[:B])
print(ctx.stats())
ctx.free()
