"""A few coarse-model prompt passes of 887 rows (the re-encoded windows of the coarse stage) for a rocprofv3 pass; BARK_HIP_CROSSCHECK=512 keeps attn_rows_kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
pkg = load_package()
ctx = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0), 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 887
tok = (np.arange(N) * 7 % 10000).astype(np.int32)
ctx.gpt_eval(1, tok, 0, False)
t0 = time.perf_counter()
for _ in range(5):
    ctx.gpt_eval(1, tok, 0, False)
print("prompt pass of %d rows: %.2f ms" % (N, (time.perf_counter() - t0) / 5 * 1e3))
ctx.free()
