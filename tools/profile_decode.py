#!/usr/bin/env python3
"""Small driver for rocprofv3 passes: decode steps only (argv[1]: f16 | q4_0 | ..., argv[2]: number of steps), so that the kernel
statistics of a pass are the kernels of the decode step and nothing else; argv[3]: context length of the steps (default 640)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
pkg = load_package()
fmt = sys.argv[1] if len(sys.argv) > 1 else "f16"
path = ensure_model("small", 0)
if fmt != "f16":
    q = path[:-4] + "_%s.bin" % fmt
    if not os.path.exists(q):
        assert pkg.load_library().bark_model_quantize(path.encode(), q.encode(), {"q4_0": 2, "q4_1": 3, "q8_0": 7, "q5_0": 8, "q5_1": 9}[fmt])
    path = q
ctx = pkg.BarkContext.load_model(path, pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=32), 0)
print("decode step", ctx.time_decode_step(0, int(sys.argv[3]) if len(sys.argv) > 3 else 640, int(sys.argv[2]) if len(sys.argv) > 2 else 200))
ctx.free()
