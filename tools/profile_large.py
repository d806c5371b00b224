"""bark-large shapes (BASELINE config 3) for a rocprofv3 pass: decode steps of the semantic model at context 640, then fine forward passes (default order C1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
pkg = load_package()
ctx = pkg.BarkContext.load_model(ensure_model("large", 0), pkg.default_params(temp=0.0, fine_temp=0.0), 0)
what = sys.argv[1] if len(sys.argv) > 1 else "decode"
if what == "decode":
    print("decode step", ctx.time_decode_step(0, 640, 96))
else:
    print("fine pass", ctx.time_fine_pass(3))
ctx.free()
