import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
pkg = load_package()
ctx = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=32), 0)
names = ["ln_qkv", "proj", "ln_fc", "mproj"]
for rep in range(1):
    for op in range(8):
        us, nb = ctx.time_gemv(0, op, 2400)
        print(f"{names[op & 3]:7s} {'hot ' if op >= 4 else 'cold'} {us:6.2f} us  {nb / us / 1e3:7.1f} GB/s")
for op, nm in ((8, "scores"), (9, "mix"), (10, "scores+mix"), (11, "fused"), (12, "split4")):
    print(nm, ctx.time_gemv(0, op, 2400)[0], "us")
print("decode step", ctx.time_decode_step(0, 640, 300))
ctx.free()
