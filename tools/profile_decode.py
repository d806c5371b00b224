#!/usr/bin/env python3
"""Small driver for rocprofv3 passes: a few hundred launches of each decode GEMV + some decode steps + one fine pass."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
pkg = load_package()
ctx = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=32), 0)
for op in range(4):
    print("gemv op", op, ctx.time_gemv(0, op, 240))
print("decode step", ctx.time_decode_step(0, 640, 50))
print("fine pass", ctx.time_fine_pass(1))
ctx.free()
