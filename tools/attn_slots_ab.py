#!/usr/bin/env python3
"""Lock-step decode attention per call by slot count: one workgroup per (head, slot) [VS = 1 / 2 by the launch rule] against the scores + mix pair."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
pkg = load_package()
ctxlen = int(sys.argv[1]) if len(sys.argv) > 1 else 640
for B in ([int(v) for v in sys.argv[2:]] or [8, 12, 16, 21, 24, 32, 42, 48, 64]):
    ctx = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0), 0)
    ctx.reserve_batch(max(B, 8))
    fused = ctx.time_slots(0, 5, B, 0, ctxlen, 480)
    pair = ctx.time_slots(0, 5, B, 1, ctxlen, 480)
    print(f"B={B:2d} ctx={ctxlen}: fused (one launch) {fused:6.2f} us   launch rule with the score buffer {pair:6.2f} us", flush=True)
    ctx.free()
