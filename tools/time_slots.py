#!/usr/bin/env python3
"""Per-kernel device times of the lock-step decode path (bark_hip_time_slots): every product of a layer by route, the LayerNorm rows kernel
and the all-slots attention.  usage: time_slots.py [preset] [B] [kinds, e.g. 0,1,2] [ctx]   -> gpurun_out/time_slots.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
preset = sys.argv[1] if len(sys.argv) > 1 else "small"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
kinds = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,1,2").split(",")]
ctxlen = int(sys.argv[4]) if len(sys.argv) > 4 else 640
pkg = load_package()
ctx = pkg.BarkContext.load_model(ensure_model(preset, 0), pkg.default_params(temp=0.0, fine_temp=0.0), 0)
names = ["qkv", "proj", "fc_gelu", "mproj"]
out = {"preset": preset, "B": B, "ctx": ctxlen, "us_per_launch": {}}
for kind in kinds:
    out["us_per_launch"]["kind%d" % kind] = {n: round(ctx.time_slots(0, op, B, kind, ctxlen, 960), 2) for op, n in enumerate(names)}
out["us_per_launch"]["ln_fused_products"] = {n: round(ctx.time_slots(0, op, B, 6, ctxlen, 960), 2) for op, n in ((0, "qkv"), (2, "fc_gelu"))}
out["us_per_launch"]["ln_rows"] = round(ctx.time_slots(0, 4, B, 0, ctxlen, 960), 2)
out["us_per_launch"]["attention_all_slots_one_workgroup_per_head_and_slot"] = round(ctx.time_slots(0, 5, B, 0, ctxlen, 480), 2)
out["us_per_launch"]["attention_all_slots"] = round(ctx.time_slots(0, 5, B, 1, ctxlen, 480), 2)       # scores + mix launches (two kernels per call)
ctx.free()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "time_slots.json"), "w"), indent=1)
print(json.dumps(out))
