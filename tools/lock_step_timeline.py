#!/usr/bin/env python3
"""Time line of one lock step of bark_hip_generate_batch (bark_hip_profile_lock_step: a HIP event behind every launch site of the step, the
step enqueued eagerly; kernel time + the gap in front of it), per slot count, summed per site over the layers, beside the graph-replayed
step.  The profiler view of the lock-step path (rocprofv3 cannot follow bark_hip_generate_batch).
  python tools/lock_step_timeline.py [preset] [ctx] -> gpurun_out/lock_step_timeline.json + a table"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
pkg = load_package()
preset = sys.argv[1] if len(sys.argv) > 1 else "small"
ctxlen = int(sys.argv[2]) if len(sys.argv) > 2 else 640
out = {}
for which, name in ((1, "coarse"), (0, "semantic")):
    for B in (8, 16, 32, 64):
        ctx = pkg.BarkContext.load_model(ensure_model(preset, 0), pkg.default_params(temp=0.0, fine_temp=0.0), 0)
        ctx.reserve_batch(B)
        tl = ctx.profile_lock_step(which, B, ctxlen, 20)
        ctx.free()
        sites = {}
        for e in tl[:-1]:
            sites[e["site"]] = sites.get(e["site"], 0.0) + e["us"]
        n_layer = sum(1 for e in tl if e["site"] == "attention")
        row = {"per_layer_us": {k: v / n_layer for k, v in sites.items() if k not in ("lnf+lm_head", "sample+embed")},
               "lm_head_us": sites.get("lnf+lm_head"), "sample_us": sites.get("sample+embed"),
               "eager_step_us": sum(sites.values()), "graph_step_us": tl[-1]["us"]}
        out[f"{name}_B{B}"] = row
        pl = row["per_layer_us"]
        print(f"{name:8s} B={B:2d} ctx={ctxlen}: per layer " + "  ".join(f"{k} {v:5.1f}" for k, v in pl.items())
              + f"  | lm head {row['lm_head_us']:.1f} sample {row['sample_us']:.1f} | eager step {row['eager_step_us']:.0f} us, graph step {row['graph_step_us']:.0f} us", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/lock_step_timeline.json", "w"), indent=1)
