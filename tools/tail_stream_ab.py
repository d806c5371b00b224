#!/usr/bin/env python3
"""The second stream of a lock-step job (engine_batch.hip: JobTail - fine passes and codec of the utterances that have left the coarse stage,
beside the decode chain of the others) against the one-stream form, on jobs where utterances leave at different times:
  ragged64   config 5's 64 prompts with step caps 64 .. 256 by prompt length (bench.ragged_caps), 64 slots
  equal128   128 utterances of 256 steps on 64 slots (the second half enters as the first leaves)
  python tools/tail_stream_ab.py [name:ENV=V,ENV=V ...]     one process per arm; prompts/s and the stage clocks of the job's context"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def arm():
    from bark_amd_loader import load_package
    from tools.make_synth_model import ensure_model
    import bench
    pkg = load_package()
    prompts = bench.synth_prompts(64)
    ctx = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=256), 0)
    ctx.reserve_batch(64)
    caps = bench.ragged_caps(prompts)
    jobs = {"ragged64": (prompts, [ctx.request_params(n_steps_text_encoder=caps[i]) for i in range(64)]),
            "equal128": (prompts + prompts, None)}
    out = {}
    for name, (texts, reqs) in jobs.items():
        best = None
        for rep in range(3):                                   # the first pass warms up (graph captures, the clone)
            t0 = time.perf_counter(); res = ctx.generate_batch(texts, params=reqs); dt = time.perf_counter() - t0
            if rep and (best is None or dt < best[0]): best = (dt, ctx.stats())
        dt, st = best
        out[name] = {"prompts_per_s": round(len(texts) / dt, 2), "wall_ms": round(dt * 1e3, 1),
                     "stage_ms": {k: round(st["t_%s_us" % k] / 1e3) for k in ("semantic", "coarse", "fine", "codec")}}
    print("ARM " + json.dumps(out), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--arm":
        arm(); sys.exit(0)
    for spec in sys.argv[1:] or ["one_stream:BARK_HIP_TAIL_STREAM=0", "second_stream"]:
        name, _, envs = spec.partition(":")
        env = dict(os.environ)
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("="); env[k] = v
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--arm"], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("ARM ")]
        print(name, line[0][4:] if line else "FAILED " + r.stderr[-400:], flush=True)
