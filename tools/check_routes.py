#!/usr/bin/env python3
"""Route checks that need a GPU but no oracle: an alternative route of the engine (selected by environment variables, which the library
reads once per process) against the default route, each in a fresh process.

  python tools/check_routes.py batch <preset> <B> <n_semantic> NAME[:K=V,...] ...
      bark_hip_generate_batch of B synthetic prompts; every arm's semantic / coarse / fine ids and PCM must equal arm 0's bit for bit.
  python tools/check_routes.py fast <preset>
      BARK_HIP_FAST_GEMM=1 (f16 matrix cores, hardware accumulation order) against the canonical route: prefill and fine logits
      (max abs difference, stated tolerance), greedy id agreement of one generation, fine-pass time of both routes.

Writes gpurun_out/check_routes_<mode>.json and prints it; exit code 1 when an equality / tolerance check fails."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BATCH_CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
import bench
pkg = load_package()
preset, B, n_sem, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
ctx = pkg.BarkContext.load_model(ensure_model(preset, 0), pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=n_sem), 0)
t0 = time.perf_counter(); res = ctx.generate_batch(bench.synth_prompts(B)); dt = time.perf_counter() - t0
st = ctx.stats()
d = {"dt": np.float64(dt), "semantic_ms": np.float64(st["t_semantic_us"] / 1e3), "coarse_ms": np.float64(st["t_coarse_us"] / 1e3)}
for i, r in enumerate(res):
    for k in ("semantic", "coarse", "fine", "pcm"):
        d["%%s%%d" %% (k, i)] = np.asarray(r[k])
np.savez(out, **d)
ctx.free()
''' % ROOT

FAST_CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
import bench
pkg = load_package()
preset, out = sys.argv[1], sys.argv[2]
ctx = pkg.BarkContext.load_model(ensure_model(preset, 0), pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=256), 0)
rng = np.random.default_rng(0)
hp1 = ctx.hparams(1)
d = {}
sem_prompt = ctx.tokenize(bench.synth_prompts(1)[0])
logits, n_past = ctx.gpt_eval(0, sem_prompt, 0, True)
d["sem_prefill"] = logits
coarse_prompt = rng.integers(0, min(hp1["n_in"], 12000), 300).astype(np.int32)
logits, n_past = ctx.gpt_eval(1, coarse_prompt, 0, False)
d["coarse_prefill"] = logits
fine_tokens = rng.integers(0, 1024, (8, 1024)).astype(np.int32)
d["fine_nn2"] = ctx.fine_eval(fine_tokens, 2)
d["fine_nn7"] = ctx.fine_eval(fine_tokens, 7)
ok = ctx.generate_audio(bench.synth_prompts(1)[0])
d["gen_semantic"] = ctx.semantic_tokens(); d["gen_coarse"] = ctx.coarse_tokens(); d["gen_fine"] = ctx.fine_tokens(); d["gen_pcm"] = ctx.audio_data()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); ctx.generate_audio(bench.synth_prompts(1)[0]); ts.append(time.perf_counter() - t0)
d["gen_s"] = np.float64(min(ts))
st = ctx.stats()
d["fine_ms"] = np.float64(st["t_fine_us"] / 1e3); d["semantic_ms"] = np.float64(st["t_semantic_us"] / 1e3); d["coarse_ms"] = np.float64(st["t_coarse_us"] / 1e3)
try:
    us, flops = ctx.time_fine_pass(5)
    d["fine_pass_us"] = np.float64(us)
except Exception as e:
    d["fine_pass_us"] = np.float64(-1)
np.savez(out, **d)
ctx.free()
''' % ROOT


def run_child(code, args, env):
    with tempfile.NamedTemporaryFile(suffix=".npz", delete=False) as f:
        path = f.name
    e = dict(os.environ); e.update(env)
    p = subprocess.run([sys.executable, "-c", code] + args + [path], env=e, capture_output=True, text=True)
    if p.returncode != 0:
        return None, p.stderr[-1500:]
    import numpy as np
    return np.load(path), ""


def parse_arm(a):
    name, _, kv = a.partition(":")
    return name, dict(x.split("=") for x in kv.split(",") if x)


def main():
    import numpy as np
    mode = sys.argv[1]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    report, bad = {"mode": mode}, False
    if mode == "batch":
        preset, B, n_sem = sys.argv[2], sys.argv[3], sys.argv[4]
        arms = [parse_arm(a) for a in sys.argv[5:]]
        ref = None
        for name, env in arms:
            got, err = run_child(BATCH_CHILD, [preset, B, n_sem], env)
            if got is None:
                report[name] = {"error": err}; bad = True; continue
            r = {"dt_s": round(float(got["dt"]), 3), "semantic_ms": round(float(got["semantic_ms"]), 1), "coarse_ms": round(float(got["coarse_ms"]), 1)}
            if ref is None:
                ref = got
            else:
                diff = [k for k in ref.files if k not in ("dt", "semantic_ms", "coarse_ms") and not (ref[k].shape == got[k].shape and np.array_equal(ref[k], got[k]))]
                r["equal_to_first_arm"] = not diff
                if diff:
                    r["first_differences"] = diff[:6]; bad = True
            report[name] = r
        report.update(preset=preset, B=int(B), n_semantic=int(n_sem))
    elif mode == "fast":
        preset = sys.argv[2]
        exact, err0 = run_child(FAST_CHILD, [preset], {"BARK_HIP_FAST_GEMM": "0"})
        fast, err1 = run_child(FAST_CHILD, [preset], {"BARK_HIP_FAST_GEMM": "1"})
        if exact is None or fast is None:
            report["error"] = err0 or err1; bad = True
        else:
            tol = 5e-3       # f32 re-association noise amplified by the f16 rounding of the activations between products (R1); logits are O(1)
            for k in ("sem_prefill", "coarse_prefill", "fine_nn2", "fine_nn7"):
                e = float(np.max(np.abs(exact[k].astype(np.float64) - fast[k].astype(np.float64))))
                report[k] = {"max_abs_diff": e, "max_abs_logit": float(np.max(np.abs(exact[k]))), "tolerance": tol,
                             "argmax_agreement": float(np.mean(np.argmax(exact[k], axis=-1) == np.argmax(fast[k], axis=-1))) if exact[k].ndim > 1 else bool(np.argmax(exact[k]) == np.argmax(fast[k]))}
                if not (0 < e <= tol):
                    bad = True
            for k in ("gen_semantic", "gen_coarse", "gen_fine"):
                a, b = exact[k].ravel(), fast[k].ravel()
                n = min(len(a), len(b))
                d = np.flatnonzero(a[:n] != b[:n])
                report[k] = {"n_exact": int(len(a)), "n_fast": int(len(b)), "equal_ids": int(n - len(d)), "first_difference": int(d[0]) if len(d) else None}
            for k in ("gen_s", "fine_ms", "semantic_ms", "coarse_ms", "fine_pass_us"):
                report[k] = {"exact": round(float(exact[k]), 3), "fast": round(float(fast[k]), 3)}
        report["preset"] = preset
    else:
        raise SystemExit("mode: batch | fast")
    report["ok"] = not bad
    with open(os.path.join(ROOT, "gpurun_out", "check_routes_%s.json" % mode), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
