#!/usr/bin/env python3
"""How much do Bark's greedy token streams depend on the summation order of the dot products?

The engine and the oracle share ONE canonical order; the reference's ggml CPU backend sums in the SIMD order of
the host it was built for, so "the reference's bits" are not a single target.  This tool runs the CPU oracle's stage loops under
three orders - 0 canonical, 1 ggml's AVX2 order (restated from upstream ggml), 2 one sequential chain (Numerics::dot_order) - on the
bench prompt and reports, per stage, where the token streams first part and how close the two leading logits were there.

  python tools/order_sensitivity.py [preset=small] [n_semantic=256] [out.json]

CPU only (test infrastructure: it drives oracle/, never the product).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                   # noqa: E402
from oracle.pyoracle import Oracle                   # noqa: E402
from tools.make_synth_model import ensure_model      # noqa: E402
import bench                                         # noqa: E402

ORDERS = {0: "canonical (16 chains x 8-element chunks; C2 / C5)", 1: "ggml AVX2 (32 chains, step 32)", 2: "sequential (one chain)"}


def first_diff(a, b):
    a = np.asarray(a).ravel(); b = np.asarray(b).ravel()
    n = min(len(a), len(b))
    d = np.flatnonzero(a[:n] != b[:n])
    if len(d):
        return int(d[0])
    return None if len(a) == len(b) else n


def semantic_margin(orc, prompt, tokens, step):
    """top-2 margin (in the sampler's units: logits / 0.7) of the canonical logits at semantic step `step`"""
    orc.set_dot_order(0)
    logits, n_past = orc.gpt_eval(0, prompt, 0, True)
    for t in tokens[:step]:
        logits, n_past = orc.gpt_eval(0, [int(t)], n_past, True)
    l = np.sort(logits / np.float32(0.7))[::-1]
    return float(l[0] - l[1]), int(np.argmax(logits))


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "small"
    n_sem = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    threads = min(os.cpu_count() or 4, 8)
    orc = Oracle(ensure_model(preset, 0), n_threads=threads)
    text = bench.synth_prompts(64)[0]
    prompt = orc.tokenize(text)
    p = orc.params(n_steps_text_encoder=n_sem)
    runs = {}
    for order in ORDERS:
        orc.set_dot_order(order)
        t0 = time.time()
        sem = orc.semantic(prompt, p)
        co = orc.coarse(sem, p)
        fi = orc.fine(co, p)
        runs[order] = {"semantic": sem, "coarse": co, "fine": fi, "s": time.time() - t0}
        print(f"order {order}: {len(sem)} semantic, {co.shape} coarse, {fi.shape} fine in {runs[order]['s']:.1f} s", flush=True)
    # teacher-forced: later stages fed with the CANONICAL tokens of the stage before, so that each stage's own sensitivity shows
    forced = {}
    for order in (1, 2):
        orc.set_dot_order(order)
        forced[order] = {"coarse": orc.coarse(runs[0]["semantic"], p)}
        forced[order]["fine"] = orc.fine(runs[0]["coarse"], p)
    report = {"preset": preset, "prompt": text, "n_steps_text_encoder": n_sem, "orders": ORDERS, "free_running": {}, "teacher_forced": {}}
    for order in (1, 2):
        fr = {}
        for stage in ("semantic", "coarse", "fine"):
            a, b = runs[0][stage], runs[order][stage]
            d = first_diff(a, b)
            n = min(np.asarray(a).size, np.asarray(b).size)
            fr[stage] = {"tokens": int(np.asarray(a).size), "first_difference_at": d,
                         "agreeing_tokens": int((np.asarray(a).ravel()[:n] == np.asarray(b).ravel()[:n]).sum())}
        d = fr["semantic"]["first_difference_at"]
        if d is not None and d < len(runs[0]["semantic"]):
            m, arg = semantic_margin(orc, prompt, runs[0]["semantic"], d)
            fr["semantic"]["top2_margin_at_first_difference"] = m
        report["free_running"][str(order)] = fr
        tf = {}
        for stage in ("coarse", "fine"):
            a, b = runs[0][stage], forced[order][stage]
            n = min(np.asarray(a).size, np.asarray(b).size)
            tf[stage] = {"tokens": int(np.asarray(a).size), "first_difference_at": first_diff(a, b),
                         "agreeing_tokens": int((np.asarray(a).ravel()[:n] == np.asarray(b).ravel()[:n]).sum())}
        report["teacher_forced"][str(order)] = tf
    # how far apart are the logits themselves (first decode step, canonical prefix)?
    orc.set_dot_order(0); l0, npast = orc.gpt_eval(0, prompt, 0, True)
    dl = {}
    for order in (1, 2):
        orc.set_dot_order(order); l, _ = orc.gpt_eval(0, prompt, 0, True)
        dl[str(order)] = {"max_abs_logit_difference_prefill": float(np.abs(l - l0).max()), "logit_scale": float(np.abs(l0).max())}
    report["logits"] = dl
    orc.close()
    print(json.dumps(report, indent=1))
    if out_path:
        json.dump(report, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
