#!/usr/bin/env python3
"""Clones of one context serving lock-step jobs from concurrent host threads, in a loop, with every difference reported in full.

This is tests/test_gpu_batch_ragged.py::test_cloned_contexts_serve_jobs_from_concurrent_host_threads turned into an instrument (round 4's
GPU suite went red on it on the driver's box: 7 of 72 coarse ids of ONE utterance, from index 54 of the first 60-step window).  Per
iteration: G clones, one host thread each, J jobs per thread back to back; every utterance is compared with the same job run alone on the
base context AND with its own oracle run (so a difference says which side is wrong), and every differing index is printed with both
values and with what the slot held at those positions in the clone's previous job (a stale read-back shows up as exactly those).

    python tools/clone_stress.py [iterations=10] [preset=mini] [G=4] [J=3] [seconds=inf]

Arms are environment variables of the engine (set by the caller): BARK_HIP_TAIL_STREAM=0, BARK_HIP_GRAPH=0, BARK_HIP_FEW_SLOTS=0, BARK_HIP_POISON=1,
BARK_HIP_GUARD=1 (the read-back / no-tail / tail-priority arms of calls 1 - 3 of round 5 were removed with their switches once the root cause was found).  Exit status 1 if anything differed."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    preset = sys.argv[2] if len(sys.argv) > 2 else "mini"
    G = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    J = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    budget_s = float(sys.argv[5]) if len(sys.argv) > 5 else float("inf")      # stop after this many seconds of iterations
    KEYS = ("semantic", "coarse", "fine", "pcm")
    import bench
    from bark_amd_loader import load_package
    from oracle import pyoracle
    from oracle.pyoracle import Oracle
    from tools.make_synth_model import ensure_model
    pyoracle.JOB_ORDER = True          # everything here is a lock-step job: the oracle computes the fine products in the jobs' order (C1m)
    pkg = load_package()
    path = ensure_model(preset, 0)
    cap = 24
    prompts = bench.synth_prompts(24)
    arms = {k: os.environ[k] for k in sorted(os.environ) if k.startswith("BARK_HIP_")}
    print(f"clone_stress: {iters} iterations, {preset}, {G} threads x {J} jobs, arms {arms}", flush=True)
    base = pkg.BarkContext.load_model(path, pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=cap), 0)
    jobs = [[[prompts[(5 * g + 7 * j + i) % 24] for i in range(3 + g + 2 * j)] for j in range(J)] for g in range(G)]
    ref = [[base.generate_batch(job) for job in jobs[g]] for g in range(G)]
    # the oracle's word on every distinct prompt (greedy, the same cap)
    orc = Oracle(path, n_threads=4)
    want = {}
    for g in range(G):
        for j in range(J):
            for i, text in enumerate(jobs[g][j]):
                if text not in want:
                    orc.seed(0)
                    want[text] = orc.generate(text, orc.params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=cap))
                for k in KEYS:
                    if not np.array_equal(np.asarray(ref[g][j][i][k]), np.asarray(want[text][k])):
                        print(f"BASE differs from the oracle: thread {g} job {j} utterance {i} {k}", flush=True)
    orc.close()
    n_bad = 0
    t0 = time.time()
    done = 0
    for it in range(iters):
        if time.time() - t0 > budget_s:
            break
        done += 1
        clones = [base.clone(g + 1) for g in range(G)]
        got = [[None] * J for _ in range(G)]
        errors = []

        def run(g):
            try:
                for j in range(J):
                    got[g][j] = clones[g].generate_batch(jobs[g][j])
            except Exception as e:                                  # noqa: BLE001
                errors.append((g, repr(e)))
        th = [threading.Thread(target=run, args=(g,)) for g in range(G)]
        for t in th: t.start()
        for t in th: t.join()
        for c in clones: c.free()
        if errors:
            print(f"iteration {it}: errors {errors}", flush=True)
            n_bad += 1
            continue
        for g in range(G):
            for j in range(J):
                for i, (a, b) in enumerate(zip(got[g][j], ref[g][j])):
                    for k in KEYS:
                        x, y = np.asarray(a[k]).ravel(), np.asarray(b[k]).ravel()
                        if x.shape == y.shape and np.array_equal(x, y):
                            continue
                        n_bad += 1
                        w = np.asarray(want[jobs[g][j][i]][k]).ravel()
                        if x.shape != y.shape:
                            print(f"iteration {it} thread {g} job {j} ({len(jobs[g][j])} utterances) utterance {i} {k}: shape {x.shape} vs {y.shape}", flush=True)
                            continue
                        bad = np.flatnonzero(x != y)
                        side = "clone wrong (base == oracle)" if np.array_equal(y, w) else "clone == oracle, base wrong" if np.array_equal(x, w) else "both differ from the oracle"
                        line = f"iteration {it} thread {g} job {j} ({len(jobs[g][j])} utterances) utterance {i} {k}: {bad.size}/{x.size} differ at {bad[:12].tolist()}: got {x[bad[:12]].tolist()} want {y[bad[:12]].tolist()} [{side}]"
                        if k in ("semantic", "coarse") and j > 0 and i < len(got[g][j - 1]):
                            prev = np.asarray(got[g][j - 1][i][k]).ravel()
                            if prev.size > bad.max():
                                line += f"; same slot, previous job, same positions: {prev[bad[:12]].tolist()}"
                        print(line, flush=True)
        print(f"iteration {it} done, {n_bad} differing arrays so far, {time.time() - t0:.0f} s", flush=True)
    base.free()
    print(f"clone_stress: {n_bad} differing arrays in {done} iterations ({G} threads x {J} jobs), {time.time() - t0:.0f} s")
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
