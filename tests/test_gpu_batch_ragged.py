"""Lock-step jobs (bark_hip_generate_batch_ex) under raggedness: every utterance of a job carries its own step cap, stop threshold,
temperatures and seed, jobs are larger than the slot count (slots are refilled from the queue, the batch is compacted when nobody waits),
and each utterance must equal ITS OWN oracle run - ids of all three stages and the PCM, bit for bit (reference loops:
/root/reference/bark.cpp:1669-1695 step cap / stop rule, :1787-1845 coarse windows, :1998-2038 fine windows)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pkg():
    from bark_amd_loader import load_package
    return load_package()


def _exact(name, got, ref):
    got = np.asarray(got); ref = np.asarray(ref)
    assert got.shape == ref.shape, f"{name}: shape {got.shape} vs {ref.shape}"
    if not np.array_equal(got, ref):
        bad = np.flatnonzero(got.ravel() != ref.ravel())
        raise AssertionError(f"{name}: {bad.size}/{got.size} elements differ, first at {bad[0]}")


def _check_job(tag, res, orc, texts, reqs, **ctx_params):
    for i, (text, rq, r) in enumerate(zip(texts, reqs, res)):
        orc.seed(int(rq.seed))
        ref = orc.generate(text, orc.params(temp=rq.temp, fine_temp=rq.fine_temp, min_eos_p=rq.min_eos_p, n_steps_text_encoder=rq.n_steps_text_encoder, **ctx_params))
        t = f"{tag} utterance {i} (cap {rq.n_steps_text_encoder}, temp {rq.temp:.2f}/{rq.fine_temp:.2f}, min_eos_p {rq.min_eos_p:.2g}, seed {rq.seed})"
        if len(ref["semantic"]) == 0 or ref["n_frames"] == 0:
            assert r is None or len(r["pcm"]) == 0, t + ": the oracle produced no audio"
            continue
        assert r is not None, t + ": no audio"
        _exact(t + " semantic", r["semantic"], ref["semantic"])
        _exact(t + " coarse", r["coarse"], ref["coarse"])
        _exact(t + " fine", r["fine"], ref["fine"])
        _exact(t + " pcm", r["pcm"], ref["pcm"])


@pytest.mark.lock_step_job
@pytest.mark.parametrize("preset", ["toy", "mini"])
def test_randomised_lock_step_jobs_against_the_oracle(preset):
    """A seeded sweep: job sizes 2..40 on 8 / 16 / 64 slots, per-utterance caps 1..120 (one utterance per preset runs past 1024 frames:
    several fine windows), stop thresholds that fire at different steps (mini), greedy and sampled utterances mixed in one job with their
    own seeds."""
    import bench
    from oracle.pyoracle import Oracle
    from tools.make_synth_model import ensure_model
    pkg = _pkg()
    path = ensure_model(preset, 0)
    rng = np.random.default_rng({"toy": 4242, "mini": 2424}[preset])
    words = " ".join(bench.synth_prompts(16)).split()
    orc = Oracle(path, n_threads=4)
    eos_choices = [0.2, 0.05] if preset == "toy" else [1.4e-4, 2.5e-4, 0.2]
    # (slots, utterances, context parameters): the last toy trial has an ODD sliding window - the slots of a lock step share the codebook parity of
    # their step, so newcomers may join only in every second window
    trials = ([(8, 21, {}), (16, 5, {}), (64, 40, {}), (8, 2, {}), (16, 33, {}), (8, 19, dict(sliding_window_size=17, max_coarse_history=64))] if preset == "toy"
              else [(8, 13, {}), (16, 24, {}), (64, 9, {})])
    try:
        for it, (slots, n, cparams) in enumerate(trials):
            ctx = pkg.BarkContext.load_model(path, pkg.default_params(**cparams), seed=int(rng.integers(0, 2**31)))
            ctx.reserve_batch(slots)
            texts, reqs = [], []
            for i in range(n):
                texts.append(" ".join(rng.choice(words, size=int(rng.integers(1, 40)))))
                # temperatures drawn independently: greedy / sampled semantic + coarse stages with a greedy / sampled fine stage in any combination
                reqs.append(ctx.request_params(temp=float(rng.choice([0.7, 1.0])) if rng.random() < 0.4 else 0.0,
                                               fine_temp=0.5 if rng.random() < 0.3 else 0.0,
                                               min_eos_p=float(rng.choice(eos_choices)), n_steps_text_encoder=int(rng.integers(1, 121)),
                                               seed=int(rng.integers(0, 2**31))))
            if it == 0:
                reqs[n // 2].n_steps_text_encoder = 700          # > 1024 frames unless the stop rule fires: windows of 1024 hopping by 512
                reqs[n // 2].min_eos_p = 0.9
            res = ctx.generate_batch(texts, params=reqs)
            _check_job(f"{preset} job {it} ({n} utterances on {slots} slots)", res, orc, texts, reqs, **cparams)
            if it == 0:
                assert len(res[n // 2]["fine"]) > 1024 or preset == "mini"
            # the same context again with another job: slots, caches and graphs are reused
            res2 = ctx.generate_batch(texts[:3][::-1], params=reqs[:3][::-1])
            _check_job(f"{preset} job {it}, second job on the context", res2, orc, texts[:3][::-1], reqs[:3][::-1], **cparams)
            ctx.free()
    finally:
        orc.close()


@pytest.mark.lock_step_job
def test_job_larger_than_the_slots_equals_the_job_on_enough_slots(toy_model):
    """23 utterances through 8 slots (queue, refills, compaction) against the same 23 on 32 slots: the company an utterance travels in
    changes nothing."""
    import bench
    pkg = _pkg()
    texts = bench.synth_prompts(23)
    outs = []
    for slots in (8, 32):
        ctx = pkg.BarkContext.load_model(toy_model, pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=40), 0)
        ctx.reserve_batch(slots)
        reqs = [ctx.request_params(n_steps_text_encoder=10 + 7 * (i % 9)) for i in range(len(texts))]
        outs.append(ctx.generate_batch(texts, params=reqs))
        ctx.free()
    for i, (a, b) in enumerate(zip(*outs)):
        for k in ("semantic", "coarse", "fine", "pcm"):
            _exact(f"utterance {i} {k}", a[k], b[k])


@pytest.mark.lock_step_job
def test_job_tail_on_a_second_stream_equals_the_one_stream_job(toy_model, monkeypatch):
    """The fine passes and the codec of utterances that have left the coarse stage run on a clone of the context beside the lock steps of the
    others (engine_batch.hip: JobTail); BARK_HIP_TAIL_STREAM=0 keeps them behind the coarse stage on the job's own stream.  Same job (31 utterances
    with their own caps and temperatures on 8 slots: utterances leave at every window boundary), both forms, equal bit for bit - and the stage
    counters of the two-stream job still account for every utterance."""
    import bench
    pkg = _pkg()
    texts = bench.synth_prompts(31)
    outs, stats = [], []
    for arm in ("0", "1"):
        monkeypatch.setenv("BARK_HIP_TAIL_STREAM", arm)
        ctx = pkg.BarkContext.load_model(toy_model, pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=40), 0)
        ctx.reserve_batch(8)
        reqs = [ctx.request_params(n_steps_text_encoder=5 + 11 * (i % 10), temp=0.7 if i % 3 == 0 else 0.0, fine_temp=0.5 if i % 4 == 1 else 0.0, seed=100 + i)
                for i in range(len(texts))]
        outs.append(ctx.generate_batch(texts, params=reqs))
        stats.append(ctx.stats())
        ctx.free()
    for i, (a, b) in enumerate(zip(*outs)):
        for k in ("semantic", "coarse", "fine", "pcm"):
            _exact(f"utterance {i} {k}", a[k], b[k])
    for k in ("n_sample_semantic", "n_sample_coarse", "n_sample_fine", "n_frames", "n_samples"):
        assert stats[0][k] == stats[1][k] and stats[0][k] > 0, (k, stats[0][k], stats[1][k])
    # the helper's clone replays captured fine passes that bake the pick and its temperature: after bark_hip_set_params on the job's context a job
    # without per-utterance parameters must see the new ones there too (same seeds, fresh contexts as the reference)
    monkeypatch.setenv("BARK_HIP_TAIL_STREAM", "1")
    seeds = list(range(40, 46))
    ctx = pkg.BarkContext.load_model(toy_model, pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=20), 0)
    ctx.reserve_batch(8)
    got = [ctx.generate_batch(texts[:6], seeds=seeds)]
    ctx.set_params(pkg.default_params(temp=0.0, fine_temp=0.5, n_steps_text_encoder=20))
    got.append(ctx.generate_batch(texts[:6], seeds=seeds))
    ctx.set_params(pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=20))
    got.append(ctx.generate_batch(texts[:6], seeds=seeds))
    ctx.free()
    for j, ft in enumerate((0.0, 0.5, 0.0)):
        fresh = pkg.BarkContext.load_model(toy_model, pkg.default_params(temp=0.0, fine_temp=ft, n_steps_text_encoder=20), 0)
        ref = fresh.generate_batch(texts[:6], seeds=seeds)
        fresh.free()
        for i, (a, b) in enumerate(zip(got[j], ref)):
            for k in ("fine", "pcm"):
                _exact(f"job {j} (fine_temp {ft}) utterance {i} {k}", a[k], b[k])
    assert any(not np.array_equal(a["fine"], b["fine"]) for a, b in zip(got[0], got[1])), "the sampled fine stage should differ from the greedy one"


@pytest.mark.lock_step_job
@pytest.mark.slow
def test_small_ragged_job_matches_committed_oracle_outputs(small_model):
    """bark-small shapes, 16 slots, the ragged form of BASELINE config 5 (bench.ragged_caps: step caps 64..256 by prompt length): every
    16th... every 4th of the 64 bench prompts against oracle outputs committed as tests/golden/oracle_small_ragged16.npz
    (tools/make_oracle_golden.py ragged; the CPU suite re-derives a sample of them)."""
    import bench
    pkg = _pkg()
    g = np.load(os.path.join(ROOT, "tests", "golden", "oracle_small_ragged16.npz"))
    prompts = bench.synth_prompts(64)
    caps = bench.ragged_caps(prompts)
    idx = [int(i) for i in g["prompt_index"]]
    ctx = pkg.BarkContext.load_model(small_model, pkg.default_params(temp=0.0, fine_temp=0.0), 0)
    ctx.reserve_batch(16)
    reqs = [ctx.request_params(n_steps_text_encoder=caps[i]) for i in idx]
    res = ctx.generate_batch([prompts[i] for i in idx], params=reqs)
    import hashlib
    for k, i in enumerate(idx):
        assert res[k] is not None
        assert len(res[k]["semantic"]) == caps[i]
        _exact(f"prompt {i} semantic", res[k]["semantic"], g[f"semantic{k}"].astype(np.int32))
        _exact(f"prompt {i} coarse", res[k]["coarse"], g[f"coarse{k}"].astype(np.int32))
        _exact(f"prompt {i} fine", res[k]["fine"], g[f"fine{k}"].astype(np.int32))
        pcm = np.ascontiguousarray(res[k]["pcm"], np.float32)
        assert pcm.size == int(g[f"pcm_len{k}"])
        assert np.array_equal(np.frombuffer(hashlib.sha256(pcm.tobytes()).digest(), np.uint8), g[f"pcm_sha256_{k}"]), f"prompt {i}: PCM differs from the oracle's"
    ctx.free()


@pytest.mark.concurrency
@pytest.mark.job_order
def test_request_collector_admits_late_requests_with_their_own_parameters(toy_model, toy_oracle):
    """bark_hip_batcher with continuous admission: a long request opens a job; requests submitted while it is in its semantic stage join the
    running job (free slots, nobody of the job waiting) instead of waiting for the next one.  Every request carries its own parameters and
    gets the PCM of its own oracle run."""
    import threading
    import time
    import bench
    pkg = _pkg()
    texts = bench.synth_prompts(9)
    c = pkg.BarkContext.load_model(toy_model, pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=24), 0)
    b = pkg.Batcher(c, max_batch=16, max_wait_ms=1)
    reqs = [c.request_params(n_steps_text_encoder=700, min_eos_p=0.9)]                       # the opener: hundreds of lock steps
    reqs += [c.request_params(n_steps_text_encoder=8 + 5 * i, temp=0.7 if i % 3 == 0 else 0.0, fine_temp=0.5 if i % 3 == 0 else 0.0, seed=100 + i) for i in range(1, len(texts))]
    out = [None] * len(texts)

    def client(i, delay):
        time.sleep(delay)
        out[i] = b.wait(b.submit(texts[i], params=reqs[i]))

    try:
        th = [threading.Thread(target=client, args=(i, 0.0 if i == 0 else 0.02 + 0.004 * i)) for i in range(len(texts))]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        st = b.stats()
        assert st["n_requests"] == len(texts), st
        assert st["n_admitted"] >= 1, f"no request joined the running job: {st}"
        for i, (text, rq) in enumerate(zip(texts, reqs)):
            toy_oracle.seed(int(rq.seed))
            ref = toy_oracle.generate(text, toy_oracle.params(temp=rq.temp, fine_temp=rq.fine_temp, min_eos_p=rq.min_eos_p, n_steps_text_encoder=rq.n_steps_text_encoder))
            _exact(f"request {i}", out[i], ref["pcm"])
    finally:
        b.free()
        c.free()


@pytest.mark.lock_step_job
def test_lock_step_time_line_hook(toy_model):
    """bark_hip_profile_lock_step: one entry per launch site of a lock step in launch order (5 per layer + LM head + sampler), closed by the
    graph-replayed step; the eager sites add up to more than the replayed step (event records sit between the launches)."""
    pkg = _pkg()
    ctx = pkg.BarkContext.load_model(toy_model, pkg.default_params(temp=0.0, fine_temp=0.0), 0)
    try:
        ctx.reserve_batch(16)
        n_layer = ctx.hparams(1)["n_layer"]
        for which in (0, 1):
            tl = ctx.profile_lock_step(which, 9, 300, 5)
            sites = [e["site"] for e in tl]
            assert sites[:5] == ["ln1+qkv", "attention", "proj", "ln2+fc+gelu", "mlp_proj"] and len(tl) == 5 * n_layer + 3
            assert sites[-3:] == ["lnf+lm_head", "sample+embed", "step (graph replay)"]
            assert all(e["us"] > 0 for e in tl) and tl[-1]["us"] < sum(e["us"] for e in tl[:-1])
        # the hook leaves the context usable
        res = ctx.generate_batch(["hello world", "the water is cold"], params=[ctx.request_params(n_steps_text_encoder=12)] * 2)
        assert all(r is not None and len(r["pcm"]) > 0 for r in res)
    finally:
        ctx.free()


@pytest.mark.lock_step_job
@pytest.mark.parametrize("kind", ["q4_0", "f32"])
def test_ragged_job_on_quantised_and_f32_model_files(kind, toy_q4_model, toy_f32_model):
    """Per-utterance parameters on the other weight formats: a q4_0 file runs the lock-step path on the per-pair VALU products, an f32 file the
    sequential fallback of bark_hip_generate_batch_ex (one utterance in flight, the context's parameters switched per utterance) - each
    utterance equals its own oracle run on the same file."""
    from oracle.pyoracle import Oracle
    import bench
    pkg = _pkg()
    path = toy_q4_model if kind == "q4_0" else toy_f32_model
    orc = Oracle(path, n_threads=4)
    ctx = pkg.BarkContext.load_model(path, pkg.default_params(), seed=3)
    try:
        ctx.reserve_batch(8)
        texts = bench.synth_prompts(11)
        reqs = [ctx.request_params(temp=0.7 if i % 4 in (1, 3) else 0.0, fine_temp=0.5 if i % 4 in (1, 2) else 0.0, min_eos_p=0.2,
                                   n_steps_text_encoder=5 + 9 * (i % 7), seed=50 + i) for i in range(len(texts))]
        res = ctx.generate_batch(texts, params=reqs)
        _check_job(f"{kind} toy job", res, orc, texts, reqs)
    finally:
        ctx.free(); orc.close()


@pytest.mark.concurrency
def test_device_and_host_sampling_agree_on_many_sampled_utterances():
    """tools/sampling_soak.py: 64 sampled utterances (temp 0.7 / 1.0, fine_temp 0.5, own seeds) as one lock-step job with the device multinomial
    kernels against the same utterances with BARK_HIP_HOST_SAMPLING=1 (std::discrete_distribution on fetched logits): ~45 000 ids and
    ~400 000 picks (padding rows of the fine windows included), all equal - the picks next to a bin boundary go through the exact path."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sampling_soak.py"), "toy", "64", "60"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert " 0 arrays differ" in r.stdout


@pytest.mark.concurrency
def test_cloned_contexts_serve_jobs_from_concurrent_host_threads():
    """Four clones of one context (shared weights, own stream / caches / graphs), one host thread each, three different jobs per thread back to
    back: every thread captures its lock-step graphs while the others copy results to the host.  The context streams are non-blocking and no call
    goes through the legacy stream, which refuses work while any blocking stream of the process is capturing (found by tools/staggered_jobs.py:
    'operation would make the legacy stream depend on a capturing blocking stream').  Results must equal the same jobs run one after the other."""
    import threading
    import bench
    from tools.make_synth_model import ensure_model
    pkg = _pkg()
    path = ensure_model("mini", 0)
    prompts = bench.synth_prompts(24)
    base = pkg.BarkContext.load_model(path, pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=24), 0)
    G, J = 4, 3
    jobs = [[[prompts[(5 * g + 7 * j + i) % 24] for i in range(3 + g + 2 * j)] for j in range(J)] for g in range(G)]       # sizes 3 .. 10: different graphs
    ref = [[base.generate_batch(job) for job in jobs[g]] for g in range(G)]
    clones = [base.clone(g + 1) for g in range(G)]
    got = [[None] * J for _ in range(G)]
    errors = []
    def run(g):
        try:
            for j in range(J):
                got[g][j] = clones[g].generate_batch(jobs[g][j])
        except Exception as e:                                  # noqa: BLE001 - reported below with the thread's index
            errors.append((g, repr(e)))
    try:
        th = [threading.Thread(target=run, args=(g,)) for g in range(G)]
        for t in th: t.start()
        for t in th: t.join()
        assert not errors, errors
        for g in range(G):
            for j in range(J):
                for i, (a, b) in enumerate(zip(got[g][j], ref[g][j])):
                    for k in ("semantic", "coarse", "fine", "pcm"):
                        _exact(f"thread {g} job {j} utterance {i} {k}", a[k], b[k])
    finally:
        for c in clones: c.free()
        base.free()
