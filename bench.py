#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X Bark engine.

Metric (BASELINE.json): audio-sec/sec (real-time factor) of bark_generate_audio on bark-small f16, greedy (temp = fine_temp = 0),
synthetic weights / synthetic prompts (no checkpoints offline), n_steps_text_encoder = 256 -> 256 semantic + 768 coarse tokens,
6 fine passes, 384 frames = 5.12 s of audio per prompt.

  N = 1 (BASELINE config 2): a "step" = one bark_generate_audio call (single prompt, hipGraph decode).  The line also carries
        `config5_64_prompts`: the 64-prompt job of config 5 on this one GPU (ONE lock-step job on 64 slots), i.e. the N = 1 point of the
        multi-GPU curve, and `config5_ragged`, the same prompts with step caps 64 .. 256 by prompt length.
  N > 1 (BASELINE config 5): a "step" = the 64-prompt synthetic batch, sorted by length and split statically 64 / N per rank; every
        rank runs bark_hip_generate_batch (lock-step decode) on its shard - no collective inside an utterance - then the sample
        counts are all-gathered and the PCM is gathered on rank 0 (RCCL over xGMI; edge collectives only).  Total work is fixed, so
        `scaling` is "strong"; `value` = audio seconds of all 64 prompts / MAX-over-ranks time.

  python bench.py [--gpus N --steps K --warmup W]     (N > 1: one rank per GPU; under torch.distributed.run as the driver launches it, or
                                                       plain - the process then re-executes itself under that launcher, launch_ranks();
                                                       --gpus that disagrees with WORLD_SIZE or with the node's GPU count exits 2)

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel against HBM) and, at N = 1, `cpu_baseline` (the CPU oracle timed on
this host, 4 pinned threads, on the headline workload itself - about 20 s; test infrastructure used as the measured-beside baseline
only, never on the product path).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PROMPT_WORDS = ("the of and to in is that it was for on are as with his they be at one have this from or had by "
                "hot but some what there we can out other were all your when up use word how said an each she").split()


def synth_prompts(n=64, seed=0):
    import numpy as np
    rng = np.random.default_rng(seed)
    return [" ".join(rng.choice(PROMPT_WORDS, size=int(rng.integers(5, 40)))) for _ in range(n)]


def prompt_for(step: int, rank: int, world: int, prompts):
    """N = 1 headline: at step s the single rank takes prompt s."""
    return prompts[(step * world + rank) % len(prompts)]


def shard_prompts(prompts, rank: int, world: int):
    """Config 5: the prompts sorted by length (ties by index) are dealt to the ranks snake-wise - rank r takes the sorted positions
    r, 2 world - 1 - r, 2 world + r, 4 world - 1 - r, ... - so that no rank holds all the longest utterances (the job's time is the
    MAX over ranks), and every shard stays sorted by length (neighbours share a lock-step batch).  Returns the indices (into
    `prompts`) of this rank's shard."""
    order = sorted(range(len(prompts)), key=lambda i: (len(prompts[i]), i))
    return [order[p] for p in range(len(order)) if p % (2 * world) in (rank, 2 * world - 1 - rank)]


def reduce_timing(dt: float, audio_s: float, world: int, device=None):
    """Bench contract: time = MAX over ranks, work = SUM over ranks (edge collectives only)."""
    if world == 1:
        return dt, audio_s
    import torch
    import torch.distributed as dist
    t = torch.tensor([dt, audio_s], dtype=torch.float64, device=device)
    tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    return float(tmax[0]), float(tsum[1])


def gather_batch_results(pcms, idx, n_total: int, rank: int, world: int, device=None):
    """Edge collectives of config 5: all_gather of the per-prompt sample counts, gather of the PCM on rank 0.
    pcms: this rank's float32 arrays in shard order; idx: their prompt indices.  Returns (counts[n_total], {index: pcm} on rank 0)."""
    import numpy as np
    if world == 1:
        counts = np.zeros(n_total, np.int64)
        for i, p in zip(idx, pcms):
            counts[i] = len(p)
        return counts, dict(zip(idx, pcms))
    import torch
    import torch.distributed as dist
    per = (n_total + world - 1) // world
    mine = torch.full((per, 2), -1, dtype=torch.int64, device=device)
    for k, (i, p) in enumerate(zip(idx, pcms)):
        mine[k, 0] = i; mine[k, 1] = len(p)
    allc = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allc, mine)                                   # sample counts (and which prompt they belong to)
    table = torch.stack(allc).cpu().numpy()
    counts = np.zeros(n_total, np.int64)
    for r in range(world):
        for i, n in table[r]:
            if i >= 0:
                counts[i] = n
    longest = int(table[:, :, 1].max())
    buf = torch.zeros((per, max(longest, 1)), dtype=torch.float32, device=device)
    for k, p in enumerate(pcms):
        buf[k, :len(p)] = torch.from_numpy(p).to(buf.device)
    parts = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, parts, dst=0)                                # PCM to rank 0
    out = {}
    if rank == 0:
        for r in range(world):
            host = parts[r].cpu().numpy()
            for k, (i, n) in enumerate(table[r]):
                if i >= 0:
                    out[int(i)] = host[k, :n].copy()
    return counts, out


def n1_reference(a, n_prompts: int):
    """The N = 1 point of the N > 1 workload, from the newest committed single-GPU bench line (profiles/rNN_bench_small_n1.json; labelled as a committed reference - nothing ties it to the build being run): the
    64-prompt job on one GPU (`config5_64_prompts`, or `config5_ragged` with --ragged).  The N > 1 line measures config 5; the N = 1 line's `value` is
    config 2 (one prompt at a time), so the curve of config 5 starts from THIS number, not from value(1).  None when the workload differs from the
    committed one (other preset / prompt count / step cap) or the file is absent."""
    try:
        if a.preset != "small" or n_prompts != 64 or a.n_semantic != 256 or a.scaling != "strong":
            return None
        import glob
        name = os.path.basename(sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_small_n1.json")))[-1])
        line = json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])
        leg = line["config5_ragged" if a.ragged else "config5_64_prompts"]
        return {"audio_s_per_s": leg["audio_s_per_s"], "prompts_per_s": leg["prompts_per_s"], "source": "committed reference: profiles/%s (an N = 1 run on one MI355X at the commit that file was added in, not measured by this job)" % name}
    except Exception:      # noqa: BLE001
        return None


def lock_step_roofline(ctx, slots: int, context: int = 640) -> dict:
    """`roofline` of an N > 1 line, measured on rank 0 after the timed region: one lock step of the coarse model (the stage a config-5 job spends 60 % of
    its wall in) over this rank's live slots at `context`, replayed from its hipGraph between two HIP events on the engine's stream.  Algorithmic bytes
    (SURVEY.md 8d per step, x the slots one launch chain serves): the model's weights ONCE for all slots + every slot's f32 K and V rows."""
    try:
        hp = ctx.hparams(1)
        _, b1 = ctx.time_decode_step(1, context, 8)                       # weights + ONE sequence's K / V rows at this context
        kv = 2.0 * context * hp["n_embd"] * hp["n_layer"] * 4.0
        nbytes = b1 + (slots - 1) * kv
        us = ctx.profile_lock_step(1, slots, context, 20)[-1]["us"]      # "step (graph replay)"
        return {"bound": "hbm", "unit_of_work": f"one lock step of the coarse model over {slots} live slots at context {context} (graph replay, HIP events on the engine's stream, rank 0)",
                "achieved": nbytes / (us * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": nbytes / (us * 1e-6) / 8e12,
                "us_per_step": us, "bytes_per_step": nbytes, "traffic": None,
                "traffic_note": "PMC passes cannot run inside a multi-rank job; the single-GPU line carries the decode step's counter traffic"}
    except Exception as e:      # noqa: BLE001
        return {"error": str(e)}


def ragged_caps(prompts, lo: int = 64, hi: int = 256):
    """Ragged form of config 5: the step cap of a prompt grows with its length, from `lo` for the shortest prompt of the set to `hi` for
    the longest (the synthetic weights never meet the stop rule, so the caps are where the utterances end: 1.3 - 5.1 s of audio) - the
    mix a server sees, where the equal-length job is the best case of a lock-step batch."""
    n = [len(p) for p in prompts]
    a, b = min(n), max(n)
    return [int(lo + round((hi - lo) * (k - a) / max(1, b - a))) for k in n]


def run_shard(ctx, prompts, idx, caps=None):
    """This rank's shard as ONE job of bark_hip_generate_batch[_ex] (the context's lock-step slots serve it: at most 64 utterances in
    flight, refilled from the queue); caps: per-prompt step caps (ragged job).  Returns the PCM arrays."""
    reqs = None if caps is None else [ctx.request_params(n_steps_text_encoder=caps[i]) for i in idx]
    res = ctx.generate_batch([prompts[i] for i in idx], params=reqs)
    for r in res:
        assert r is not None, "an utterance of the batch failed"
    return [r["pcm"] for r in res]


def cpu_baseline_leg(model_path: str, prompt: str, n_semantic: int) -> dict:
    """The CPU oracle (a restatement of the reference's algorithm, NOT the reference: ggml / encodec.cpp are absent) on 4 pinned cores
    like BASELINE config 1 (`-t 4`), on the SAME workload as the headline: same prompt, same n_steps_text_encoder (about 17 s of CPU on
    the GPU box's host).  `value` is the RTF of that run - no extrapolation."""
    from oracle.pyoracle import Oracle
    cores = min(os.cpu_count() or 4, 4)
    before = None
    try:
        before = os.sched_getaffinity(0)
        os.sched_setaffinity(0, set(range(cores)))              # taskset -c 0-3
    except (AttributeError, OSError):
        pass
    try:
        orc = Oracle(model_path, n_threads=cores)
        # the CPU baseline times the reference's algorithm in the CPU's own summation order: the parity mode of the oracle emulates the f16
        # matrix cores' accumulation for the fine model (order C1m), which costs a CPU five times the plain chains and is no baseline
        orc.set_fine_mfma(False)
        t1 = time.perf_counter()
        ref = orc.generate(prompt, orc.params(n_steps_text_encoder=n_semantic))
        cdt = time.perf_counter() - t1
        orc.close()
    finally:
        if before is not None:
            os.sched_setaffinity(0, before)                     # the caller's affinity comes back
    audio_s = ref["n_samples"] / 24000.0
    return {"value": audio_s / cdt, "unit": "audio-s/s", "cores": cores, "kind": "port",
            "sample": f"the headline workload itself: same prompt, n_steps_text_encoder={n_semantic} ({audio_s:.2f} s audio, {cdt:.1f} s CPU wall, "
                      f"threads pinned to cores 0-{cores - 1})",
            "label": "CPU restatement of the reference (oracle/, all three models on the CPU-friendly C1 chains: set_fine_mfma(False)), about 2x slower per token "
                     "than the reference's own README transcript (README.md:55, hardware unstated); a reported baseline, not a target",
            "stage_ms_per_token": {"semantic": ref["t_predict_semantic_us"] / 1000.0 / max(1, ref["n_sample_semantic"]),
                                   "coarse": ref["t_predict_coarse_us"] / 1000.0 / max(1, ref["n_sample_coarse"]),
                                   "fine": ref["t_predict_fine_us"] / 1000.0 / max(1, ref["n_sample_fine"])}}


FEW_SLOT_JOB_CHILD = r"""
import sys, json, time, hashlib
sys.path.insert(0, %r)
import numpy as np
from bark_amd_loader import load_package
pkg = load_package()
ctx = pkg.BarkContext.load_model(%r, pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=%d), 0)
slots, texts, ragged = %d, %r, %d
ctx.reserve_batch(slots)
reqs = None
if ragged:
    import bench
    reqs = [ctx.request_params(n_steps_text_encoder=c) for c in bench.ragged_caps(texts)]
res = ctx.generate_batch(texts, params=reqs)         # warm-up: graphs, allocations
t0 = time.perf_counter()
res = ctx.generate_batch(texts, params=reqs)
dt = time.perf_counter() - t0
h = hashlib.sha256()
for r in res:
    assert r is not None
    for k in ("semantic", "coarse", "fine", "pcm"):
        h.update(np.ascontiguousarray(r[k]).tobytes())
st = ctx.stats()
out = {"slots": slots, "prompts": len(texts), "prompts_per_s": len(texts) / dt, "sha256_ids_and_pcm": h.hexdigest(),
       "stage_ms": {k: st["t_%%s_us" %% k] / 1e3 for k in ("semantic", "coarse", "fine", "codec")}}
if not ragged:
    # where a lock step of the coarse model at context 640 spends its time under this arm: microseconds per launch site summed over the layers
    # (kernel + the gap in front, eager launches), and the graph-replayed step
    sites = {}
    for r in ctx.profile_lock_step(1, slots, 640, 10):
        sites[r["site"]] = round(sites.get(r["site"], 0.0) + r["us"], 1)
    out["lock_step_us_by_site"] = sites
print("RESULT " + json.dumps(out))
ctx.free()
"""


def few_slot_jobs_leg(path: str, prompts, n_semantic: int, deadline: float = float("inf")) -> dict:
    """Config 5's per-GPU share at N = 8 / N = 4: a job of 3 x slots prompts on a context with 8 / 16 lock-step slots (a process of its own per
    slot count: the slot count of a context is fixed by its first job), prompts/s, stage times and the per-site time line of one lock step.
    These are the points the strong-scaling curve of the 64-prompt job passes through."""
    import subprocess
    out = {}
    for slots in (8, 16):
        texts = [prompts[i % len(prompts)] for i in range(3 * slots)]
        key = "%d_slots" % slots
        if time.perf_counter() > deadline:
            out[key] = {"skipped": "time budget of the leg used up"}
            continue
        try:
            p = subprocess.run([sys.executable, "-c", FEW_SLOT_JOB_CHILD % (ROOT, path, n_semantic, slots, texts, 0)], env=dict(os.environ), capture_output=True, text=True, timeout=90)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            out[key] = json.loads(line[0][7:]) if (p.returncode == 0 and line) else {"error": "rc %d: %s" % (p.returncode, p.stderr[-300:])}
            out[key].pop("sha256_ids_and_pcm", None)
        except Exception as e:      # noqa: BLE001
            out[key] = {"error": str(e)}
    out["note"] = "3 x slots prompts as one job on a context with that many slots, second run timed"
    return out


def launch_ranks(n: int, all_on_device0: bool = False):
    """`python bench.py --gpus N` without a launcher: this process replaces itself by the driver's own N > 1 form,
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>`
    (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher).  Refuses (exit 2, nothing on stdout) when the node
    shows fewer than N GPUs - an N-GPU request must never turn into a line that says n_gpus = 1."""
    import socket
    if not all_on_device0:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write(f"bench.py: --gpus {n} but this node shows {have} GPU(s): refusing to run\n")
            sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--preset", default="small")
    ap.add_argument("--n-semantic", type=int, default=256)
    ap.add_argument("--n-prompts", type=int, default=64, help="size of the synthetic prompt set of config 5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true")
    ap.add_argument("--no-q4", action="store_true")
    ap.add_argument("--no-large", action="store_true", help="skip the bark-large leg (BASELINE config 3)")
    ap.add_argument("--no-fast", action="store_true", help="skip the tolerance-route leg (BARK_HIP_FAST_GEMM=1)")
    ap.add_argument("--no-few-slots", action="store_true", help="skip the 8- / 16-slot jobs (config 5's per-GPU share at N = 8 / N = 4; separate processes)")
    ap.add_argument("--no-roofline-legs", action="store_true", help="skip the kernel timing legs (rocprofv3 passes: the statistics then hold the prompts' kernels only)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the --n-prompts job split over the ranks (BASELINE config 5); weak = --n-prompts per rank")
    ap.add_argument("--no-weak-leg", action="store_true", help="N > 1, strong scaling: skip the weak-scaling leg (a 64-prompt job per rank) reported beside it")
    ap.add_argument("--ragged", action="store_true", help="N > 1: the ragged job (bench.ragged_caps: step caps 64..256 by prompt length) instead of equal caps")
    ap.add_argument("--dump-pcm", default=None, help="rank 0 writes the gathered PCM of the last step here (.npz; tests)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for the 1-GPU dry run)")
    ap.add_argument("--all-ranks-on-device0", action="store_true", help="dry run of the N > 1 path on a single GPU (with --backend gloo)")
    a = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and a.gpus > 1:
        launch_ranks(a.gpus, a.all_ranks_on_device0)            # does not return: `python bench.py --gpus N` IS the N-rank job
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != a.gpus:
        # never print an N = 1 line for an N-GPU request (or the reverse): the driver computes the scaling curve from n_gpus / value
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: refusing to run (launch with --nproc-per-node {a.gpus}, or drop the launcher: "
                         f"`python bench.py --gpus {a.gpus}` starts its own ranks)\n")
        sys.exit(2)
    if a.all_ranks_on_device0:
        local_rank = 0
    os.environ["BARK_HIP_DEVICE"] = str(local_rank)

    import numpy as np
    import torch
    import torch.distributed as dist
    from bark_amd_loader import load_package
    from tools.make_synth_model import ensure_model

    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(a.backend)       # "nccl" is RCCL on ROCm
    pkg = load_package()
    if rank == 0:
        path = ensure_model(a.preset, 0)
    if world > 1:
        dist.barrier()
    path = ensure_model(a.preset, 0)
    params = pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=a.n_semantic)
    ctx = pkg.BarkContext.load_model(path, params, seed=0)
    prompts = synth_prompts(a.n_prompts)
    coll_dev = "cuda" if (world > 1 and a.backend == "nccl") else None

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize() if torch.cuda.is_available() else None

    if world > 1:
        # ---------------------------------------------------------------------------------------- config 5
        if a.scaling == "weak":
            # every rank runs its own copy of the job: prompt set of rank r = synth_prompts(seed r); the gather carries world x n_prompts utterances
            prompts = [t for r in range(world) for t in synth_prompts(a.n_prompts, seed=r)]
            idx = list(range(rank * a.n_prompts, (rank + 1) * a.n_prompts))
            idx.sort(key=lambda i: (len(prompts[i]), i))
        else:
            idx = shard_prompts(prompts, rank, world)
        caps = ragged_caps(prompts) if a.ragged else None
        gathered = {}
        for _ in range(a.warmup):
            gather_batch_results(run_shard(ctx, prompts, idx, caps), idx, len(prompts), rank, world, coll_dev)
        sync_all()
        t0 = time.perf_counter()
        audio_s = 0.0
        t_gen = t_gather = 0.0
        for _ in range(a.steps):
            t1 = time.perf_counter()
            pcms = run_shard(ctx, prompts, idx, caps)
            t2 = time.perf_counter()
            counts, gathered = gather_batch_results(pcms, idx, len(prompts), rank, world, coll_dev)
            t3 = time.perf_counter()
            t_gen += t2 - t1; t_gather += t3 - t2
            audio_s += sum(len(p) for p in pcms) / 24000.0
        sync_all()
        dt = time.perf_counter() - t0
        dt, audio_total = reduce_timing(dt, audio_s, world, device=coll_dev)
        # per-rank wall of the generation and of the edge collectives (seconds per step), gathered for the line
        per_rank = torch.tensor([t_gen / max(1, a.steps), t_gather / max(1, a.steps)], dtype=torch.float64, device=coll_dev)
        all_ranks = [torch.zeros_like(per_rank) for _ in range(world)]
        dist.all_gather(all_ranks, per_rank)
        # the weak-scaling point of the same path beside the strong one (per-GPU work fixed: a 64-prompt job of its own on EVERY rank, no gather of PCM -
        # "no data-path collective"): what a node serving independent batches sees.  A context of its own: the slot count is fixed by a context's first job
        weak = None
        if a.scaling == "strong" and not a.no_weak_leg and not a.ragged:
            wctx = pkg.BarkContext.load_model(path, params, seed=0)
            wprompts = synth_prompts(a.n_prompts, seed=rank)
            widx = sorted(range(len(wprompts)), key=lambda i: (len(wprompts[i]), i))
            run_shard(wctx, wprompts, widx)
            sync_all()
            tw = time.perf_counter()
            wpcm = run_shard(wctx, wprompts, widx)
            sync_all()
            wdt, waudio = reduce_timing(time.perf_counter() - tw, sum(len(p) for p in wpcm) / 24000.0, world, device=coll_dev)
            weak = {"scaling": "weak", "prompts_per_rank": len(wprompts), "audio_s_per_s": waudio / wdt, "prompts_per_s": len(wprompts) * world / wdt,
                    "wall_ms": wdt * 1e3, "note": "every rank runs its own %d-prompt lock-step job (MAX wall over ranks, SUM of audio); no collective on the data path" % len(wprompts)}
            wctx.free()
        if rank == 0:
            roof = lock_step_roofline(ctx, min(64, len(idx)))
            assert len(gathered) == len(prompts) and int(counts.sum()) == sum(len(v) for v in gathered.values())
            if a.dump_pcm:
                np.savez(a.dump_pcm, **{"p%03d" % i: v for i, v in gathered.items()})
            out = {
                "metric": "audio-sec/sec (RTF), bark-small f16 greedy", "value": audio_total / dt, "unit": "audio-s/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1000.0 * dt / max(1, a.steps),
                "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f16 (f32 accumulate)", "data": "synthetic",
                "config": {"workload": f"BASELINE config 5: bark-{a.preset} f16, {len(prompts)} synthetic prompts "
                                       + ("(weak scaling: %d per rank) " % a.n_prompts if a.scaling == "weak" else "sorted by length and dealt snake-wise, ")
                                       + f"{len(idx)} per rank as ONE lock-step job of bark_hip_generate_batch per rank (up to 64 slots), "
                                       + ("ragged step caps 64..256 by prompt length" if a.ragged else f"n_steps_text_encoder={a.n_semantic}")
                                       + "; all_gather of sample counts + gather of PCM on rank 0 inside the timed region",
                           "prompts_per_step": len(prompts), "audio_s_per_step": audio_total / max(1, a.steps)},
                "prompts_per_s": len(prompts) * a.steps / dt,
                "per_rank_s_per_step": {"generate": [float(x[0]) for x in all_ranks], "gather": [float(x[1]) for x in all_ranks]},
                "weak_scaling_leg": weak,
                "roofline": roof, "note": "cpu_baseline is reported by the N = 1 run (rank 0 at N = 1 only, bench contract)",
                "n1_point_of_this_workload": n1_reference(a, len(prompts)),
                "scaling_reference": "the N = 1 point of THIS workload is `config5_64_prompts.audio_s_per_s` of the N = 1 line (the 64-prompt job on one GPU), "
                                     "not its `value`: the N = 1 line's `value` is BASELINE config 2 (one prompt at a time, latency mode), so value(N) / (N x value(1)) "
                                     "compares two workloads",
            }
            print(json.dumps(out))
        ctx.free()
        dist.destroy_process_group()
        return

    # -------------------------------------------------------------------------------------------- N = 1: config 2 headline
    def one_step(i):
        text = prompt_for(i, rank, world, prompts)
        ok = ctx.generate_audio(text)
        assert ok
        return ctx.stats()

    for i in range(a.warmup):
        one_step(i)
    sync_all()
    t0 = time.perf_counter()
    audio_s = 0.0
    agg = {"t_semantic_us": 0, "t_coarse_us": 0, "t_fine_us": 0, "t_codec_us": 0, "n_sample_semantic": 0, "n_sample_coarse": 0,
           "n_sample_fine": 0, "n_near_tie": 0}
    for i in range(a.steps):
        st = one_step(a.warmup + i)
        audio_s += st["n_samples"] / 24000.0
        for k in agg:
            agg[k] += st[k]
    sync_all()
    dt = time.perf_counter() - t0

    out = {
        "metric": "audio-sec/sec (RTF), bark-small f16 greedy", "value": audio_s / dt, "unit": "audio-s/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1000.0 * dt / max(1, a.steps),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16 (f32 accumulate)",
        "data": "synthetic",
        "config": {"workload": f"BASELINE config 2: bark-{a.preset} f16 on 1xMI355X, single prompt per step, greedy, hipGraph decode, "
                               f"n_steps_text_encoder={a.n_semantic}", "prompts_per_step": world,
                   "audio_s_per_prompt": audio_s / max(1, a.steps)},
        "stage_ms_per_token": {
            "semantic": agg["t_semantic_us"] / 1000.0 / max(1, agg["n_sample_semantic"]),
            "coarse": agg["t_coarse_us"] / 1000.0 / max(1, agg["n_sample_coarse"]),
            "fine": agg["t_fine_us"] / 1000.0 / max(1, agg["n_sample_fine"]),
            "codec_ms": agg["t_codec_us"] / 1000.0 / max(1, a.steps)},
        "samples_settled_by_exact_path": agg["n_near_tie"],
        "prompts_per_s": world * a.steps / dt,
    }
    # roofline.  Dominant kernel by time: gemv_ln_wg_kernel (LayerNorm-fused decode GEMV: QKV, FC and LM head = 25 of the 62 launches
    # of a step, ~37 % of it; profiles/).  Its largest per-layer instance is the FC product (4 E^2 f16 weights = 4.72 MB per launch).
    # `achieved` = algorithmic bytes / average launch duration, measured here with HIP events on the engine's stream over a graph of
    # 48 launches that rotate through the layers' weights.  `traffic`: HBM bytes from PMC counters need a separate rocprofv3 pass
    # (they cannot be read inside this run): null here; profiles/r02_pmc_gemv_fc.json is the PMC summary of this kernel, regenerated
    # on the final build by the same gpurun call as the committed bench line (tools/collect_profiles.sh).
    try:
        if a.no_roofline_legs:
            raise RuntimeError("timing legs switched off (--no-roofline-legs)")
        # The headline is 92 % semantic + coarse decode steps, so `roofline` IS the decode step, priced with the driver-reproducible time:
        # algorithmic bytes of one semantic step at the workload's mean context (weights + KV rows read, from the engine's own count)
        # divided by stage_ms_per_token.semantic of the timed region above.  `kernel` is the dominant kernel's micro-loop (FC product: HIP
        # events on the engine's stream over a graph of 48 launches rotating through the layers' weights), `step_microloop` the same for a
        # whole step at context 640.  `traffic`: HBM bytes per step from the committed rocprofv3 PMC summaries (FETCH_SIZE x 2 on gfx950 +
        # WRITE_SIZE, separate passes over tools/profile_decode.py, which runs steps at context 640).
        mean_ctx = 257 + a.n_semantic // 2
        _, step_bytes = ctx.time_decode_step(0, mean_ctx, 8)
        step_us = 1000.0 * out["stage_ms_per_token"]["semantic"]
        us, nbytes = ctx.time_gemv(0, 2, 2400)
        pmc = {}
        for name in ("r06_pmc_decode_step.json", "r05_pmc_decode_step.json", "r04_pmc_decode_step.json"):
            f = os.path.join(ROOT, "profiles", name)
            if os.path.exists(f):
                pmc = json.load(open(f)); pmc["file"] = "profiles/" + name
                break
        mus, mbytes = ctx.time_decode_step(0, 640, 320)
        out["roofline"] = {"bound": "hbm", "unit_of_work": f"one semantic decode step (62 kernels, eight steps per hipGraph) at the workload's mean context {mean_ctx}",
                           "achieved": step_bytes / (step_us * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": step_bytes / (step_us * 1e-6) / 8e12,
                           "us_per_step": step_us, "bytes_per_step": step_bytes,
                           "time_source": "stage_ms_per_token.semantic of this run's timed region (host wall clock of the stage / sampled tokens: includes the prompt pass and the polls)",
                           "traffic": pmc.get("step_traffic_bytes"),
                           "traffic_note": "HBM-side bytes per decode step at context %s from the committed PMC summary %s (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate rocprofv3 passes; counters cannot be read inside this run): %.2f x the algorithmic bytes at that context; null when absent"
                                           % (pmc.get("context", 640), pmc.get("file"), pmc.get("step_traffic_over_algorithmic", float("nan"))),
                           "step_microloop": {"ctx": 640, "us_per_step": mus, "bytes_per_step": mbytes, "GB/s": mbytes / (mus * 1e-6) / 1e9, "frac": mbytes / (mus * 1e-6) / 8e12},
                           "kernel": {"name": "gemv_ln_wg_kernel<6> (LayerNorm + FC 3072x768 f16 + GELU, decode): dominant kernel by time (25 of 62 launches with its QKV / LM-head instances)",
                                      "us_per_launch": us, "bytes_per_launch": nbytes, "GB/s": nbytes / (us * 1e-6) / 1e9, "frac": nbytes / (us * 1e-6) / 8e12,
                                      "traffic": pmc.get("traffic_bytes_per_launch")}}
        gem = {}
        for op, name in enumerate(("ln_qkv_partial_scores", "attn_proj", "ln_fc_gelu", "mlp_proj")):
            u, nb = ctx.time_gemv(0, op, 1200)
            gem[name] = {"us": u, "GB/s": nb / (u * 1e-6) / 1e9}
        out["roofline_gemv_variants"] = gem
        # fine forward pass, default order C1: products and attention on the f32 matrix cores
        fus, flops = ctx.time_fine_pass(6)
        att_flops = 12 * 12 * 2 * 2 * 1024 * 1024 * 64 if a.preset == "small" else None
        out["roofline_fine_pass"] = {"bound": "mfma", "achieved": flops / (fus * 1e-6) / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": flops / (fus * 1e-6) / 157.3e12, "us_per_pass": fus,
                                     "peak_note": "one window through the single-utterance path = the DEFAULT arithmetic: products (%.0f %% of the flops) in the reference order C1 and the attention "
                                                  "(C2 / C4e / C5) both on the f32 matrix cores (v_mfma_f32_32x32x2_f32, dense peak 157.3 TFLOP/s); `eight_windows_side_by_side` is the pass of a "
                                                  "lock-step job: products in C1m on the f16 matrix cores (peak 2500), attention on the f32 cores" % (100.0 * (1 - att_flops / flops) if att_flops else 0),
                                     "frac_of_f32_mfma_peak_157TF": flops / (fus * 1e-6) / 157.3e12}
        # north_star's phrasing ("HBM roofline on the fine-model forward"), SURVEY.md 8(d): algorithmic bytes of one pass = the fine model's weights
        # + the window ids in + the picks out, over the pass time and the 8 TB/s datasheet rate.  The pass is MFMA-bound (arithmetic intensity
        # flops / bytes ~ 1.2 k FLOP/B against a ridge of ~ 312): the >= 40 % HBM target cannot apply to it, the MFMA fractions above are the binding ones
        hp2 = ctx.hparams(2)
        fine_bytes = None
        if hp2["ftype"] == 1:            # f16 file: 2 bytes per weight (171.5 MB at bark-small)
            fine_bytes = 2.0 * (12.0 * hp2["n_layer"] * hp2["n_embd"] ** 2 + hp2["n_out"] * hp2["n_embd"]) + 8 * 1024 * 4 + 1024 * 4
        if fine_bytes:
            out["roofline_fine_pass"].update({"hbm_algorithmic_bytes": fine_bytes, "hbm_achieved_GBps": fine_bytes / (fus * 1e-6) / 1e9,
                                              "hbm_frac": fine_bytes / (fus * 1e-6) / 8.0e12, "arithmetic_intensity_flop_per_byte": flops / fine_bytes,
                                              "hbm_target_note": "north_star's >= 40 % HBM roofline on the fine forward is inapplicable: at ~1.2 k FLOP/B the pass is bound by the matrix cores, "
                                                                 ""
                                                                 "not by memory (ridge ~20 FLOP/B at the f32 matrix cores' 157 TFLOP/s over 8 TB/s, ~312 at the f16 cores' 2.5 PFLOP/s); hbm_frac is reported for completeness"})
        # the pass a lock-step batch runs: the fine windows of 8 utterances side by side (engine_fine_many)
        fus8, flops8 = ctx.time_fine_pass(3, 8)
        out["roofline_fine_pass"]["eight_windows_side_by_side"] = {"order": "C1m (lock-step jobs)", "us_per_window": fus8 / 8, "achieved": flops8 / (fus8 * 1e-6) / 1e12,
                                                                   "frac_of_f16_mfma_peak_2500TF": flops8 / (fus8 * 1e-6) / 2.5e15}
    except Exception as e:      # noqa: BLE001
        out["roofline"] = {"error": str(e)}
    # config 5 at N = 1: the 64-prompt job on this GPU as ONE lock-step job on 64 slots (the point the multi-GPU curve starts from), and its
    # ragged form (step caps 64..256 by prompt length: what a server sees; the equal-length job is the best case of a lock-step batch)
    if not a.no_batched:
        try:
            bctx = pkg.BarkContext.load_model(path, params, seed=0)
            idx = shard_prompts(prompts, 0, 1)
            caps = ragged_caps(prompts)
            run_shard(bctx, prompts, idx)                            # warm-up: graph capture, allocations
            tb = time.perf_counter()
            pcms = run_shard(bctx, prompts, idx)
            counts, _ = gather_batch_results(pcms, idx, len(prompts), 0, 1)
            dtb = time.perf_counter() - tb
            stb = bctx.stats()
            out["config5_64_prompts"] = {"prompts_per_s": len(prompts) / dtb, "audio_s_per_s": float(counts.sum()) / 24000.0 / dtb,
                                         "wall_ms": dtb * 1e3, "slots": min(64, len(prompts)),
                                         "stage_ms": {k: stb["t_%s_us" % k] / 1e3 for k in ("semantic", "coarse", "fine", "codec")},
                                         "note": "bark_hip_generate_batch: lock-step decode, window prompts of all slots in one pass, fine windows of 8 utterances side by side, "
                                                 "codec of all utterances in one pass; fine products in C1m (the jobs' order: bark_hip_set_fine_order); per-utterance results are those of bark_generate_audio "
                                                 "bit for bit in the semantic and coarse ids always, and in fine ids / PCM when the single path runs the same order"}
            run_shard(bctx, prompts, idx, caps)                      # warm-up of the ragged job (graphs of the shrinking slot counts)
            tb = time.perf_counter()
            pcms = run_shard(bctx, prompts, idx, caps)
            dtr = time.perf_counter() - tb
            str_ = bctx.stats()
            out["config5_ragged"] = {"prompts_per_s": len(prompts) / dtr, "audio_s_per_s": sum(len(p) for p in pcms) / 24000.0 / dtr, "wall_ms": dtr * 1e3,
                                     "slots": min(64, len(prompts)), "step_caps": "64..256 by prompt length (bench.ragged_caps), mean %.0f" % (sum(caps) / len(caps)),
                                     "stage_ms": {k: str_["t_%s_us" % k] / 1e3 for k in ("semantic", "coarse", "fine", "codec")},
                                     "parity": "tests/test_gpu_batch_ragged.py (randomised ragged jobs against the oracle; 16 of these 64 utterances against committed oracle outputs)"}
            # the request collector under load: 256 requests of the same shape at once from 16 host threads, jobs of up to 64, TWO job streams
            # (bark_hip_batcher_create_ex: a second worker on a clone of the context - two decode chains share the chip)
            try:
                import threading
                col = pkg.Batcher(bctx, max_batch=64, max_wait_ms=20, streams=2)
                def client(k, n):
                    for t in [col.submit(prompts[i % len(prompts)]) for i in range(k, n, 16)]:
                        col.wait(t)
                for n_req in (128, 256):                                 # first pass: warm-up (the clone, its graphs)
                    th = [threading.Thread(target=client, args=(k, n_req)) for k in range(16)]
                    tb = time.perf_counter()
                    for t in th: t.start()
                    for t in th: t.join()
                    dtc = time.perf_counter() - tb
                out["config5_request_collector"] = {"requests": 256, "job_streams": 2, "max_batch": 64, "requests_per_s": 256 / dtc, "wall_ms": dtc * 1e3,
                                                    "note": "native request collector (batcher.hip) in front of the lock-step jobs; one job stream: profiles/r04_batcher_load.txt"}
                col.free()
            except Exception as e:      # noqa: BLE001
                out["config5_request_collector"] = {"error": str(e)}
            bctx.free()
        except Exception as e:      # noqa: BLE001
            out["config5_64_prompts"] = {"error": str(e)}
    # Which arithmetic the headline ran in, and the other order of the fine model's products beside it (round-5 review item 4).  The timed region above is
    # bark_generate_audio in its DEFAULT arithmetic: every weight product of the three GPT stages in the restated reference order C1 (fine model included,
    # f32 matrix cores), attention C2 / C4e / C5 - ids bit-equal to the CPU restatement of the reference (tests: test_bench_workload_matches_the_oracle,
    # prompts 1, 5, 24) - and the codec's convolutions in the f16 matrix cores' order C9m (PCM within the stated tolerance of C9: >= 55 dB, measured 63 - 66).
    # `fine_order_c1m` = the same prompt with the fine products in the f16 matrix cores' accumulation order (what lock-step jobs use; bark_hip_set_fine_order 2):
    # RTF, fine-pass time and how many fine ids agree with the default run - measured here, on the device, at bark-small.  `codec_c9`: the codec in C9.
    out["arithmetic"] = {"gpt_products": "C1 (restated reference order) for all three models", "attention": "C2 / C4e / C5", "codec": "C9m (f16 matrix cores)",
                         "ids": "bit-equal to oracle/ (the CPU restatement of the reference) on this workload", "pcm": "bit-equal to the oracle's C9m mode; vs C9: SNR >= 55 dB (tests/test_order_divergence.py)"}
    try:
        text = prompts[a.warmup % len(prompts)]
        assert ctx.generate_audio(text)
        ref_fine, ref_pcm = ctx.fine_tokens().copy(), ctx.audio_data().copy()
        f_c1, _ = ctx.time_fine_pass(6)
        ctx.set_fine_order(2)
        try:
            ctx.generate_audio(text)
            t5 = time.perf_counter(); assert ctx.generate_audio(text); d5 = time.perf_counter() - t5
            m_fine, m_pcm = ctx.fine_tokens().copy(), ctx.audio_data().copy()
            f_c1m, _ = ctx.time_fine_pass(6)
        finally:
            ctx.set_fine_order(0)
        err = (m_pcm.astype(np.float64) - ref_pcm.astype(np.float64))
        out["fine_order_c1m"] = {"switch": "bark_hip_set_fine_order(ctx, 2) / BARK_HIP_FINE_ORDER=c1m (the default inside lock-step jobs)",
                                 "rtf": ctx.stats()["n_samples"] / 24000.0 / d5, "ms_per_prompt": d5 * 1e3, "fine_pass_us": f_c1m, "fine_pass_us_default_c1": f_c1,
                                 "fine_ids_equal_to_the_default_run": "%d / %d" % (int(np.sum(m_fine == ref_fine)), int(ref_fine.size)),
                                 "fine_ids_equal_frac": float(np.mean(m_fine == ref_fine)),
                                 "pcm_snr_db_vs_default": float(10 * np.log10((ref_pcm.astype(np.float64) ** 2).sum() / max(float((err ** 2).sum()), 1e-30)))}
    except Exception as e:      # noqa: BLE001
        out["fine_order_c1m"] = {"error": str(e)}
    # Tolerance route (BARK_HIP_FAST_GEMM=1, read when a context is loaded): the many-row products and the fine model's attention on the
    # f16 matrix cores in hardware accumulation order (fast_kernels.hip).  Reported BESIDE the canonical numbers, never instead of them:
    # its ids are not promised to be the oracle's (tests/test_gpu_parity.py bounds its logits against the canonical route).
    if not a.no_fast:
        try:
            os.environ["BARK_HIP_FAST_GEMM"] = "1"
            try:
                fctx = pkg.BarkContext.load_model(path, params, seed=0)
            finally:
                del os.environ["BARK_HIP_FAST_GEMM"]
            text = prompts[a.warmup % len(prompts)]
            assert ctx.generate_audio(text)
            ref_ids = (ctx.semantic_tokens().copy(), ctx.coarse_tokens().copy(), ctx.fine_tokens().copy())
            fctx.generate_audio(text)
            tf0 = time.perf_counter(); assert fctx.generate_audio(text); dtf = time.perf_counter() - tf0
            got_ids = (fctx.semantic_tokens(), fctx.coarse_tokens(), fctx.fine_tokens())
            agree = {k: (int(np.sum(x.ravel()[:min(x.size, y.size)] == y.ravel()[:min(x.size, y.size)])), int(x.size))
                     for k, x, y in zip(("semantic", "coarse", "fine"), ref_ids, got_ids)}
            f1, fl1 = fctx.time_fine_pass(6)
            f8, fl8 = fctx.time_fine_pass(3, 8)
            leg = {"switch": "BARK_HIP_FAST_GEMM=1", "rtf": fctx.stats()["n_samples"] / 24000.0 / dtf, "ms_per_prompt": dtf * 1e3,
                   "ids_equal_to_the_canonical_run_of_the_same_prompt": {k: "%d / %d" % v for k, v in agree.items()},
                   "fine_pass": {"bound": "mfma-f16", "us_per_pass": f1, "achieved": fl1 / (f1 * 1e-6) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                                 "frac": fl1 / (f1 * 1e-6) / 2.5e15,
                                 "eight_windows_side_by_side": {"us_per_window": f8 / 8, "achieved": fl8 / (f8 * 1e-6) / 1e12, "frac": fl8 / (f8 * 1e-6) / 2.5e15}},
                   "tolerance": "logits within 5e-3 of the canonical route (fine model 1e-2), checked by test_tolerance_route_stays_within_its_stated_tolerance"}
            if not a.no_batched:
                idx = shard_prompts(prompts, 0, 1)
                run_shard(fctx, prompts, idx)
                tb = time.perf_counter(); pcms = run_shard(fctx, prompts, idx); dtb = time.perf_counter() - tb
                leg["config5_64_prompts"] = {"prompts_per_s": len(prompts) / dtb, "audio_s_per_s": sum(len(p) for p in pcms) / 24000.0 / dtb, "wall_ms": dtb * 1e3, "slots": min(64, len(prompts))}
            out["tolerance_route"] = leg
            fctx.free()
        except Exception as e:      # noqa: BLE001
            out["tolerance_route"] = {"error": str(e)}
    # the reference's default step cap (n_steps_text_encoder = 768, bark.cpp:2212): 1154 frames = 15.4 s per prompt, two fine windows
    try:
        ctx.set_params(pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=768))
        ctx.generate_audio(prompts[3])
        t7 = time.perf_counter(); assert ctx.generate_audio(prompts[4]); d7 = time.perf_counter() - t7
        s7 = ctx.stats()
        out["default_cap_768_steps"] = {"rtf": s7["n_samples"] / 24000.0 / d7, "ms_per_prompt": d7 * 1e3, "audio_s": s7["n_samples"] / 24000.0,
                                        "n_frames": s7["n_frames"]}
        ctx.set_params(params)
    except Exception as e:      # noqa: BLE001
        out["default_cap_768_steps"] = {"error": str(e)}
    # BASELINE config 3: bark-large shapes (1024 wide, 24 layers, 16 heads; reference shapes convert.py:86-110), single prompt, greedy
    if not a.no_large:
        try:
            lpath = ensure_model("large", 0)
            lctx = pkg.BarkContext.load_model(lpath, params, seed=0)
            lctx.generate_audio(prompts[0])
            tl = time.perf_counter(); la = 0.0
            lagg = {"t_semantic_us": 0, "t_coarse_us": 0, "t_fine_us": 0, "t_codec_us": 0, "n_sample_semantic": 0, "n_sample_coarse": 0, "n_sample_fine": 0}
            for i in range(2):
                assert lctx.generate_audio(prompts[1 + i])
                ls = lctx.stats(); la += ls["n_samples"] / 24000.0
                for k in lagg:
                    lagg[k] += ls[k]
            dtl = time.perf_counter() - tl
            dus, dbytes = lctx.time_decode_step(0, 640, 160)
            fus, flops = lctx.time_fine_pass(3)
            out["bark_large"] = {"config": "BASELINE config 3: bark-large f16 on 1xMI355X, single prompt, greedy, n_steps_text_encoder=%d" % a.n_semantic,
                                 "rtf": la / dtl, "ms_per_prompt": dtl * 500.0,
                                 "stage_ms_per_token": {"semantic": lagg["t_semantic_us"] / 1000.0 / max(1, lagg["n_sample_semantic"]),
                                                        "coarse": lagg["t_coarse_us"] / 1000.0 / max(1, lagg["n_sample_coarse"]),
                                                        "fine": lagg["t_fine_us"] / 1000.0 / max(1, lagg["n_sample_fine"]), "codec_ms": lagg["t_codec_us"] / 2000.0},
                                 "decode_step_us": dus, "decode_step_GB/s": dbytes / (dus * 1e-6) / 1e9, "decode_step_hbm_frac": dbytes / (dus * 1e-6) / 8e12,
                                 "fine_pass_us": fus, "fine_pass_TFLOP/s": flops / (fus * 1e-6) / 1e12, "fine_pass_frac_of_f32_mfma_peak": flops / (fus * 1e-6) / 157.3e12,
                                 "parity": "tests/test_gpu_parity.py::test_large_model_shapes (64- and 256-step oracle fixtures of this model file)"}
            lctx.free()
        except Exception as e:      # noqa: BLE001
            out["bark_large"] = {"error": str(e)}
    # BASELINE config 4: the same model quantised to q4_0 by the native bark_model_quantize (reported beside the headline)
    if not a.no_q4:
        try:
            qpath = path[:-4] + "_q4_0.bin"
            if not os.path.exists(qpath):
                assert pkg.load_library().bark_model_quantize(path.encode(), (qpath + ".tmp").encode(), 2)
                os.replace(qpath + ".tmp", qpath)
            qctx = pkg.BarkContext.load_model(qpath, params, seed=0)
            qctx.generate_audio(prompts[0])
            tq = time.perf_counter(); qa = 0.0
            for i in range(2):
                assert qctx.generate_audio(prompts[1 + i]); qa += qctx.stats()["n_samples"] / 24000.0
            dtq = time.perf_counter() - tq
            dus, dbytes = qctx.time_decode_step(0, 640, 320)
            fus, flops = qctx.time_fine_pass(6)
            out["q4_0"] = {"rtf": qa / dtq, "ms_per_prompt": dtq * 500.0, "file_MB": os.path.getsize(qpath) / 1e6,
                           "decode_step_us": dus, "decode_step_GB/s": dbytes / (dus * 1e-6) / 1e9,
                           "fine_pass_us": fus, "fine_pass_equiv_TFLOP/s": flops / (fus * 1e-6) / 1e12,
                           "note": "q4_0 x q8_0 block products: v_dot4 GEMV (decode), v_mfma_i32_32x32x32_i8 (prefill / fine); bit-exact vs the oracle"}
            qctx.free()
        except Exception as e:      # noqa: BLE001
            out["q4_0"] = {"error": str(e)}
    if not a.no_batched and not a.no_few_slots:
        try:
            out["config5_per_gpu_share_at_8_and_4_gpus"] = few_slot_jobs_leg(path, prompts, a.n_semantic, time.perf_counter() + 100.0)
        except Exception as e:      # noqa: BLE001
            out["config5_per_gpu_share_at_8_and_4_gpus"] = {"error": str(e)}
    if not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_leg(path, prompts[a.warmup % len(prompts)], a.n_semantic)
    print(json.dumps(out))
    ctx.free()


if __name__ == "__main__":
    main()
