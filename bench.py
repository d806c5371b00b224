#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X Bark engine.

Metric (BASELINE.json): audio-sec/sec (real-time factor) of bark_generate_audio on bark-small f16,
greedy (temp = fine_temp = 0), synthetic weights / synthetic prompts (no checkpoints offline),
n_steps_text_encoder = 256 -> 256 semantic + 768 coarse tokens, 6 fine passes, 384 frames = 5.12 s of
audio per prompt.  A "step" = one bark_generate_audio call on every rank (weak scaling:
one prompt per rank per step, ranks = independent replicas, no collective on the data path).

  python bench.py [--gpus N --steps K --warmup W]     (N > 1: launched under torch.distributed.run)

Prints ONE JSON line on rank 0 with `roofline` (decode step vs HBM) and `cpu_baseline` (the CPU
oracle timed on this host on a bounded sample).
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
    """Replica-per-GPU batch split: at every step rank r takes prompt (step * world + r); ranks never exchange data."""
    return prompts[(step * world + rank) % len(prompts)]


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--preset", default="small")
    ap.add_argument("--n-semantic", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true")
    ap.add_argument("--no-q4", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for the 1-GPU dry run)")
    ap.add_argument("--all-ranks-on-device0", action="store_true", help="dry run of the N > 1 path on a single GPU (with --backend gloo)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
    ctx = pkg.BarkContext.load_model(path, pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=a.n_semantic), seed=0)
    prompts = synth_prompts(64)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize() if torch.cuda.is_available() else None

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
    dt, audio_total = reduce_timing(dt, audio_s, world, device="cuda" if (world > 1 and a.backend == "nccl") else None)

    if rank == 0:
        out = {
            "metric": "audio-sec/sec (RTF), bark-small f16 greedy", "value": audio_total / dt, "unit": "audio-s/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1000.0 * dt / max(1, a.steps),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"bark-{a.preset} f16 on 1xMI355X per rank, single prompt per step, greedy, hipGraph decode, "
                                   f"n_steps_text_encoder={a.n_semantic}", "prompts_per_step": world,
                       "audio_s_per_prompt": audio_s / max(1, a.steps)},
            "stage_ms_per_token": {
                "semantic": agg["t_semantic_us"] / 1000.0 / max(1, agg["n_sample_semantic"]),
                "coarse": agg["t_coarse_us"] / 1000.0 / max(1, agg["n_sample_coarse"]),
                "fine": agg["t_fine_us"] / 1000.0 / max(1, agg["n_sample_fine"]),
                "codec_ms": agg["t_codec_us"] / 1000.0 / max(1, a.steps)},
            "near_ties": agg["n_near_tie"],
            "prompts_per_s": world * a.steps / dt,
        }
        # roofline.  Dominant kernel by time: gemv_kernel (decode GEMV, ~45 % of a generate call, profiles/);
        # its largest instance is the LayerNorm-fused FC GEMV (4 E^2 f16 weights = 4.72 MB per launch).
        try:
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "r01_pmc_gemv_fc.json")
            if os.path.exists(tpath):
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            us, nbytes = ctx.time_gemv(0, 2, 2400)
            out["roofline"] = {"bound": "hbm", "kernel": "gemv_kernel<6,LN> (LayerNorm + FC 3072x768 f16 + GELU, decode)",
                               "achieved": nbytes / (us * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": nbytes / (us * 1e-6) / 8e12,
                               "traffic": traffic, "us_per_launch": us, "bytes_per_launch": nbytes}
            gem = {}
            for op, name in enumerate(("ln_qkv", "attn_proj", "ln_fc_gelu", "mlp_proj")):
                u, nb = ctx.time_gemv(0, op, 1200)
                gem[name] = {"us": u, "GB/s": nb / (u * 1e-6) / 1e9}
            out["roofline_gemv_variants"] = gem
            us, nbytes = ctx.time_decode_step(0, 640, 300)
            out["roofline_decode_step"] = {"bound": "hbm", "unit_of_work": "one semantic decode step @ctx 640 (hipGraph: 62 kernels)",
                                           "achieved": nbytes / (us * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                           "frac": nbytes / (us * 1e-6) / 8e12, "us_per_step": us, "bytes_per_step": nbytes}
            fus, flops = ctx.time_fine_pass(6)
            out["roofline_fine_pass"] = {"bound": "mfma-f32", "achieved": flops / (fus * 1e-6) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                                         "frac": flops / (fus * 1e-6) / 157.3e12, "us_per_pass": fus,
                                         "hbm_frac_on_algorithmic_bytes": 171.5e6 / (fus * 1e-6) / 8e12}
        except Exception as e:      # noqa: BLE001
            out["roofline"] = {"error": str(e)}
        # in-engine batching: 8 utterances in lock step on this GPU (reported beside the
        # single-prompt headline, never instead of it)
        if world == 1 and not a.no_batched:
            try:
                bctx = pkg.BarkContext.load_model(path, pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=a.n_semantic), seed=0)
                bctx.generate_batch(prompts[:8])
                tb = time.perf_counter()
                res = bctx.generate_batch(prompts[8:16])
                dtb = time.perf_counter() - tb
                out["batched_8_utterances"] = {"prompts_per_s": 8 / dtb, "audio_s_per_s": sum(len(r["pcm"]) for r in res) / 24000.0 / dtb,
                                               "wall_ms": dtb * 1e3, "note": "bark_hip_generate_batch: lock-step decode, per-utterance results bit-identical to the single path"}
                bctx.free()
            except Exception as e:      # noqa: BLE001
                out["batched_8_utterances"] = {"error": str(e)}
        # the reference's default step cap (n_steps_text_encoder = 768, bark.cpp:2212): 1154 frames = 15.4 s per prompt, two fine windows
        if world == 1:
            try:
                ctx.set_params(pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=768))
                ctx.generate_audio(prompts[3])
                t7 = time.perf_counter(); assert ctx.generate_audio(prompts[4]); d7 = time.perf_counter() - t7
                s7 = ctx.stats()
                out["default_cap_768_steps"] = {"rtf": s7["n_samples"] / 24000.0 / d7, "ms_per_prompt": d7 * 1e3, "audio_s": s7["n_samples"] / 24000.0,
                                                "n_frames": s7["n_frames"]}
                ctx.set_params(pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=a.n_semantic))
            except Exception as e:      # noqa: BLE001
                out["default_cap_768_steps"] = {"error": str(e)}
        # BASELINE config 4: the same model quantised to q4_0 by the native bark_model_quantize (reported beside the headline)
        if world == 1 and not a.no_q4:
            try:
                qpath = path[:-4] + "_q4_0.bin"
                if not os.path.exists(qpath):
                    assert pkg.load_library().bark_model_quantize(path.encode(), (qpath + ".tmp").encode(), 2)
                    os.replace(qpath + ".tmp", qpath)
                qctx = pkg.BarkContext.load_model(qpath, pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=a.n_semantic), seed=0)
                qctx.generate_audio(prompts[0])
                tq = time.perf_counter(); qa = 0.0
                for i in range(2):
                    assert qctx.generate_audio(prompts[1 + i]); qa += qctx.stats()["n_samples"] / 24000.0
                dtq = time.perf_counter() - tq
                dus, dbytes = qctx.time_decode_step(0, 640, 300)
                fus, flops = qctx.time_fine_pass(6)
                out["q4_0"] = {"rtf": qa / dtq, "ms_per_prompt": dtq * 500.0, "file_MB": os.path.getsize(qpath) / 1e6,
                               "decode_step_us": dus, "decode_step_GB/s": dbytes / (dus * 1e-6) / 1e9,
                               "fine_pass_us": fus, "fine_pass_equiv_TFLOP/s": flops / (fus * 1e-6) / 1e12,
                               "note": "q4_0 x q8_0 block products: v_dot4 GEMV (decode), v_mfma_i32_32x32x32_i8 (prefill / fine); bit-exact vs the oracle"}
                qctx.free()
            except Exception as e:      # noqa: BLE001
                out["q4_0"] = {"error": str(e)}
        if not a.no_cpu_baseline and world == 1:
            from oracle.pyoracle import Oracle
            cores = min(os.cpu_count() or 4, 4)                 # BASELINE config 1: examples/main -t 4
            orc = Oracle(path, n_threads=cores)
            n_small = 24
            t1 = time.perf_counter()
            ref = orc.generate(prompts[a.warmup % len(prompts)], orc.params(n_steps_text_encoder=n_small))
            cdt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": ref["n_samples"] / 24000.0 / cdt, "unit": "audio-s/s", "cores": cores, "kind": "port",
                                   "sample": f"same prompt, n_steps_text_encoder={n_small} ({ref['n_samples'] / 24000.0:.2f} s audio, {cdt:.1f} s CPU wall)",
                                   "stage_ms_per_token": {
                                       "semantic": ref["t_predict_semantic_us"] / 1000.0 / max(1, ref["n_sample_semantic"]),
                                       "coarse": ref["t_predict_coarse_us"] / 1000.0 / max(1, ref["n_sample_coarse"]),
                                       "fine": ref["t_predict_fine_us"] / 1000.0 / max(1, ref["n_sample_fine"])}}
            orc.close()
        print(json.dumps(out))
    ctx.free()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
