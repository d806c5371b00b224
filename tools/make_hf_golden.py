#!/usr/bin/env python3
"""Pin the CPU oracle's *architecture and layout* against an independent implementation.

The reference's own engine cannot be built here (ggml / encodec.cpp submodule absent), and it
ships no golden vectors.  The closest independent implementation available offline is the
PyTorch model that the reference's convert.py converts FROM: HuggingFace `transformers`
Bark (modeling_bark.py) and EnCodec (modeling_encodec.py).  This script

  1. writes the deterministic synthetic `toy` model file (tools/make_synth_model.py, seed 0),
  2. loads the very same tensors into HF BarkCausalModel / BarkFineModel / EncodecDecoder
     through the inverse of convert.py's name map (convert.py:222-267, 151-167),
  3. runs single forward passes on fixed inputs and stores the outputs as
     tests/golden/hf_toy_s0.npz.

tests/test_oracle_golden.py then checks the oracle (with its ggml-specific rounding switched
off and HF's erf GELU selected) against these vectors.  HF differs from bark.cpp *by design* in
GELU flavour (erf vs tanh-LUT) and in f16 activation rounding, so this pins structure — layer
order, tensor layout, masks, padding, gate order — not ggml's rounding.

Run in the build container only (needs torch + transformers); the GPU box uses the .npz.
"""
from __future__ import annotations

import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.make_synth_model import ensure_model  # noqa: E402


def read_model_file(path: str):
    """Minimal reader of the on-disk layout -> {section: (hparams, {name: ndarray (torch order)})}."""
    buf = np.fromfile(path, dtype=np.uint8)
    pos = 0

    def i32(n=1):
        nonlocal pos
        v = struct.unpack_from("<%di" % n, buf, pos)
        pos += 4 * n
        return v if n > 1 else v[0]

    def tensor():
        nonlocal pos
        n_dims, ln, tt = i32(3)
        dims = [i32() for _ in range(n_dims)]
        name = bytes(buf[pos:pos + ln]).decode()
        pos += ln
        n = int(np.prod(dims)) if dims else 1
        dt = np.float16 if tt == 1 else np.float32
        arr = np.frombuffer(buf, dtype=dt, count=n, offset=pos).reshape(list(reversed(dims)))
        pos += n * arr.itemsize
        return name, arr

    assert i32() == 0x67676D6C
    n_vocab = i32()
    vocab = []
    for _ in range(n_vocab):
        ln = i32()
        vocab.append(bytes(buf[pos:pos + ln]))
        pos += ln
    out = {"vocab": vocab}
    for sec in ("semantic", "coarse", "fine"):
        hp = dict(zip(["n_layer", "n_head", "n_embd", "block_size", "bias", "n_in", "n_out", "n_lm_heads", "n_wtes", "ftype"], i32(10)))
        tens = dict(tensor() for _ in range(i32()))
        out[sec] = (hp, tens)
    assert i32() == 0x67676D6C
    hp = dict(zip(["in_channels", "hidden_dim", "n_filters", "kernel_size", "residual_kernel_size", "n_bins", "bandwidth", "sr", "ftype"], i32(9)))
    tens = {}
    while pos < len(buf):
        k, v = tensor()
        tens[k] = v
    out["codec"] = (hp, tens)
    return out


def build_hf_gpt(hp, tens, fine: bool):
    import torch
    from transformers.models.bark.configuration_bark import BarkFineConfig, BarkSemanticConfig
    from transformers.models.bark.modeling_bark import BarkCausalModel, BarkFineModel

    kw = dict(block_size=hp["block_size"], input_vocab_size=hp["n_in"], output_vocab_size=hp["n_out"],
              num_layers=hp["n_layer"], num_heads=hp["n_head"], hidden_size=hp["n_embd"], dropout=0.0, bias=False)
    if fine:
        # the real checkpoints tie lm_heads[i] to input_embeds_layers[i+1]; the converted file stores both
        # tensors and bark.cpp loads both independently, so the synthetic file keeps them independent
        cfg = BarkFineConfig(n_codes_total=hp["n_wtes"], n_codes_given=hp["n_wtes"] - hp["n_lm_heads"],
                             tie_word_embeddings=False, **kw)
        cfg._attn_implementation = "eager"
        model = BarkFineModel(cfg)
    else:
        cfg = BarkSemanticConfig(**kw)
        cfg._attn_implementation = "eager"
        model = BarkCausalModel(cfg)
    t = lambda a: torch.from_numpy(np.array(a, dtype=np.float32))
    sd = {}
    if fine:
        for i in range(hp["n_wtes"]):
            sd[f"input_embeds_layers.{i}.weight"] = t(tens[f"model/wte/{i}"])
        for i in range(hp["n_lm_heads"]):
            sd[f"lm_heads.{i}.weight"] = t(tens[f"model/lm_head/{i}"])
    else:
        sd["input_embeds_layer.weight"] = t(tens["model/wte/0"])
        sd["lm_head.weight"] = t(tens["model/lm_head/0"])
    sd["position_embeds_layer.weight"] = t(tens["model/wpe"])
    sd["layernorm_final.weight"] = t(tens["model/ln_f/g"])
    if "model/ln_f/b" in tens:
        sd["layernorm_final.bias"] = t(tens["model/ln_f/b"])
    for l in range(hp["n_layer"]):
        p = f"model/h{l}"
        sd[f"layers.{l}.layernorm_1.weight"] = t(tens[p + "/ln_1/g"])
        sd[f"layers.{l}.layernorm_2.weight"] = t(tens[p + "/ln_2/g"])
        if p + "/ln_1/b" in tens:
            sd[f"layers.{l}.layernorm_1.bias"] = t(tens[p + "/ln_1/b"])
            sd[f"layers.{l}.layernorm_2.bias"] = t(tens[p + "/ln_2/b"])
        sd[f"layers.{l}.attn.att_proj.weight"] = t(tens[p + "/attn/c_attn/w"])
        sd[f"layers.{l}.attn.out_proj.weight"] = t(tens[p + "/attn/c_proj/w"])
        sd[f"layers.{l}.mlp.in_proj.weight"] = t(tens[p + "/mlp/c_fc/w"])
        sd[f"layers.{l}.mlp.out_proj.weight"] = t(tens[p + "/mlp/c_proj/w"])
    if fine:   # make sure no parameter is shared before loading independent tensors
        for i in range(hp["n_lm_heads"]):
            assert model.lm_heads[i].weight.data_ptr() != model.input_embeds_layers[i + 1].weight.data_ptr()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if not m.endswith("attn.bias")]
    assert not unexpected, unexpected
    assert not missing, missing
    return model.eval()


def build_hf_codec(hp, tens):
    import torch
    from transformers.models.encodec.configuration_encodec import EncodecConfig
    from transformers.models.encodec.modeling_encodec import EncodecDecoder

    cfg = EncodecConfig(num_filters=hp["n_filters"], hidden_size=hp["hidden_dim"], codebook_size=hp["n_bins"],
                        codebook_dim=hp["hidden_dim"], use_causal_conv=True, norm_type="time_group_norm")
    # norm_type would be weight_norm in the real model; the file stores the already folded weight, so build
    # plain convolutions and strip the GroupNorm that "time_group_norm" adds.
    dec = EncodecDecoder(cfg)
    for mod in dec.modules():
        if hasattr(mod, "norm_type"):
            mod.norm_type = "weight_norm"     # forward() then skips self.norm
    t = lambda a: torch.from_numpy(np.array(a, dtype=np.float32))
    sd = {}
    for name, arr in tens.items():
        if not name.startswith("decoder."):
            continue
        k = name.replace("decoder.model.", "layers.")
        k = k.replace(".convtr.convtr.", ".conv.").replace(".conv.conv.", ".conv.")
        a = t(arr)
        if k.endswith(".bias") and a.ndim == 0:
            a = a.reshape(1)
        sd[k] = a
    own = dec.state_dict()
    for k, v in sd.items():
        assert k in own, k
        assert tuple(own[k].shape) == tuple(v.shape), (k, own[k].shape, v.shape)
    load = {k: sd.get(k, v) for k, v in own.items()}
    assert all(k in sd or ".norm." in k for k in own), [k for k in own if k not in sd and ".norm." not in k]
    dec.load_state_dict(load)
    return dec.eval()


def use_tanh_gelu(model):
    """ggml_gelu is the tanh approximation; HF Bark's MLP uses the exact erf form (modeling_bark.py:264)."""
    import torch
    for layer in model.layers:
        layer.mlp.gelu = torch.nn.GELU(approximate="tanh")
    return model


def greedy_semantic_loop(sem, prompt, n_steps, eos_token=10000, min_eos_p=0.2):
    """The reference's semantic loop (bark.cpp:1645-1701) driven through HF forward passes: sample over ALL logits with
    gpt_argmax_sample's rule (l / 0.7, softmax, first maximum; eos_p = p[last]), stop on the eos id or eos_p >= min_eos_p.
    Returns ids, per-step top-2 margins (in l / 0.7 units) and eos probabilities."""
    import torch
    emb = sem.input_embeds_layer(torch.from_numpy(prompt)[None])
    merged = torch.cat([emb[:, :256] + emb[:, 256:512], emb[:, 512:]], dim=1)
    r = sem(inputs_embeds=merged, use_cache=True)
    pkv = r.past_key_values
    ids, margins, eos = [], [], []
    for _ in range(n_steps):
        l = (r.logits[0, -1].numpy().astype(np.float32) / np.float32(0.7)).astype(np.float32)
        e = np.exp((l - l.max()).astype(np.float64)).astype(np.float32)
        p = e / np.float32(e.sum(dtype=np.float32))
        nxt = int(np.argmax(p))
        srt = np.sort(l)[::-1]
        margins.append(float(srt[0] - srt[1])); eos.append(float(p[-1]))
        if nxt == eos_token or p[-1] >= min_eos_p:
            break
        ids.append(nxt)
        r = sem(input_ids=torch.tensor([[nxt]]), past_key_values=pkv, use_cache=True)
        pkv = r.past_key_values
    return np.array(ids, np.int32), np.array(margins, np.float32), np.array(eos, np.float32)


def main_tanh(preset: str, n_greedy: int):
    """tests/golden/hf_<preset>_tanh_s0.npz: HF forward passes with the tanh GELU (= ggml_gelu without its f16 table) on the synthetic
    `preset` weights, plus an n_greedy-step greedy semantic loop.  The oracle is compared with gelu_mode=1, act_round_f16=0 (2e-4) and
    in its default mode (f16 rounding noise, bound stated in tests/test_oracle_golden.py)."""
    import torch
    torch.set_num_threads(8)
    mf = read_model_file(ensure_model(preset, 0))
    rng = np.random.default_rng(4321)
    out = {}
    with torch.no_grad():
        hp, tens = mf["semantic"]
        sem = use_tanh_gelu(build_hf_gpt(hp, tens, fine=False))
        prompt = np.concatenate([rng.integers(10048, 129595, 40), np.full(216, 129595), np.full(256, 10000), [129599]]).astype(np.int64)
        emb = sem.input_embeds_layer(torch.from_numpy(prompt)[None])
        merged = torch.cat([emb[:, :256] + emb[:, 256:512], emb[:, 512:]], dim=1)
        r = sem(inputs_embeds=merged, use_cache=True)
        out["sem_prompt"] = prompt.astype(np.int32)
        out["sem_logits0"] = r.logits[0, -1].numpy()
        r = sem(input_ids=torch.tensor([[4242]]), past_key_values=r.past_key_values, use_cache=True)
        out["sem_logits1"] = r.logits[0, -1].numpy()
        ids, margins, eos = greedy_semantic_loop(sem, prompt, n_greedy)
        out["greedy_ids"] = ids; out["greedy_margins"] = margins; out["greedy_eos_p"] = eos
        del sem
        hp, tens = mf["coarse"]
        co = use_tanh_gelu(build_hf_gpt(hp, tens, fine=False))
        cprompt = np.concatenate([rng.integers(0, 10000, 256), [12050], rng.integers(10000, 12048, 43)]).astype(np.int64)
        r = co(input_ids=torch.from_numpy(cprompt)[None], use_cache=True)
        out["coarse_prompt"] = cprompt.astype(np.int32)
        out["coarse_logits0"] = r.logits[0, -1].numpy()
        r = co(input_ids=torch.tensor([[10777]]), past_key_values=r.past_key_values, use_cache=True)
        out["coarse_logits1"] = r.logits[0, -1].numpy()
        del co
        hp, tens = mf["fine"]
        fi = use_tanh_gelu(build_hf_gpt(hp, tens, fine=True))
        ftok = rng.integers(0, 1024, (8, 1024)).astype(np.int64)
        ftok[:, 900:] = 1024
        out["fine_tokens"] = ftok.astype(np.int32)
        rows = np.array([0, 1, 7, 100, 511, 512, 899, 900, 1023])
        out["fine_rows"] = rows.astype(np.int32)
        r = fi(codebook_idx=3, input_ids=torch.from_numpy(ftok.T.copy())[None])
        out["fine_logits_nn3"] = r.logits[0, rows].numpy()
    dst = os.path.join(ROOT, "tests", "golden", f"hf_{preset}_tanh_s0.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", len(out["greedy_ids"]), "greedy ids, smallest top-2 margin", float(out["greedy_margins"].min()))


def main_codec(preset: str):
    """tests/golden/hf_<preset>_codec_s0.npz: HF EncodecDecoder (RVQ de-embedding done here, SEANet decoder by HF) on the synthetic
    `preset` codec weights - for `small` / `large` those are EnCodec-24kHz's real dimensions (32 filters, 2 x LSTM 512, ratios 8-5-4-2)."""
    import torch
    torch.set_num_threads(8)
    mf = read_model_file(ensure_model(preset, 0))
    rng = np.random.default_rng(777)
    out = {}
    with torch.no_grad():
        hp, tens = mf["codec"]
        dec = build_hf_codec(hp, tens)
        for T in (5, 40):
            codes = rng.integers(0, 1024, (8, T)).astype(np.int64)
            z = sum(torch.from_numpy(np.array(tens[f"quantizer.vq.layers.{q}._codebook.embed"]))[codes[q]] for q in range(8))
            pcm = dec(z.T[None])[0, 0].numpy()
            out[f"codec_codes_T{T}"] = codes.astype(np.int32)
            out[f"codec_pcm_T{T}"] = pcm
    dst = os.path.join(ROOT, "tests", "golden", f"hf_{preset}_codec_s0.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes", {k: v.shape for k, v in out.items()})


def main_stages(preset: str, n_sem: int):
    """tests/golden/hf_<preset>_stages_s0.npz: HF's own stage loops - BarkCoarseModel.generate (sliding windows of 60 tokens over 630
    tokens of history, alternating codebooks) and BarkFineModel.generate (windows of 1024 frames hopping by 512) - greedy, tanh GELU, on
    the synthetic `preset` weights, from n_sem random semantic ids.  An implementation of bark.cpp:1745-1863 / :1916-2059 that shares
    no code with the oracle.  One adjustment: HF's AlternatingCodebooksLogitsProcessor leaves ids ABOVE the second codebook unmasked
    (trained weights never produce them, random weights do); bark.cpp takes exactly one codebook's window of logits
    (bark.cpp:1829-1833), so the fixture masks them too."""
    import torch
    import transformers.models.bark.modeling_bark as mb
    from transformers.models.bark.configuration_bark import BarkCoarseConfig
    from transformers.models.bark.generation_configuration_bark import (BarkCoarseGenerationConfig, BarkFineGenerationConfig,
                                                                         BarkSemanticGenerationConfig)
    orig_call = mb.AlternatingCodebooksLogitsProcessor.__call__

    def call(self, input_ids, scores):
        out = orig_call(self, input_ids, scores)
        out[:, self.semantic_vocab_size + 2 * self.codebook_size:] = -float("inf")
        return out
    mb.AlternatingCodebooksLogitsProcessor.__call__ = call
    torch.set_num_threads(8)
    mf = read_model_file(ensure_model(preset, 0))
    rng = np.random.default_rng(99)
    semantic = rng.integers(0, 10000, n_sem).astype(np.int64)
    with torch.no_grad():
        # ---- semantic stage through HF's own generate: raw word-piece ids in, HF adds the offset, pads, merges the history row by row.
        #      Two adjustments: (1) the reference hands gpt_sample the FULL logit vector (bark.cpp:1682 passes `logits`, the
        #      `relevant_logits` built two lines above are never used), so HF's suppression of the ids 10001 .. 10047 is switched off;
        #      (2) min_eos_p off (HF evaluates it at temperature 1, the reference at 0.7 - not comparable under greedy decoding), so
        #      the only stop rule is argmax == eos
        class _NoSuppress:
            def __init__(self, *a, **k): pass
            def __call__(self, input_ids, scores): return scores
        mb.SuppressTokensLogitsProcessor = _NoSuppress
        hps, tenss = mf["semantic"]
        sbase = build_hf_gpt(hps, tenss, fine=False)
        scfg = mb.BarkSemanticConfig(block_size=hps["block_size"], input_vocab_size=hps["n_in"], output_vocab_size=hps["n_out"], num_layers=hps["n_layer"],
                                     num_heads=hps["n_head"], hidden_size=hps["n_embd"], dropout=0.0, bias=False)
        scfg._attn_implementation = "eager"
        sm = mb.BarkSemanticModel(scfg)
        sm.load_state_dict(sbase.state_dict())
        sm = use_tanh_gelu(sm).eval()
        n_text = 37
        text_ids = rng.integers(0, 100000, n_text).astype(np.int64)
        ids256 = np.zeros(256, np.int64); ids256[:n_text] = text_ids
        mask = np.zeros(256, np.int64); mask[:n_text] = 1
        n_sem_steps = 48
        sgen = BarkSemanticGenerationConfig(do_sample=False, temperature=1.0, min_eos_p=None, max_new_tokens=n_sem_steps)
        assert (sgen.text_encoding_offset, sgen.text_pad_token, sgen.semantic_pad_token, sgen.semantic_infer_token, sgen.max_input_semantic_length,
                sgen.eos_token_id) == (10048, 129595, 10000, 129599, 256, 10000)
        sout = sm.generate(torch.from_numpy(ids256)[None], semantic_generation_config=sgen, attention_mask=torch.from_numpy(mask)[None])
        sem_ids = sout[0].numpy()
        if (sem_ids == 10000).any():
            sem_ids = sem_ids[: int(np.argmax(sem_ids == 10000))]
        del sm, sbase
        hp, tens = mf["coarse"]
        base = build_hf_gpt(hp, tens, fine=False)
        cfg = BarkCoarseConfig(block_size=hp["block_size"], input_vocab_size=hp["n_in"], output_vocab_size=hp["n_out"], num_layers=hp["n_layer"],
                               num_heads=hp["n_head"], hidden_size=hp["n_embd"], dropout=0.0, bias=False)
        cfg._attn_implementation = "eager"
        co = mb.BarkCoarseModel(cfg)
        co.load_state_dict(base.state_dict())
        co = use_tanh_gelu(co).eval()
        sg = BarkSemanticGenerationConfig()
        cg = BarkCoarseGenerationConfig(do_sample=False, temperature=1.0)
        assert (sg.semantic_vocab_size, sg.semantic_rate_hz, cg.max_coarse_input_length, cg.max_coarse_history, cg.sliding_window_len,
                cg.coarse_semantic_pad_token, cg.coarse_infer_token, cg.coarse_rate_hz) == (10000, 49.9, 256, 630, 60, 12048, 12050, 75)
        out = co.generate(torch.from_numpy(semantic)[None].clone(), semantic_generation_config=sg, coarse_generation_config=cg, codebook_size=1024)
        flat = out[0].numpy()
        coarse = np.stack([flat[0::2] - 10000, flat[1::2] - 10000 - 1024], axis=1)
        assert coarse.min() >= 0 and coarse.max() < 1024
        hpf, tensf = mf["fine"]
        fi = use_tanh_gelu(build_hf_gpt(hpf, tensf, fine=True))
        fout = fi.generate(out.clone(), semantic_generation_config=sg, coarse_generation_config=cg,
                           fine_generation_config=BarkFineGenerationConfig(temperature=None), codebook_size=1024)
        fine = fout[0].numpy().T
    dst = os.path.join(ROOT, "tests", "golden", f"hf_{preset}_stages_s0.npz")
    np.savez_compressed(dst, semantic=semantic.astype(np.int32), coarse=coarse.astype(np.int32), fine=fine.astype(np.int32),
                        text_ids=text_ids.astype(np.int32), semantic_from_text=sem_ids.astype(np.int32), n_semantic_steps=np.int32(n_sem_steps))
    print("wrote", dst, os.path.getsize(dst), "bytes:", n_sem, "semantic ids ->", coarse.shape, "coarse,", fine.shape, "fine")


def main():
    import torch

    if len(sys.argv) > 1 and sys.argv[1] == "stages":
        main_stages(sys.argv[2] if len(sys.argv) > 2 else "small", int(sys.argv[3]) if len(sys.argv) > 3 else 256)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "codec":
        main_codec(sys.argv[2] if len(sys.argv) > 2 else "small")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "tanh":
        main_tanh(sys.argv[2] if len(sys.argv) > 2 else "small", int(sys.argv[3]) if len(sys.argv) > 3 else 64)
        return
    torch.set_num_threads(4)
    path = ensure_model("toy", 0)
    mf = read_model_file(path)
    rng = np.random.default_rng(1234)
    out = {}

    with torch.no_grad():
        # ---- semantic: merged 513-token prompt (257 rows) then two decode steps ------------------
        hp, tens = mf["semantic"]
        sem = build_hf_gpt(hp, tens, fine=False)
        prompt = np.concatenate([rng.integers(10048, 129595, 40), np.full(216, 129595), np.full(256, 10000), [129599]]).astype(np.int64)
        emb = sem.input_embeds_layer(torch.from_numpy(prompt)[None])
        merged = torch.cat([emb[:, :256] + emb[:, 256:512], emb[:, 512:]], dim=1)
        r = sem(inputs_embeds=merged, use_cache=True)
        out["sem_prompt"] = prompt.astype(np.int32)
        out["sem_logits0"] = r.logits[0, -1].numpy()
        nxt = [int(out["sem_logits0"].argmax()), 4242]
        pkv = r.past_key_values
        for i, tok in enumerate(nxt):
            r = sem(input_ids=torch.tensor([[tok]]), past_key_values=pkv, use_cache=True)
            pkv = r.past_key_values
            out[f"sem_logits{i + 1}"] = r.logits[0, -1].numpy()
        out["sem_next"] = np.array(nxt, np.int32)

        # ---- coarse: 300-token prefill then one decode step -------------------------------------
        hp, tens = mf["coarse"]
        co = build_hf_gpt(hp, tens, fine=False)
        cprompt = np.concatenate([rng.integers(0, 10000, 256), [12050], rng.integers(10000, 12048, 43)]).astype(np.int64)
        r = co(input_ids=torch.from_numpy(cprompt)[None], use_cache=True)
        out["coarse_prompt"] = cprompt.astype(np.int32)
        out["coarse_logits0"] = r.logits[0, -1].numpy()
        r = co(input_ids=torch.tensor([[10777]]), past_key_values=r.past_key_values, use_cache=True)
        out["coarse_logits1"] = r.logits[0, -1].numpy()

        # ---- fine: N = 1024, codebooks 2 and 7 ------------------------------------------------------
        hp, tens = mf["fine"]
        fi = build_hf_gpt(hp, tens, fine=True)
        ftok = rng.integers(0, 1024, (8, 1024)).astype(np.int64)
        ftok[:, 900:] = 1024                       # padding rows, as bark.cpp:1990-1996 produces
        out["fine_tokens"] = ftok.astype(np.int32)
        rows = np.array([0, 1, 7, 100, 511, 512, 899, 900, 1023])
        out["fine_rows"] = rows.astype(np.int32)
        for nn in (2, 7):
            r = fi(codebook_idx=nn, input_ids=torch.from_numpy(ftok.T.copy())[None])
            out[f"fine_logits_nn{nn}"] = r.logits[0, rows].numpy()

        # ---- codec: RVQ de-embedding + SEANet decoder -------------------------------------------------
        hp, tens = mf["codec"]
        dec = build_hf_codec(hp, tens)
        for T in (3, 50):                                   # T=3 exercises the short-input reflect-pad rule
            codes = rng.integers(0, 1024, (8, T)).astype(np.int64)
            z = sum(torch.from_numpy(np.array(tens[f"quantizer.vq.layers.{q}._codebook.embed"]))[codes[q]] for q in range(8))
            pcm = dec(z.T[None])[0, 0].numpy()
            out[f"codec_codes_T{T}"] = codes.astype(np.int32)
            out[f"codec_pcm_T{T}"] = pcm

    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    dst = os.path.join(ROOT, "tests", "golden", "hf_toy_s0.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")
    for k, v in out.items():
        print(f"  {k:20s} {v.shape} {v.dtype}")


if __name__ == "__main__":
    main()
