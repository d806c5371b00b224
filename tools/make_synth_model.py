#!/usr/bin/env python3
"""Synthetic `ggml_weights.bin` writer.

No Bark checkpoint, BERT vocab or network exists in this environment, so every
test and benchmark runs on random-initialised weights of the real architecture,
written in the exact on-disk layout that the reference's converter emits and its
loader parses:

  * container / vocab  : /root/reference/convert.py:310-322,342 ; bark.cpp:664-690,1095-1102
  * GPT hparams        : convert.py:82-110 ; bark.cpp:700-709
  * GPT tensor records : convert.py:202-290 ; bark.cpp:995-1068
  * codec hparams      : convert.py:59-79
  * codec tensors      : convert.py:113-199 (names after the rewrite rules :151-167,
                         weight-norm already folded, `.squeeze()` quirk on biases)

f16 rounding happens here, once, so the CPU oracle and the HIP engine read
identical bits.  Everything is derived from `numpy.random.default_rng(seed)`.

Usage: python tools/make_synth_model.py --preset small --out /tmp/bark_small.bin
"""
from __future__ import annotations

import argparse
import mmap
import os
import struct
import sys
from dataclasses import dataclass

import numpy as np

MAGIC = 0x67676D6C

# Real Bark token-id constants force these vocab sizes (bark.cpp:2215-2223):
#   text ids live at 10048 + wordpiece_id, pad = 129595, infer = 129599.
SEM_IN_VOCAB, SEM_OUT_VOCAB = 129_600, 10_048
COARSE_VOCAB = 12_096
FINE_VOCAB = 1_056
N_WORDPIECE = 129_595 - 10_048  # 119 547 entries, so that offset + n_vocab == text_pad_token


@dataclass
class Preset:
    n_embd: int
    n_layer: int
    n_head: int
    block_size: int = 1024
    codec_filters: int = 32      # EnCodec num_filters (24 kHz model: 32)
    codec_hidden: int = 128      # RVQ / latent dimension
    codec_n_q: int = 32          # quantizers stored in the file (24 kbps); 8 are used at 6 kbps
    with_encoder: bool = True    # the real file also carries the (unused here) SEANet encoder


PRESETS = {
    # head_dim is 64 in every real Bark model; the toy presets keep that.
    "toy": Preset(n_embd=128, n_layer=2, n_head=2, codec_filters=8, codec_n_q=8, with_encoder=False),
    "mini": Preset(n_embd=256, n_layer=4, n_head=4, codec_filters=16, codec_n_q=8, with_encoder=False),
    "small": Preset(n_embd=768, n_layer=12, n_head=12),
    "large": Preset(n_embd=1024, n_layer=24, n_head=16),
}

_WORDS = """the of and to in is that it was for on are as with his they be at one have this from or had by
hot but some what there we can out other were all your when up use word how said an each she which do their
time if will way about many then them would write like so these her long make thing see him two has look
more day could go come did my sound no most number who over know water than call first people may down side
been now find any new work part take get place made live where after back little only round man year came
show every good me give our under name very through just form much great think say help low line before
turn cause same mean differ move right boy old too does tell sentence set three want air well also play
small end put home read hand port large spell add even land here must big high such follow act why ask men
change went light kind off need house picture try us again animal point mother world near build self earth
father head stand own page should country found answer school grow study still learn plant cover food sun
four thought let keep eye never last door between city tree cross since hard start might story saw far sea
draw left late run while press close night real life few stop open seem together next white children begin
got walk example ease paper often always music those both mark book letter until mile river car feet care
second group carry took rain eat room friend began idea fish mountain north once base hear horse cut sure
watch color face wood main enough plain girl usual young ready above ever red list though feel talk bird
soon body dog family direct pose leave song measure state product black short numeral class wind question
happen complete ship area half rock order fire south problem piece told knew pass farm top whole king size
heard best hour better true during hundred am remember step early hold west ground interest reach fast five
sing listen six table travel less morning ten simple several vowel toward war lay against pattern slow
center love person money serve appear road map science rule govern pull cold notice voice fall power town
fine certain fly unit lead cry dark machine note wait plan figure star box noun field rest correct able
pound done beauty drive stood contain front teach week final gave green oh quick develop sleep warm free
minute strong special mind behind clear tail produce fact street inch lot nothing course stay wheel full
force blue object decide surface deep moon island foot yet busy test record boat common gold possible plane
age dry wonder laugh thousand ago ran check game shape yes hot miss brought heat snow bed bring sit perhaps
fill east weight language among audio generated bark speech model hello world""".split()


def synth_vocab() -> list[bytes]:
    """119 547 WordPiece-style entries: specials, ASCII singles, '##' continuations, words, filler.

    Every printable ASCII character exists both bare and as a '##' continuation,
    so no ASCII prompt can reach the tokenizer's unknown-character path
    (bark.cpp:611-615) unless a test wants it to (non-ASCII bytes still do).
    """
    toks: list[str] = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    singles = [chr(c) for c in range(33, 127)]
    toks += singles
    toks += ["##" + s for s in singles]
    seen = set(toks)
    for w in _WORDS:
        for cand in (w, w.capitalize(), "##" + w):
            if cand not in seen:
                seen.add(cand)
                toks.append(cand)
    i = 0
    while len(toks) < N_WORDPIECE:
        cand = f"zq{i:06d}x"
        i += 1
        if cand not in seen:
            toks.append(cand)
    assert len(toks) == N_WORDPIECE and len(set(toks)) == N_WORDPIECE
    return [t.encode("utf-8") for t in toks]


_CHUNK = 1 << 22  # elements generated per step


def _populated(nbytes: int) -> np.ndarray:
    """Anonymous memory that is faulted in up front (MAP_POPULATE).  First-touch page faults
    cost ~0.2 ms each in the sandboxed VMs this runs in, so fresh numpy buffers of model size
    would take minutes; two small reusable scratch buffers take milliseconds."""
    flags = mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | getattr(mmap, "MAP_POPULATE", 0)
    m = mmap.mmap(-1, nbytes, flags=flags)
    return np.frombuffer(m, dtype=np.uint8)


class _DirectSink:
    """File sink that bypasses the page cache (O_DIRECT) through one aligned, pre-faulted staging
    buffer.  On a cold VM every new page-cache page costs a slow first-touch fault, which made a
    buffered 0.8 GB write take a minute; direct IO writes it at disk speed.  Falls back to
    buffered IO where O_DIRECT is unsupported (tmpfs)."""

    BLK = 4096
    CAP = 16 << 20

    def __init__(self, path: str):
        self.path = path
        self.buf = _populated(self.CAP)
        self.fill = 0
        try:
            self.fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC | os.O_DIRECT, 0o644)
            os.write(self.fd, memoryview(self.buf)[: self.BLK])  # probe: tmpfs rejects direct writes here
            os.lseek(self.fd, 0, os.SEEK_SET)
            self.direct = True
        except OSError:
            self.fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            self.direct = False
        self.total = 0

    def write(self, data):
        mv = memoryview(data).cast("B")
        n, off = len(mv), 0
        while off < n:
            k = min(n - off, self.CAP - self.fill)
            self.buf[self.fill:self.fill + k] = np.frombuffer(mv[off:off + k], dtype=np.uint8)
            self.fill += k
            off += k
            if self.fill == self.CAP:
                self._flush(self.CAP)
        self.total += n

    def _flush(self, nbytes: int):
        view = memoryview(self.buf)[:nbytes]
        done = 0
        while done < nbytes:
            done += os.write(self.fd, view[done:])
        rest = self.fill - nbytes
        if rest:
            self.buf[:rest] = self.buf[nbytes:self.fill].copy()
        self.fill = rest

    def close(self):
        if self.direct:
            whole = (self.fill // self.BLK) * self.BLK
            if whole:
                self._flush(whole)
            if self.fill:  # unaligned tail: pad the last block, then cut the file to its true size
                self.buf[self.fill:self.BLK] = 0
                os.write(self.fd, memoryview(self.buf)[: self.BLK])
            os.close(self.fd)
            os.truncate(self.path, self.total)
        else:
            if self.fill:
                self._flush(self.fill)
            os.close(self.fd)


class _Writer:
    def __init__(self, path: str):
        self.f = _DirectSink(path)
        self._f32 = _populated(_CHUNK * 4).view(np.float32)
        self._f16 = _populated(_CHUNK * 2).view(np.float16)

    def i32(self, *v: int):
        self.f.write(struct.pack("<%di" % len(v), *v))

    def u32(self, v: int):
        self.f.write(struct.pack("<I", v))

    def header(self, name: str, shape, f16: bool):
        """Record header: n_dims, name_len, ttype, dims (reversed torch order), name."""
        nb = name.encode("utf-8")
        self.i32(len(shape), len(nb), 1 if f16 else 0)
        for d in reversed(shape):
            self.i32(int(d))
        self.f.write(nb)

    def tensor(self, name: str, arr: np.ndarray):
        """Small, already materialised tensor."""
        assert arr.dtype in (np.float32, np.float16)
        self.header(name, arr.shape, arr.dtype == np.float16)
        self.f.write(np.ascontiguousarray(arr).tobytes())

    def random(self, name: str, rng, shape, dist: str, scale: float, f16: bool, offset: float = 0.0):
        """Stream a random tensor to disk chunk by chunk: offset + scale * {N(0,1) | U(-1,1)}."""
        self.header(name, shape, f16)
        n = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
        done = 0
        while done < n:
            m = min(_CHUNK, n - done)
            buf = self._f32[:m]
            if dist == "normal":
                rng.standard_normal(size=m, dtype=np.float32, out=buf)
            else:
                rng.random(size=m, dtype=np.float32, out=buf)
                buf *= np.float32(2.0)
                buf -= np.float32(1.0)
            buf *= np.float32(scale)
            if offset:
                buf += np.float32(offset)
            if f16:
                h = self._f16[:m]
                np.copyto(h, buf, casting="same_kind")
                self.f.write(memoryview(h))
            else:
                self.f.write(memoryview(buf))
            done += m

    def close(self):
        self.f.close()


def _write_gpt(w: _Writer, rng, p: Preset, n_in: int, n_out: int, n_wtes: int, n_lm_heads: int,
               ln_bias: bool, use_f16: bool, logit_gain: float):
    """One GPT section.  `ln_bias` mirrors convert.py:91-94: the fine model sets the
    `bias` hparam although only its LayerNorms carry a bias."""
    E, L = p.n_embd, p.n_layer
    w.i32(L, p.n_head, E, p.block_size, int(ln_bias), n_in, n_out, n_lm_heads, n_wtes, int(use_f16))
    per_layer = 6 + (2 if ln_bias else 0)
    n_tensors = n_wtes + 1 + L * per_layer + 1 + (1 if ln_bias else 0) + n_lm_heads
    w.i32(n_tensors)
    proj_std = 0.02 / np.sqrt(2.0 * L)
    f16 = use_f16
    for i in range(n_wtes):
        w.random(f"model/wte/{i}", rng, (n_in, E), "normal", 0.02, f16)
    w.random("model/wpe", rng, (p.block_size, E), "normal", 0.02, False)
    for l in range(L):
        pre = f"model/h{l}"
        w.random(pre + "/ln_1/g", rng, (E,), "normal", 0.05, False, offset=1.0)
        if ln_bias:
            w.random(pre + "/ln_1/b", rng, (E,), "normal", 0.02, False)
        w.random(pre + "/attn/c_attn/w", rng, (3 * E, E), "normal", 0.02, f16)
        w.random(pre + "/attn/c_proj/w", rng, (E, E), "normal", proj_std, f16)
        w.random(pre + "/ln_2/g", rng, (E,), "normal", 0.05, False, offset=1.0)
        if ln_bias:
            w.random(pre + "/ln_2/b", rng, (E,), "normal", 0.02, False)
        w.random(pre + "/mlp/c_fc/w", rng, (4 * E, E), "normal", 0.02, f16)
        w.random(pre + "/mlp/c_proj/w", rng, (E, 4 * E), "normal", proj_std, f16)
    w.random("model/ln_f/g", rng, (E,), "normal", 0.05, False, offset=1.0)
    if ln_bias:
        w.random("model/ln_f/b", rng, (E,), "normal", 0.02, False)
    for i in range(n_lm_heads):
        w.random(f"model/lm_head/{i}", rng, (n_out, E), "normal", 0.02 * logit_gain, f16)


def _codec_plan(p: Preset):
    """(name, shape, kind) for the SEANet decoder + RVQ, HF layer indices
    (modeling_encodec.py:316-347) renamed as convert.py:151-167 does."""
    F, H = p.codec_filters, p.codec_hidden
    D = 16 * F  # channels after the first conv == LSTM width
    plan = []
    plan.append(("decoder.model.0.conv.conv", (D, H, 7), "conv"))
    plan.append(("decoder.model.1.lstm", D, "lstm"))
    ch = D
    idx = 3
    for ratio in (8, 5, 4, 2):
        plan.append((f"decoder.model.{idx}.convtr.convtr", (ch, ch // 2, 2 * ratio), "convtr"))
        c = ch // 2
        plan.append((f"decoder.model.{idx + 1}.block.1.conv.conv", (c // 2, c, 3), "conv"))
        plan.append((f"decoder.model.{idx + 1}.block.3.conv.conv", (c, c // 2, 1), "conv"))
        plan.append((f"decoder.model.{idx + 1}.shortcut.conv.conv", (c, c, 1), "conv"))
        ch = c
        idx += 3
    plan.append((f"decoder.model.{idx}.conv.conv", (1, F, 7), "conv"))
    return plan


def _encoder_plan(p: Preset):
    F, H = p.codec_filters, p.codec_hidden
    plan = [("encoder.model.0.conv.conv", (F, 1, 7), "conv")]
    ch, idx = F, 1
    for ratio in (2, 4, 5, 8):
        plan.append((f"encoder.model.{idx}.block.1.conv.conv", (ch // 2, ch, 3), "conv"))
        plan.append((f"encoder.model.{idx}.block.3.conv.conv", (ch, ch // 2, 1), "conv"))
        plan.append((f"encoder.model.{idx}.shortcut.conv.conv", (ch, ch, 1), "conv"))
        plan.append((f"encoder.model.{idx + 2}.conv.conv", (2 * ch, ch, 2 * ratio), "conv"))
        ch *= 2
        idx += 3
    plan.append((f"encoder.model.{idx}.lstm", ch, "lstm"))
    plan.append((f"encoder.model.{idx + 2}.conv.conv", (H, ch, 7), "conv"))
    return plan


def _write_codec(w: _Writer, rng, p: Preset, use_f16: bool):
    w.u32(MAGIC)  # convert.py:303
    # in_channels, hidden_dim, n_filters, kernel, residual_kernel, n_bins, bandwidth, sr, ftype  (convert.py:59-79)
    w.i32(1, p.codec_hidden, p.codec_filters, 7, 3, 1024, 24, 24000, int(use_f16))
    f16 = use_f16

    def emit(plan):
        for name, shape, kind in plan:
            if kind == "lstm":
                D = shape
                k = 1.0 / np.sqrt(D)
                for layer in range(2):
                    w.random(f"{name}.weight_ih_l{layer}", rng, (4 * D, D), "uniform", k, f16)
                    w.random(f"{name}.weight_hh_l{layer}", rng, (4 * D, D), "uniform", k, f16)
                    w.random(f"{name}.bias_ih_l{layer}", rng, (4 * D,), "uniform", k, False)
                    w.random(f"{name}.bias_hh_l{layer}", rng, (4 * D,), "uniform", k, False)
            else:
                if kind == "conv":
                    fan_in, n_bias = shape[1] * shape[2], shape[0]
                else:  # transposed conv weight is [in, out, k]; each output sees in*k/stride taps
                    fan_in, n_bias = shape[0] * 2, shape[1]
                # `.squeeze()` quirk (convert.py:134-136): a 1-element bias is written 0-d
                bshape = (n_bias,) if n_bias > 1 else ()
                w.random(name + ".bias", rng, bshape, "normal", 0.02, False)
                w.random(name + ".weight", rng, shape, "normal", 1.0 / np.sqrt(fan_in), f16)

    if p.with_encoder:
        emit(_encoder_plan(p))
    emit(_codec_plan(p))
    for q in range(p.codec_n_q):
        w.random(f"quantizer.vq.layers.{q}._codebook.embed", rng, (1024, p.codec_hidden), "normal", 1.0, False)


def write_model(path: str, preset: str = "small", seed: int = 0, use_f16: bool = True,
                logit_gain: float = 1.0) -> str:
    """Write a complete synthetic model file; returns `path`."""
    p = PRESETS[preset]
    rng = np.random.default_rng(seed)
    tmp = path + ".tmp%d" % os.getpid()
    w = _Writer(tmp)
    w.u32(MAGIC)
    vocab = synth_vocab()
    w.i32(len(vocab))
    for t in vocab:
        w.i32(len(t))
        w.f.write(t)
    _write_gpt(w, rng, p, SEM_IN_VOCAB, SEM_OUT_VOCAB, 1, 1, False, use_f16, logit_gain)
    _write_gpt(w, rng, p, COARSE_VOCAB, COARSE_VOCAB, 1, 1, False, use_f16, logit_gain)
    _write_gpt(w, rng, p, FINE_VOCAB, FINE_VOCAB, 8, 7, True, use_f16, logit_gain)
    _write_codec(w, rng, p, use_f16)
    w.close()
    os.replace(tmp, path)
    return path


def ensure_model(preset: str = "small", seed: int = 0, cache_dir: str | None = None) -> str:
    """Create the file once per (preset, seed) under `cache_dir` (default: $BARK_SYNTH_DIR or /tmp)."""
    cache_dir = cache_dir or os.environ.get("BARK_SYNTH_DIR", "/tmp/bark_synth")
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, f"bark_{preset}_s{seed}.bin")
    if not os.path.exists(path):
        write_model(path, preset, seed)
    return path


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="small", choices=sorted(PRESETS))
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", required=True)
    ap.add_argument("--f32", action="store_true")
    a = ap.parse_args()
    write_model(a.out, a.preset, a.seed, use_f16=not a.f32)
    print(a.out, os.path.getsize(a.out), "bytes", file=sys.stderr)
