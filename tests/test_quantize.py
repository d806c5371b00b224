"""bark_model_quantize (bark.h:229-232): the native q4_0 / q4_1 / q5_0 / q5_1 / q8_0 writers against independent numpy
restatements of ggml's reference block quantisers, and the container rules of /root/reference/bark.cpp:272-478,2234-2377."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

from tools.make_hf_golden import read_model_file


F32 = np.float32


def _inv(d):
    return np.where(d != 0, F32(1.0) / np.where(d != 0, d, 1).astype(F32), F32(0.0)).astype(F32)


def _signed_max(xb):
    idx = np.argmax(np.abs(xb), axis=1)                       # first element of largest magnitude, sign kept
    return xb[np.arange(len(xb)), idx]


def _pack5(q):
    """32 five-bit values per row -> (16 nibble bytes, u32 of fifth bits)."""
    qs = ((q[:, :16] & 0x0F) | ((q[:, 16:] & 0x0F) << 4)).astype(np.uint8)
    bits = ((q >> 4) & 1).astype(np.uint32)
    qh = (bits << np.arange(32, dtype=np.uint32)[None, :]).sum(axis=1).astype(np.uint32)
    return qs, qh


def q4_0_ref(x):
    """quantize_row_q4_0_ref: 32-element blocks -> (f16 d, 16 nibble bytes)."""
    xb = np.ascontiguousarray(x, dtype=F32).reshape(-1, 32)
    d = (_signed_max(xb) / F32(-8.0)).astype(F32)
    q = np.minimum(15, (xb * _inv(d)[:, None] + F32(8.5)).astype(F32).astype(np.int8).astype(np.int32)).astype(np.uint8)
    qs = (q[:, :16] | (q[:, 16:] << 4)).astype(np.uint8)
    return d.astype(np.float16), qs


def q4_1_ref(x):
    xb = np.ascontiguousarray(x, dtype=F32).reshape(-1, 32)
    mn, mx = xb.min(axis=1), xb.max(axis=1)
    d = ((mx - mn).astype(F32) / F32(15.0)).astype(F32)
    q = np.minimum(15, ((xb - mn[:, None]).astype(F32) * _inv(d)[:, None] + F32(0.5)).astype(F32).astype(np.int8).astype(np.int32)).astype(np.uint8)
    return d.astype(np.float16), mn.astype(np.float16), (q[:, :16] | (q[:, 16:] << 4)).astype(np.uint8)


def q5_0_ref(x):
    xb = np.ascontiguousarray(x, dtype=F32).reshape(-1, 32)
    d = (_signed_max(xb) / F32(-16.0)).astype(F32)
    q = np.minimum(31, (xb * _inv(d)[:, None] + F32(16.5)).astype(F32).astype(np.int8).astype(np.int32)).astype(np.uint8)
    qs, qh = _pack5(q)
    return d.astype(np.float16), qh, qs


def q5_1_ref(x):
    xb = np.ascontiguousarray(x, dtype=F32).reshape(-1, 32)
    mn, mx = xb.min(axis=1), xb.max(axis=1)
    d = ((mx - mn).astype(F32) / F32(31.0)).astype(F32)
    q = ((xb - mn[:, None]).astype(F32) * _inv(d)[:, None] + F32(0.5)).astype(F32).astype(np.int32).astype(np.uint8)     # no clamp in ggml
    qs, qh = _pack5(q)
    return d.astype(np.float16), mn.astype(np.float16), qh, qs


def q8_0_ref(x):
    xb = np.ascontiguousarray(x, dtype=F32).reshape(-1, 32)
    d = (np.abs(xb).max(axis=1) / F32(127.0)).astype(F32)
    t = (xb * _inv(d)[:, None]).astype(F32)
    q = (np.sign(t) * np.floor(np.abs(t).astype(np.float64) + 0.5)).astype(np.int8)       # roundf: half away from zero
    return d.astype(np.float16), q


# name -> (ggml_ftype, ggml_type, block bytes, reference -> block bytes)
def _blocks(fmt, ref):
    x = ref.astype(F32)
    if fmt == "q4_0":
        d, qs = q4_0_ref(x); parts = [d.view(np.uint8).reshape(-1, 2), qs]
    elif fmt == "q4_1":
        d, m, qs = q4_1_ref(x); parts = [d.view(np.uint8).reshape(-1, 2), m.view(np.uint8).reshape(-1, 2), qs]
    elif fmt == "q5_0":
        d, qh, qs = q5_0_ref(x); parts = [d.view(np.uint8).reshape(-1, 2), qh.view(np.uint8).reshape(-1, 4), qs]
    elif fmt == "q5_1":
        d, m, qh, qs = q5_1_ref(x); parts = [d.view(np.uint8).reshape(-1, 2), m.view(np.uint8).reshape(-1, 2), qh.view(np.uint8).reshape(-1, 4), qs]
    else:
        d, q = q8_0_ref(x); parts = [d.view(np.uint8).reshape(-1, 2), q.view(np.uint8)]
    return np.concatenate(parts, axis=1)


FORMATS = {"q4_0": (2, 2, 18), "q4_1": (3, 3, 20), "q5_0": (8, 6, 22), "q5_1": (9, 7, 24), "q8_0": (7, 8, 34)}
BLOCK_BYTES = {tt: nb for (_, tt, nb) in FORMATS.values()}


def parse_quantized(path):
    """name -> (ttype, dims, raw bytes) for the three GPT sections; returns also the trailing codec bytes."""
    buf = open(path, "rb").read()
    pos = 0

    def i32(n=1):
        nonlocal pos
        v = struct.unpack_from("<%di" % n, buf, pos); pos += 4 * n
        return v if n > 1 else v[0]
    assert i32() == 0x67676D6C
    for _ in range(i32()):
        ln = i32()
        pos += ln
    out = []
    for _ in range(3):
        hp = i32(10)
        tens = {}
        for _ in range(i32()):
            n_dims, ln, tt = i32(3)
            dims = [i32() for _ in range(n_dims)]
            name = buf[pos:pos + ln].decode(); pos += ln
            nel = int(np.prod(dims)) if dims else 1
            nbytes = nel // 32 * BLOCK_BYTES[tt] if tt in BLOCK_BYTES else nel * (2 if tt == 1 else 4)
            tens[name] = (tt, dims, buf[pos:pos + nbytes]); pos += nbytes
        out.append((hp, tens))
    return out, buf[pos:]


QUANT = [r"model/wte/.*", r"model/lm_head/.*", r"model/h.*/attn/c_attn/w", r"model/h.*/attn/c_proj/w", r"model/h.*/mlp/c_fc/w", r"model/h.*/mlp/c_proj/w"]


@pytest.mark.parametrize("fmt", sorted(FORMATS))
def test_quantize_file(toy_model, tmp_path, fmt):
    from bark_amd_loader import load_package
    lib = load_package().load_library()
    ftype, ttype, nb = FORMATS[fmt]
    dst = str(tmp_path / f"toy_{fmt}.bin")
    assert lib.bark_model_quantize(toy_model.encode(), dst.encode(), ftype)      # GGML_FTYPE_MOSTLY_*
    src = read_model_file(toy_model)
    secs, tail = parse_quantized(dst)
    raw = open(toy_model, "rb").read()
    assert raw.endswith(tail) and tail[:4] == struct.pack("<I", 0x67676D6C)       # codec copied verbatim
    n_q = 0
    for (hp, tens), key in zip(secs, ("semantic", "coarse", "fine")):
        shp, stens = src[key]
        assert hp[9] == 2000 + ftype and list(hp[:9]) == [shp[k] for k in ("n_layer", "n_head", "n_embd", "block_size", "bias", "n_in", "n_out", "n_lm_heads", "n_wtes")]
        assert set(tens) == set(stens)
        for name, (tt, dims, data) in tens.items():
            ref = stens[name]
            want_q = any(re.fullmatch(p, name) for p in QUANT) and ref.ndim == 2
            assert (tt == ttype) == want_q and tt in (0, 1, ttype), name
            assert dims == list(reversed(ref.shape))
            if not want_q:
                assert data == np.ascontiguousarray(ref).tobytes(), name
                continue
            n_q += 1
            blocks = np.frombuffer(data, np.uint8).reshape(-1, nb)
            want = _blocks(fmt, ref)
            assert np.array_equal(blocks, want), f"{name}: {np.flatnonzero((blocks != want).any(axis=1))[:4]}"
    assert n_q == 3 * (4 * 2) + 1 + 1 + 1 + 1 + 8 + 7                            # 4 matrices x 2 layers x 3 models + wte / lm_head tensors


def test_quantize_rejects_other_types_and_bad_paths(toy_model, tmp_path):
    from bark_amd_loader import load_package
    lib = load_package().load_library()
    assert not lib.bark_model_quantize(toy_model.encode(), str(tmp_path / "x.bin").encode(), 10)    # Q2_K: not a bark.cpp quantize type
    assert not lib.bark_model_quantize(b"/nonexistent.bin", str(tmp_path / "y.bin").encode(), 2)
    q = str(tmp_path / "q.bin")
    assert lib.bark_model_quantize(toy_model.encode(), q.encode(), 2)
    assert not lib.bark_model_quantize(q.encode(), str(tmp_path / "qq.bin").encode(), 7)            # already quantised


def test_malformed_model_files_are_rejected_cleanly(toy_model, tmp_path):
    """The container parser (model_file.cpp) on damaged input: truncation anywhere, bad magic, absurd counts - `false` and a
    message, never a crash (bark.cpp:1095-1102 checks the magic; the reference trusts everything after it)."""
    from bark_amd_loader import load_package
    lib = load_package().load_library()
    raw = open(toy_model, "rb").read()
    out = str(tmp_path / "out.bin").encode()
    rng = np.random.default_rng(3)
    cuts = [0, 3, 4, 8, 100, len(raw) // 3, len(raw) - 1] + [int(v) for v in rng.integers(8, len(raw) - 1, 12)]
    for i, cut in enumerate(cuts):
        p = tmp_path / f"cut{i}.bin"
        p.write_bytes(raw[:cut])
        assert not lib.bark_model_quantize(str(p).encode(), out, 2), cut
    bad = bytearray(raw); bad[0] ^= 0xFF
    (tmp_path / "magic.bin").write_bytes(bytes(bad))
    assert not lib.bark_model_quantize(str(tmp_path / "magic.bin").encode(), out, 2)
    bad = bytearray(raw); bad[4:8] = struct.pack("<i", 2**30)                     # vocabulary size
    (tmp_path / "vocab.bin").write_bytes(bytes(bad))
    assert not lib.bark_model_quantize(str(tmp_path / "vocab.bin").encode(), out, 2)
    # a flipped tensor-type field somewhere inside the first GPT section
    off = raw.index(b"model/wpe") - 4 * 3 - 8
    bad = bytearray(raw); bad[off + 8:off + 12] = struct.pack("<i", 77)
    (tmp_path / "ttype.bin").write_bytes(bytes(bad))
    assert not lib.bark_model_quantize(str(tmp_path / "ttype.bin").encode(), out, 2)
