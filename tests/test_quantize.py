"""bark_model_quantize (bark.h:229-232): the native Q4_0 writer against an independent numpy restatement of ggml's
reference block quantiser, and the container rules of /root/reference/bark.cpp:272-478,2234-2377."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

from tools.make_hf_golden import read_model_file


def q4_0_ref(x):
    """quantize_row_q4_0_ref: 32-element blocks -> (f16 d, 16 nibble bytes)."""
    xb = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 32)
    idx = np.argmax(np.abs(xb), axis=1)                       # first element of largest magnitude
    mx = xb[np.arange(len(xb)), idx]
    d = (mx / np.float32(-8.0)).astype(np.float32)
    inv = np.where(d != 0, np.float32(1.0) / np.where(d != 0, d, 1).astype(np.float32), np.float32(0.0)).astype(np.float32)
    q = np.minimum(15, (xb * inv[:, None] + np.float32(8.5)).astype(np.float32).astype(np.int8).astype(np.int32)).astype(np.uint8)
    qs = (q[:, :16] | (q[:, 16:] << 4)).astype(np.uint8)
    return d.astype(np.float16), qs


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
            nbytes = nel // 32 * 18 if tt == 2 else nel * (2 if tt == 1 else 4)
            tens[name] = (tt, dims, buf[pos:pos + nbytes]); pos += nbytes
        out.append((hp, tens))
    return out, buf[pos:]


QUANT = [r"model/wte/.*", r"model/lm_head/.*", r"model/h.*/attn/c_attn/w", r"model/h.*/attn/c_proj/w", r"model/h.*/mlp/c_fc/w", r"model/h.*/mlp/c_proj/w"]


def test_quantize_q4_0_file(toy_model, tmp_path):
    from bark_amd_loader import load_package
    lib = load_package().load_library()
    dst = str(tmp_path / "toy_q4_0.bin")
    assert lib.bark_model_quantize(toy_model.encode(), dst.encode(), 2)          # GGML_FTYPE_MOSTLY_Q4_0
    src = read_model_file(toy_model)
    secs, tail = parse_quantized(dst)
    raw = open(toy_model, "rb").read()
    assert raw.endswith(tail) and tail[:4] == struct.pack("<I", 0x67676D6C)       # codec copied verbatim
    n_q = 0
    for (hp, tens), key in zip(secs, ("semantic", "coarse", "fine")):
        shp, stens = src[key]
        assert hp[9] == 2002 and list(hp[:9]) == [shp[k] for k in ("n_layer", "n_head", "n_embd", "block_size", "bias", "n_in", "n_out", "n_lm_heads", "n_wtes")]
        assert set(tens) == set(stens)
        for name, (tt, dims, data) in tens.items():
            ref = stens[name]
            want_q = any(re.fullmatch(p, name) for p in QUANT) and ref.ndim == 2
            assert (tt == 2) == want_q, name
            assert dims == list(reversed(ref.shape))
            if not want_q:
                assert data == np.ascontiguousarray(ref).tobytes(), name
                continue
            n_q += 1
            d, qs = q4_0_ref(ref.astype(np.float32))
            blocks = np.frombuffer(data, np.uint8).reshape(-1, 18)
            assert np.array_equal(blocks[:, :2].copy().view(np.float16).ravel(), d), name
            assert np.array_equal(blocks[:, 2:], qs), name
    assert n_q == 3 * (4 * 2) + 1 + 1 + 1 + 1 + 8 + 7                            # 4 matrices x 2 layers x 3 models + wte / lm_head tensors


def test_quantize_rejects_other_types_and_bad_paths(toy_model, tmp_path):
    from bark_amd_loader import load_package
    lib = load_package().load_library()
    assert not lib.bark_model_quantize(toy_model.encode(), str(tmp_path / "x.bin").encode(), 7)     # Q8_0: not implemented
    assert not lib.bark_model_quantize(b"/nonexistent.bin", str(tmp_path / "y.bin").encode(), 2)
