"""Pins the canonical summation orders (DESIGN.md "Canonical numerics") with an INDEPENDENT, exact
restatement: Python Fractions evaluate every fused multiply-add exactly and round once to float32, in
the order the specification states.  The oracle's vectorised code must agree bit for bit."""
import ctypes as C
import math
from fractions import Fraction

import numpy as np
import pytest

from oracle import pyoracle


@pytest.fixture(scope="module")
def lib():
    pyoracle.build()
    L = C.CDLL(pyoracle.LIB_PATH)
    L.orc_test_wdot.restype = C.c_float
    L.orc_test_wdot.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.orc_test_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.orc_test_q4dot.restype = C.c_float
    L.orc_test_q4dot.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.orc_test_layer_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    return L


def rn32(fr: Fraction) -> np.float32:
    """Round an exact rational to the nearest float32, ties to even (normal range)."""
    if fr == 0:
        return np.float32(0.0)
    sign = -1 if fr < 0 else 1
    a = abs(fr)
    e = math.floor(math.log2(float(a)))
    while Fraction(2) ** e > a:
        e -= 1
    while Fraction(2) ** (e + 1) <= a:
        e += 1
    e = max(e, -126)
    q = a / Fraction(2) ** (e - 23)                 # significand scaled to an integer grid
    n = q.numerator // q.denominator
    rem = q - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
        n += 1
    return np.float32(sign * float(Fraction(n) * Fraction(2) ** (e - 23)))


def fma32(a, b, c) -> np.float32:
    return rn32(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))


def add32(a, b) -> np.float32:
    return rn32(Fraction(float(a)) + Fraction(float(b)))


def tree16(v):
    v = list(v)
    for st in (1, 2, 4, 8):
        for c in range(0, 16, 2 * st):
            v[c] = add32(v[c], v[c + st])
    return v[0]


def c1_dot(w, x):
    """C1: 8-element chunks, chunk q -> chain q % 16, chains walk their chunks in ascending order."""
    K = len(w)
    acc = [np.float32(0.0)] * 16
    for q in range((K + 7) // 8):
        c = q % 16
        for k in range(8 * q, min(8 * q + 8, K)):
            acc[c] = fma32(w[k], x[k], acc[c])
    return tree16(acc)


@pytest.mark.parametrize("K", [128, 256, 384, 768])
def test_c1_weight_dot(lib, K):
    rng = np.random.default_rng(K)
    w = (rng.standard_normal(K) * 0.05).astype(np.float16)
    x = rng.standard_normal(K).astype(np.float16).astype(np.float32)      # f16-rounded activations
    got = lib.orc_test_wdot(w.ctypes.data, x.ctypes.data, K)
    want = c1_dot(w.astype(np.float32), x)
    assert np.float32(got) == want, (got, want)
    # and it is NOT what a plain left-to-right sum gives in general (the order matters)
    seq = np.float32(0.0)
    for k in range(K):
        seq = fma32(w[k], x[k], seq)
    assert abs(float(seq) - float(want)) < 1e-4


def canon_expf(x) -> np.float32:
    """C4e in exact rational arithmetic, one rounding per stated operation: n = round(x log2 e) by the 1.5 x 2^23 shift,
    b = x - n ln2 (hi, lo), 2^n from the exponent bits, degree-5 polynomial - the vector expf of ARM's optimised routines / ggml_v_expf."""
    f = np.float32
    h = lambda t: f(float.fromhex(t))
    x = f(x)
    r = h("0x1.8p23")
    z = fma32(x, h("0x1.715476p+0"), r)
    n = add32(z, -r)
    b = fma32(-n, h("0x1.62e4p-1"), x)
    b = fma32(-n, h("0x1.7f7d1cp-20"), b)
    k = np.array([(int(np.array([z], f).view(np.uint32)[0]) << 23) + 0x3f800000 & 0xffffffff], np.uint32).view(f)[0]
    u = rn32(Fraction(float(b)) * Fraction(float(b)))
    c1b = rn32(Fraction(float(h("0x1.ffffecp-1"))) * Fraction(float(b)))
    j = fma32(fma32(fma32(h("0x1.0e4020p-7"), b, h("0x1.573e2ep-5")), u, fma32(h("0x1.555e66p-3"), b, h("0x1.fffdb6p-2"))), u, c1b)
    return f(0.0) if n < -125 else fma32(k, j, k)


def test_c4e_exponential_is_the_stated_routine_and_accurate(lib):
    """The oracle's C4e routine equals the exact-arithmetic restatement bit for bit and stays within 2 ulp of exp()."""
    lib.bark_oracle_canon_expf.restype = C.c_float
    lib.bark_oracle_canon_expf.argtypes = [C.c_float]
    rng = np.random.default_rng(4)
    xs = np.concatenate([-rng.random(600) * 12, -rng.random(200) * 86, [0.0, -1e-7, -3e-4, -86.0, -86.5, -86.9, -87.5, -200.0, -1e4]]).astype(np.float32)
    worst = 0.0
    for x in xs:
        got = np.float32(lib.bark_oracle_canon_expf(float(x)))
        want = canon_expf(x)
        assert got.tobytes() == want.tobytes(), (x, got, want)
        t = math.exp(float(x))
        if x > -86.0:
            worst = max(worst, abs(float(got) - t) / float(np.spacing(np.float32(t))))
        elif x < -87.4:
            assert got == 0.0
    assert worst < 2.0, worst


def softmax_rows(s, valid):
    mx = np.float32(max(s[:valid]))
    e = [canon_expf(np.float32(v - mx)) for v in s[:valid]]
    tot = 0.0
    for v in e:
        tot += float(v)                              # double accumulation
    inv = np.float32(1.0 / tot)
    return [np.float32(v * inv) for v in e]


@pytest.mark.parametrize("N,ctx,n_past,causal", [(1, 37, 36, 1), (3, 40, 37, 1), (5, 33, 0, 0)])
def test_c2_c5_attention(lib, N, ctx, n_past, causal):
    rng = np.random.default_rng(100 * N + ctx)
    q = rng.standard_normal((N, 64)).astype(np.float32)
    k = rng.standard_normal((ctx, 64)).astype(np.float32)
    v = rng.standard_normal((ctx, 64)).astype(np.float32)
    out = np.zeros((N, 64), np.float32)
    lib.orc_test_attention(q.ctypes.data, k.ctypes.data, v.ctypes.data, N, ctx, n_past, causal, out.ctypes.data)
    for i in range(N):
        valid = min(ctx, n_past + i + 1) if causal else ctx
        s = []
        for j in range(valid):                       # C2: four chains over the 16-d blocks, (c0 + c1) + (c2 + c3)
            blk = []
            for b in range(4):
                acc = np.float32(0.0)
                for d in range(16 * b, 16 * b + 16):
                    acc = fma32(k[j, d], q[i, d], acc)
                blk.append(acc)
            acc = add32(add32(blk[0], blk[1]), add32(blk[2], blk[3]))
            s.append(np.float32(acc * np.float32(0.125)))
        p = softmax_rows(s, valid)
        for d in range(0, 64, 13):                   # C5: key j -> chain j % 16, tree-combined
            acc = [np.float32(0.0)] * 16
            for j in range(valid):
                acc[j % 16] = fma32(v[j, d], p[j], acc[j % 16])
            assert out[i, d] == tree16(acc), (i, d)


def test_layer_norm_double_sums(lib):
    rng = np.random.default_rng(5)
    E = 256
    x = (rng.standard_normal(E) * 3 + 1).astype(np.float32)
    g = (1 + 0.05 * rng.standard_normal(E)).astype(np.float32)
    b = (0.02 * rng.standard_normal(E)).astype(np.float32)
    y = np.zeros(E, np.float32)
    lib.orc_test_layer_norm(x.ctypes.data, y.ctypes.data, E, g.ctypes.data, b.ctypes.data)
    mean = np.float32(sum(float(v) for v in x) / E)
    d = (x - mean).astype(np.float32)
    var = np.float32(sum(float(np.float32(v * v)) for v in d) / E)
    scale = np.float32(1.0) / np.sqrt(np.float32(var + np.float32(1e-5)))
    want = ((d * scale).astype(np.float32) * g).astype(np.float32) + b
    assert np.array_equal(y, want.astype(np.float32))


@pytest.mark.parametrize("K", [128, 768])
def test_c1q_q4_0_times_q8_0_dot(lib, K):
    """q4_0 weights x q8_0-quantised activations (ggml_vec_dot_q4_0_q8_0) in the canonical block-chain order."""
    from tests.test_quantize import q4_0_ref
    rng = np.random.default_rng(K)
    w = (rng.standard_normal(K) * 0.05).astype(np.float32)
    x = rng.standard_normal(K).astype(np.float32)
    d4, qs = q4_0_ref(w[None, :])
    blocks = np.concatenate([d4.view(np.uint8).reshape(-1, 2), qs], axis=1).astype(np.uint8).copy()
    got = np.float32(lib.orc_test_q4dot(blocks.ctypes.data, x.ctypes.data, K))
    acc = [np.float32(0.0)] * 16
    for b in range(K // 32):
        xb = x[32 * b:32 * b + 32]
        amax = np.float32(np.max(np.abs(xb)))
        d = np.float32(amax / np.float32(127.0))
        inv = np.float32(1.0) / d if d != 0 else np.float32(0.0)
        q8 = [int(math.copysign(math.floor(abs(float(np.float32(v * inv))) + 0.5), float(v))) for v in xb]     # roundf: half away from zero
        d8 = np.float32(np.float16(d))
        q4 = [int(qs[b, j] & 15) - 8 for j in range(16)] + [int(qs[b, j] >> 4) - 8 for j in range(16)]
        sumi = sum(a * c for a, c in zip(q4, q8))
        t = np.float32(np.float32(np.float32(sumi) * np.float32(d4[b])) * d8)
        acc[b % 16] = add32(acc[b % 16], t)
    assert got == tree16(acc)


@pytest.mark.parametrize("fmt", ["q4_0", "q4_1", "q5_0", "q5_1", "q8_0"])
def test_c1q_all_block_formats(lib, fmt):
    """Every restated ggml block format x q8_0 / q8_1 activations: integer levels decoded from the bytes independently of
    the oracle, block terms and chains in numpy float32."""
    from tests.test_quantize import FORMATS, _blocks
    K = 640                                                   # 20 blocks: chains 0..3 hold two blocks, the rest one
    rng = np.random.default_rng(len(fmt) * 131 + ord(fmt[1]) + ord(fmt[3]))
    w = (rng.standard_normal(K) * 0.05 + 0.01).astype(np.float32)
    x = rng.standard_normal(K).astype(np.float32)
    _, ttype, nb = FORMATS[fmt]
    blocks = np.ascontiguousarray(_blocks(fmt, w[None, :]).astype(np.uint8))
    assert blocks.shape == (K // 32, nb)
    lib.orc_test_qdot.restype = C.c_float
    lib.orc_test_qdot.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    got = np.float32(lib.orc_test_qdot(ttype, blocks.ctypes.data, x.ctypes.data, K))
    f32 = np.float32
    acc = [f32(0.0)] * 16
    for b in range(K // 32):
        blk = blocks[b]
        dw = f32(blk[0:2].copy().view(np.float16)[0]); pos = 2
        mw = f32(0.0)
        if fmt in ("q4_1", "q5_1"):
            mw = f32(blk[2:4].copy().view(np.float16)[0]); pos = 4
        if fmt == "q8_0":
            wq = [int(v) for v in blk[2:34].copy().view(np.int8)]
        else:
            qh = 0
            if fmt in ("q5_0", "q5_1"):
                qh = int(blk[pos:pos + 4].copy().view(np.uint32)[0]); pos += 4
            qs = blk[pos:pos + 16]
            wq = [int(qs[j] & 15) | (((qh >> j) & 1) << 4) for j in range(16)] + [int(qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4) for j in range(16)]
            wq = [v - {"q4_0": 8, "q5_0": 16}.get(fmt, 0) for v in wq]
        xb = x[32 * b:32 * b + 32]
        amax = f32(np.max(np.abs(xb)))
        d = f32(amax / f32(127.0))
        inv = f32(1.0) / d if d != 0 else f32(0.0)
        q8 = [int(math.copysign(math.floor(abs(float(f32(v * inv))) + 0.5), float(v))) for v in xb]     # roundf: half away from zero
        dx = f32(np.float16(d))
        sx = f32(np.float16(f32(f32(sum(q8)) * d)))
        sumi = sum(a * c for a, c in zip(wq, q8))
        if fmt == "q4_0":
            t = f32(f32(f32(sumi) * dw) * dx)
        else:
            t = f32(f32(dw * dx) * f32(sumi))
            if fmt in ("q4_1", "q5_1"):
                t = add32(t, f32(mw * sx))
        acc[b % 16] = add32(acc[b % 16], t)
    assert got == tree16(acc)


@pytest.mark.parametrize("K", [768, 384, 640, 896])
def test_division_by_the_row_length_is_correctly_rounded(K):
    """device_utils.h div_by_const<K>: q0 = RN(a y), r = a - K q0 (exact, one fma), q = RN(q0 + r y) with y = RN(1 / K) must be the
    correctly rounded a / K for every double a - it replaces the fp64 division of the LayerNorm sums in the decode kernels.
    Restated with exact rationals (float(Fraction) rounds to nearest even)."""
    rng = np.random.default_rng(K)
    y = float(Fraction(1, K))
    cases = [float(v) for v in rng.standard_normal(20000) * 10.0 ** rng.uniform(-6, 6, 20000)]
    cases += [float(np.float32(v)) * K for v in rng.standard_normal(2000)]            # quotients that are floats: rounding boundaries of the later f32 cast
    cases += [float(np.nextafter(np.float64(c), np.inf)) for c in cases[:4000]] + [0.0, 1.0, float(K), K * (1.0 + 2.0 ** -52), 2.0 ** -30, 1e300 / K]
    for a in cases:
        fa = Fraction(a)
        q0 = float(fa * Fraction(y))
        r_exact = fa - K * Fraction(q0)
        r = float(r_exact)
        assert Fraction(r) == r_exact, "the residual must be exact in double precision"
        q = float(Fraction(q0) + Fraction(r) * Fraction(y))
        assert q == float(fa / K), (a, q, float(fa / K))
