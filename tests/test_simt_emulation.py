"""The few-slot lock-step kernels (default below 17 / 9 live slots since round 5, BARK_HIP_FEW_SLOTS) were written in round 4 without a GPU.  This test
RUNS them - on the host: the kernel sources are compiled for x86 against a stand-in for <hip/hip_runtime.h> (tests/simt/hip/hip_runtime.h: one thread per
work-item, one workgroup at a time, barriers / DPP / readlane as rendezvous) and executed work-item for work-item next to the kernels they must equal bit for
bit, which the device has already been checked with:
    gemv_ln_slots_ps_kernel<PS>      against gemv_ln_wg_kernel<PS> slot after slot (q, the appended K / V rows, every partial score)
    attn_fused_ps_kernel<1 | 2>      against attn_fused_kernel<1 | 2>, the default lock-step attention (f16 output rows)
    gemv_ln_slots_ps_kernel<!PS>     against gemv_ln_wg_kernel (LayerNorm + FC + GELU table)
    gemv_slots_kernel                against gemv_kernel (out-projection + residual)
The sources are patched textually for the host (inline asm, the buffer-load intrinsic binding, dynamic LDS): copies in a temporary directory, the product
is not touched.  What this cannot show: launch geometry and host plumbing (engine_batch.hip) - the xfail-guarded GPU test does that on the device."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.environ.get("BARK_SIM_CSRC", os.path.join(ROOT, "bark.cpp_amd", "csrc"))      # (a deliberately broken copy makes the test fail: its negative control)
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _patch(text: str) -> str:
    # the LLVM buffer-load intrinsics bound by name: plain loads from the descriptor's base address
    text = re.sub(r'__device__ float4v llvm_amdgcn_raw_buffer_load_v4f32\([^;]*;',
                  'inline float4v llvm_amdgcn_raw_buffer_load_v4f32(int4v rsrc, int voffset, int soffset, int) { const char * b = reinterpret_cast<const char *>('
                  '((unsigned long long) (unsigned) rsrc.y << 32) | (unsigned) rsrc.x); float4v r; memcpy(&r, b + voffset + soffset, 16); return r; }', text)
    text = re.sub(r'__device__ float   llvm_amdgcn_raw_buffer_load_f32\([^;]*;',
                  'inline float llvm_amdgcn_raw_buffer_load_f32(int4v rsrc, int voffset, int soffset, int) { const char * b = reinterpret_cast<const char *>('
                  '((unsigned long long) (unsigned) rsrc.y << 32) | (unsigned) rsrc.x); float r; memcpy(&r, b + voffset + soffset, 4); return r; }', text)
    text = text.replace('asm("" : "+v"(v));', '')                                   # the f16 rounding point: no fused conversion to keep apart on the host
    text = re.sub(r'asm volatile\("global_load_dword.*?\)\);', '(void) after;', text)   # prefetch requests: nothing to do
    text = text.replace('asm volatile("; NWPF hold %0" :: "v"(sink));', '(void) sink;')
    text = re.sub(r'extern __shared__ (__attribute__\(\(aligned\(16\)\)\) )?(\w+) (\w+)\[\];', r'static \2 \3[65536];', text)      # dynamic LDS
    return text


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("ROCm's clang is not installed")
    d = str(tmp_path_factory.mktemp("simt"))
    for name in ("kernels.h", "quant_formats.h", "device_utils.h"):
        open(os.path.join(d, name), "w").write(_patch(open(os.path.join(CSRC, name)).read()))
    for name in ("kernels.hip", "attention_kernels.hip", "fast_kernels.hip"):
        open(os.path.join(d, name.replace(".hip", "_sim.hip")), "w").write(_patch(open(os.path.join(CSRC, name)).read()))
    so = os.path.join(d, "libsim.so")
    cmd = [CLANG, "-x", "c++", "-std=c++20", "-O1", "-mfma", "-mf16c", "-mavx2", "-ffp-contract=off", "-pthread", "-shared", "-fPIC", "-Wno-everything",
           "-I", os.path.join(ROOT, "tests", "simt"), "-I", d, os.path.join(ROOT, "tests", "simt", "sim_driver.cpp"), "-o", so]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    lib = C.CDLL(so)
    assert lib.sim_state_size() == 32
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _states(ctxs):
    st = np.zeros((len(ctxs), 8), np.int32)                 # StepState: n_past, cur_token, step, eos_step, near_tie, n_out, last_eos_p, fault
    st[:, 0] = [c - 1 for c in ctxs]
    return st


@pytest.mark.parametrize("E,ctxs", [(128, [5, 300, 700]), (256, [1, 513]), (128, [5, 300, 700, 1, 64, 65, 255, 256, 257, 512, 1000])])      # 11 slots: two slot groups, 8 + 3
def test_slot_kernels_equal_the_kernels_they_replace(sim, E, ctxs):
    rng = np.random.default_rng(E)
    B, H, P = len(ctxs), E // 64, 1024
    stride = H * 16 * P * 4                                # floats of one slot's layer cache (K [H][16][P][4]; V [H][P][64] has the same size)
    W = (rng.standard_normal((3 * E, E)) * 0.08).astype(np.float16)
    x = rng.standard_normal((B, E)).astype(np.float32)
    g, b_ln = (1 + 0.1 * rng.standard_normal(E)).astype(np.float32), (0.1 * rng.standard_normal(E)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(3 * E)).astype(np.float32)
    kc0 = rng.standard_normal(B * stride).astype(np.float32)
    vc0 = rng.standard_normal(B * stride).astype(np.float32)
    out = []
    for route in (0, 1):
        kc, vc = kc0.copy(), vc0.copy()
        q = np.zeros((B, E), np.float32)
        ps = np.full(B * H * 4 * P, np.float32(-77.0))
        st = _states(ctxs)
        assert sim.sim_qkv(route, _p(W), _p(x), _p(g), _p(b_ln), _p(bias), _p(kc), _p(vc), _p(q), _p(ps), _p(st), E, B, C.c_long(stride)) == 0
        assert not st[:, 7].any()                          # no fault flag
        out.append((q, kc, vc, ps))
    for name, a0, a1 in zip(("q", "K cache", "V cache", "partial scores"), out[0], out[1]):
        assert a0.tobytes() == a1.tobytes(), f"QKV per slot: {name} differs from the single-utterance kernel"
    if E % 256 == 0:
        # ... and the DEFAULT lock-step product they replace: gemm_slots16_kernel<LNF> (16 x 16 tiles on v_mfma_f32_16x16x1_4b_f32, LayerNorm in the kernel)
        kc, vc, q16 = kc0.copy(), vc0.copy(), np.zeros((B, E), np.float32)
        assert sim.sim_qkv(2, _p(W), _p(x), _p(g), _p(b_ln), _p(bias), _p(kc), _p(vc), _p(q16), _p(np.zeros(1, np.float32)), _p(_states(ctxs)), E, B, C.c_long(stride)) == 0
        for name, a0, a1 in zip(("q", "K cache", "V cache"), (q16, kc, vc), out[0][:3]):
            assert a0.tobytes() == a1.tobytes(), f"the matrix-core lock-step product: {name} differs from the per-slot kernel"
    q, kc, vc, ps = out[0]
    assert (kc != kc0).sum() == B * E and (vc != vc0).sum() == B * E          # exactly one appended row per slot
    for s, c in enumerate(ctxs):                           # every cached key of every (head, block) scored, nothing beyond
        blk = ps[s * H * 4 * P:(s + 1) * H * 4 * P].reshape(H * 4, P)
        assert (blk[:, :c - 1] != np.float32(-77.0)).all() and (blk[:, c - 1:] == np.float32(-77.0)).all()
    # ---- attention: partial scores + the appended key against K streamed through the workgroup
    for vs in (1, 2):
        att = []
        for route in (0, 1):
            o = np.zeros((B, E), np.float16)
            assert sim.sim_attention(route, vs, _p(q), _p(kc), _p(vc), _p(ps), _p(_states(ctxs)), _p(o), H, B, C.c_long(stride)) == 0
            att.append(o)
        assert att[0].tobytes() == att[1].tobytes(), f"attention on partial scores (VS = {vs}) differs from attn_fused_kernel"
        assert np.isfinite(att[0].astype(np.float32)).all() and np.abs(att[0].astype(np.float32)).max() > 0
    # ---- FC with the LayerNorm in the workgroup, GELU table = identity on the f16 bits
    Wf = (rng.standard_normal((4 * E, E)) * 0.08).astype(np.float16)
    bf = (0.1 * rng.standard_normal(4 * E)).astype(np.float32)
    lut = np.arange(65536, dtype=np.uint16)
    fc = []
    for route in (0, 1):
        o = np.zeros((B, 4 * E), np.float16)
        assert sim.sim_fc(route, _p(Wf), _p(x), _p(g), _p(b_ln), _p(bf), _p(lut), _p(o), E, 4 * E, B) == 0
        fc.append(o)
    assert fc[0].tobytes() == fc[1].tobytes() and np.abs(fc[0].astype(np.float32)).max() > 0
    # ---- the two out-projections with their residual
    for K in (E, 4 * E):
        Wp = (rng.standard_normal((E, K)) * 0.05).astype(np.float16)
        xh = rng.standard_normal((B, K)).astype(np.float16)
        bp = (0.1 * rng.standard_normal(E)).astype(np.float32)
        res = []
        for route in (0, 1):
            r = x.copy()
            assert sim.sim_proj(route, _p(Wp), _p(xh), _p(bp), _p(r), K, E, B) == 0
            res.append(r)
        assert res[0].tobytes() == res[1].tobytes() and (res[0] != x).all()


def test_decode_kernels_run_on_the_host_equal_the_oracle(sim):
    """The same emulation against the ORACLE's own functions (oracle/bark_oracle.cpp: layer_norm_row, gemm_w, attention - the C6 / R1 / C1 / C2 / C4 / C5
    statements): the LayerNorm-fused QKV kernel of a decode step (gemv_ln_wg_kernel<PS>: q, the appended K and V rows), the lock step's attention
    (attn_fused_kernel) and the plain out-projection (gemv_kernel) as the product's source computes them, work-item for work-item on the host - a parity check
    of the kernel SOURCE that needs no GPU (the -m gpu tests check the compiled kernels on the device)."""
    from oracle import pyoracle
    pyoracle.build()
    orc = C.CDLL(pyoracle.LIB_PATH)
    orc.orc_test_wdot.restype = C.c_float
    orc.orc_test_wdot.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    orc.orc_test_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    orc.orc_test_layer_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(11)
    E, ctxs = 128, [9, 411]
    B, H, P = len(ctxs), E // 64, 1024
    stride = H * 16 * P * 4
    W = (rng.standard_normal((3 * E, E)) * 0.08).astype(np.float16)
    x = rng.standard_normal((B, E)).astype(np.float32)
    g, b_ln = (1 + 0.1 * rng.standard_normal(E)).astype(np.float32), (0.1 * rng.standard_normal(E)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(3 * E)).astype(np.float32)
    kc = rng.standard_normal(B * stride).astype(np.float32)
    vc = rng.standard_normal(B * stride).astype(np.float32)
    q = np.zeros((B, E), np.float32)
    ps = np.zeros(B * H * 4 * P, np.float32)
    st = _states(ctxs)
    assert sim.sim_qkv(1, _p(W), _p(x), _p(g), _p(b_ln), _p(bias), _p(kc), _p(vc), _p(q), _p(ps), _p(st), E, B, C.c_long(stride)) == 0
    K4 = kc.reshape(B, H, 16, P, 4)                        # [slot][head][d / 4][position][d % 4]
    V4 = vc.reshape(B, H, P, 64)
    for s, c in enumerate(ctxs):
        y = np.zeros(E, np.float32)
        orc.orc_test_layer_norm(_p(x[s]), _p(y), E, _p(g), _p(b_ln))
        yh = y.astype(np.float16).astype(np.float32)        # R1: the activation entering a weight product is rounded to f16
        want = np.array([np.float32(orc.orc_test_wdot(_p(W[m]), _p(yh), E)) + bias[m] for m in range(3 * E)], np.float32)
        assert q[s].tobytes() == want[:E].tobytes(), f"slot {s}: q differs from the oracle"
        k_row = K4[s, :, :, c - 1, :].reshape(H, 64)        # the appended row, head by head
        v_row = V4[s, :, c - 1, :]
        assert k_row.reshape(-1).tobytes() == want[E:2 * E].tobytes() and v_row.reshape(-1).tobytes() == want[2 * E:].tobytes(), f"slot {s}: appended K / V rows"
    att = np.zeros((B, E), np.float16)
    assert sim.sim_attention(1, 1, _p(q), _p(kc), _p(vc), _p(ps), _p(_states(ctxs)), _p(att), H, B, C.c_long(stride)) == 0
    for s, c in enumerate(ctxs):
        for h in range(H):
            kh = np.ascontiguousarray(K4[s, h, :, :c, :].transpose(1, 0, 2).reshape(c, 64))
            vh = np.ascontiguousarray(V4[s, h, :c, :])
            qh = np.ascontiguousarray(q[s, 64 * h:64 * h + 64])
            o = np.zeros(64, np.float32)
            orc.orc_test_attention(_p(qh), _p(kh), _p(vh), 1, c, c - 1, 1, _p(o))
            assert att[s, 64 * h:64 * h + 64].tobytes() == o.astype(np.float16).tobytes(), f"slot {s} head {h}: attention row differs from the oracle"
    # plain out-projection + residual (gemv_kernel): C1 dot of the f16 row, bias, residual added last
    Wp = (rng.standard_normal((E, E)) * 0.05).astype(np.float16)
    bp = (0.1 * rng.standard_normal(E)).astype(np.float32)
    r = x.copy()
    assert sim.sim_proj(1, _p(Wp), _p(att), _p(bp), _p(r), E, E, B) == 0
    for s in range(B):
        a32 = att[s].astype(np.float32)
        want = np.array([(np.float32(orc.orc_test_wdot(_p(Wp[m]), _p(a32), E)) + bp[m]) + x[s, m] for m in range(E)], np.float32)
        assert r[s].tobytes() == want.tobytes(), f"slot {s}: out-projection + residual differs from the oracle"


@pytest.mark.parametrize("wide", [0, 1])
def test_prefill_product_run_on_the_host_equals_the_oracle(sim, wide):
    """gemm_kernel (the prompt passes of the causal models: persistent workgroups, operands staged through LDS, C1 chains on v_mfma_f32_32x32x2_f32 emulated as
    the k-ordered fmaf chains the device probes established) against the oracle's gemm_w, element by element - ragged N and M, several tiles per workgroup."""
    from oracle import pyoracle
    pyoracle.build()
    orc = C.CDLL(pyoracle.LIB_PATH)
    orc.orc_test_wdot.restype = C.c_float
    orc.orc_test_wdot.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(21)
    N, K, M = (70, 256, 132) if not wide else (70, 256, 192)          # wide = the 32 x 96 tile shape (M a multiple of 96, as launch_linear demands)
    W = (rng.standard_normal((M, K)) * 0.06).astype(np.float16)
    xh = rng.standard_normal((N, K)).astype(np.float16)
    bias = (0.1 * rng.standard_normal(M)).astype(np.float32)
    out = np.zeros((N, M), np.float32)
    assert sim.sim_gemm(_p(W), _p(xh), _p(bias), _p(out), N, K, M, 4, wide) == 0    # 6 tiles on 4 persistent workgroups
    for n in range(0, N, 7):
        x32 = xh[n].astype(np.float32)
        want = np.array([np.float32(orc.orc_test_wdot(_p(W[m]), _p(x32), K)) + bias[m] for m in range(M)], np.float32)
        assert out[n].tobytes() == want.tobytes(), f"row {n} of the prefill product differs from the oracle"


@pytest.mark.parametrize("ctx", [1, 2, 300, 641])
def test_single_utterance_decode_attention_run_on_the_host_equals_the_oracle(sim, ctx):
    """The default decode attention of one utterance: the QKV kernel's partial scores (copies of the q workgroups sized for the launch's context bound),
    the fixed-address copy of the appended K row, V in the K layout, and attn_ps_kernel (one 1024-thread workgroup per head and value quad) - against the
    oracle's attention over the cache this very step extends."""
    from oracle import pyoracle
    pyoracle.build()
    orc = C.CDLL(pyoracle.LIB_PATH)
    orc.orc_test_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.default_rng(ctx)
    E, P = 128, 1024
    H = E // 64
    n = H * 16 * P * 4
    W = (rng.standard_normal((3 * E, E)) * 0.08).astype(np.float16)
    x = rng.standard_normal(E).astype(np.float32)
    g, b_ln = (1 + 0.1 * rng.standard_normal(E)).astype(np.float32), (0.1 * rng.standard_normal(E)).astype(np.float32)
    bias = (0.1 * rng.standard_normal(3 * E)).astype(np.float32)
    kc = rng.standard_normal(n).astype(np.float32)
    v_rows = rng.standard_normal((H, P, 64)).astype(np.float32)           # V [H][P][64] and its copy in the K layout [H][16][P][4]
    vc = v_rows.reshape(-1).copy()
    vt = np.ascontiguousarray(v_rows.reshape(H, P, 16, 4).transpose(0, 2, 1, 3)).reshape(-1).copy()
    q, ps, knew = np.zeros(E, np.float32), np.zeros(H * 4 * P, np.float32), np.zeros(E, np.float32)
    att = np.zeros(E, np.float16)
    st = _states([ctx])
    ng = (ctx + 255) // 256                                               # the graph variant the stage loop would pick
    assert sim.sim_decode_attention(_p(W), _p(x), _p(g), _p(b_ln), _p(bias), _p(kc), _p(vc), _p(vt), _p(q), _p(ps), _p(knew), _p(st), _p(att), E, ng) == 0
    K4, V4 = kc.reshape(H, 16, P, 4), vc.reshape(H, P, 64)
    assert np.array_equal(vt.reshape(H, 16, P, 4)[:, :, ctx - 1, :].reshape(H, 64), V4[:, ctx - 1, :])        # the appended V row reached both layouts
    assert np.array_equal(knew.reshape(H, 64), K4[:, :, ctx - 1, :].reshape(H, 64))
    for h in range(H):
        kh = np.ascontiguousarray(K4[h, :, :ctx, :].transpose(1, 0, 2).reshape(ctx, 64))
        vh = np.ascontiguousarray(V4[h, :ctx, :])
        o = np.zeros(64, np.float32)
        orc.orc_test_attention(_p(np.ascontiguousarray(q[64 * h:64 * h + 64])), _p(kh), _p(vh), 1, ctx, ctx - 1, 1, _p(o))
        assert att[64 * h:64 * h + 64].tobytes() == o.astype(np.float16).tobytes(), f"head {h}: attn_ps_kernel differs from the oracle at context {ctx}"


@pytest.mark.parametrize("tile,N,K,M", [(128, 130, 192, 140), (64, 70, 448, 64)])
def test_fine_model_product_run_on_the_host_equals_the_oracle(sim, tile, N, K, M):
    """gemm_f16_tile_kernel (the fine model's weight products: 128 x 128 / 64 x 64 tiles, operands through LDS, three register sets in flight, guarded tail of the
    K loop) with v_mfma_f32_32x32x16_f16 emulated by the oracle's restatement of that instruction, against the oracle's own C1m product (gemm_mfma): the
    kernel walks K in the canonical order - ascending groups of 8 per accumulator - whatever the pipeline does around it."""
    from oracle import pyoracle
    pyoracle.build()
    orc = C.CDLL(pyoracle.LIB_PATH)
    orc.orc_test_mfma_gemm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.default_rng(K)
    W = (rng.standard_normal((M, K)) * 0.06).astype(np.float16)
    xh = rng.standard_normal((N, K)).astype(np.float16)
    bias = (0.1 * rng.standard_normal(M)).astype(np.float32)
    out = np.zeros((N, M), np.float32)
    assert sim.sim_gemm_f16(_p(W), _p(xh), _p(bias), _p(out), N, K, M, tile) == 0
    want = np.zeros((N, M), np.float32)
    x32 = xh.astype(np.float32)
    orc.orc_test_mfma_gemm(_p(W), _p(x32), M, N, K, _p(want))
    want = (want + bias[None, :]).astype(np.float32)
    assert out.tobytes() == want.tobytes(), "the f16 tile product differs from the oracle's C1m product"


def _attention_reference(orc, q, kc, vc, H, N, ctx, n_past, causal):
    P = 1024
    K4, V4 = kc.reshape(H, 16, P, 4), vc.reshape(H, P, 64)
    want = np.zeros((N, H * 64), np.float16)
    for h in range(H):
        kh = np.ascontiguousarray(K4[h, :, :ctx, :].transpose(1, 0, 2).reshape(ctx, 64))
        vh = np.ascontiguousarray(V4[h, :ctx, :])
        qh = np.ascontiguousarray(q[:, 64 * h:64 * h + 64])
        o = np.zeros((N, 64), np.float32)
        orc.orc_test_attention(_p(qh), _p(kh), _p(vh), N, ctx, n_past, causal, _p(o))
        want[:, 64 * h:64 * h + 64] = o.astype(np.float16)
    return want


def _orc_attention():
    from oracle import pyoracle
    pyoracle.build()
    orc = C.CDLL(pyoracle.LIB_PATH)
    orc.orc_test_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return orc


@pytest.mark.parametrize("kernel", [1, 0])
@pytest.mark.parametrize("N,n_past", [(70, 5), (33, 0), (1, 40), (300, 250), (100, 924)])
def test_prefill_attention_run_on_the_host_equals_the_oracle(sim, N, n_past, kernel):
    """The causal prompt-pass attention against the oracle's: kernel 1 = attn_window_kernel<true> (round 6: scores in registers, only the key tiles in front of
    the query tile's last position are requested, masked scores are -inf, V rows beyond the context are never multiplied - the cache rows there hold NaN in
    this test), kernel 0 = attn_rows_kernel (32-query tiles, the score tile in LDS).  C2 / C4e / C5 on v_mfma_f32_32x32x2_f32."""
    orc = _orc_attention()
    rng = np.random.default_rng(N)
    H, P = 2, 1024
    ctx = n_past + N
    kc = rng.standard_normal(H * 16 * P * 4).astype(np.float32)
    vc = rng.standard_normal(H * P * 64).astype(np.float32)
    # rows at or beyond the context were never written: poison them (BARK_HIP_POISON does the same to the engine's caches) - nothing may leak into the result
    kc.reshape(H, 16, P, 4)[:, :, ctx:, :] = np.nan
    vc.reshape(H, P, 64)[:, ctx:, :] = np.nan
    q = rng.standard_normal((N, H * 64)).astype(np.float32)
    att = np.zeros((N, H * 64), np.float16)
    assert sim.sim_attention_rows(kernel, _p(q), _p(kc), _p(vc), _p(att), H, N, n_past, 1) == 0
    assert att.tobytes() == _attention_reference(orc, q, kc, vc, H, N, ctx, n_past, 1).tobytes()


def test_fine_window_attention_run_on_the_host_equals_the_oracle(sim):
    """attn_window_kernel (the fine model's whole-window attention with the scores in registers: transposed score tiles with permuted key rows, every wave owns
    two C5 chains) and attn_rows_kernel on the same window, both against the oracle - one head, 1024 x 1024."""
    orc = _orc_attention()
    rng = np.random.default_rng(77)
    H, P, N = 1, 1024, 1024
    kc = rng.standard_normal(H * 16 * P * 4).astype(np.float32)
    vc = rng.standard_normal(H * P * 64).astype(np.float32)
    q = rng.standard_normal((N, H * 64)).astype(np.float32)
    want = _attention_reference(orc, q, kc, vc, H, N, N, 0, 0)
    for kernel in (1, 0):
        att = np.zeros((N, H * 64), np.float16)
        assert sim.sim_attention_rows(kernel, _p(q), _p(kc), _p(vc), _p(att), H, N, 0, 0) == 0
        assert att.tobytes() == want.tobytes(), ("attn_window_kernel" if kernel else "attn_rows_kernel") + " differs from the oracle on a whole window"
