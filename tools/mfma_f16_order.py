"""What arithmetic does v_mfma_f32_32x32x16_f16 implement?  (can the fine model's products move to
the f16 matrix cores under a canonical order a CPU can restate bit for bit?)

  python tools/mfma_f16_order.py gen  OUT.npz     on the GPU box: builds trial inputs, runs tools/probes/mfma_f16_order_probe, stores raw
                                                  inputs + outputs of the four instruction variants
  python tools/mfma_f16_order.py fit  OUT.npz     anywhere: decodes the dump and scores candidate CPU restatements (exact rational
                                                  arithmetic, fractions.Fraction) against the device bits

Trial families (logical 32x32x16 operands A[i][k], B[k][j], C[i][j]):
  rand   random f16 operands, exponent spreads 0 / 2 / 6 / 12, random signs, C random or 0
  group  one BIG product (or C = BIG) and two half-ulp products at chosen k slots: are the two small ones added to each other before they
         meet the big one (exact or tree sums) or one by one (a chain)?
  align  p0 = 2^24 at slot k0 (or C = +-2^24 / 2^24 + 2), p1 = 2^(4 - i) at k1, p2 = +-2^(2 - j / 2) at k2: alignment width, sticky bits, rounding
"""
import itertools
import os
import subprocess
import sys
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PROBE = os.path.join(HERE, "probes", "mfma_f16_order_probe")


# ---- logical <-> lane layouts ---------------------------------------------------------------------------------------
def pack32(A, B, C):
    """A [32][16], B [16][32] (float, f16-representable), C [32][32] f32 -> raw lanes of the 32x32x16 instruction."""
    a = np.zeros((64, 8), np.float16); b = np.zeros((64, 8), np.float16); c = np.zeros((64, 16), np.float32)
    for lane in range(64):
        h, r = divmod(lane, 32)
        a[lane] = A[r, 8 * h:8 * h + 8]
        b[lane] = B[8 * h:8 * h + 8, r]
        for v in range(16):
            c[lane, v] = C[8 * (v // 4) + 4 * h + v % 4, r]
    return a, b, c


def unpack32_D(d):
    D = np.zeros((32, 32), np.float32)
    for lane in range(64):
        h, r = divmod(lane, 32)
        for v in range(16):
            D[8 * (v // 4) + 4 * h + v % 4, r] = d[lane, v]
    return D


def unpack32_AB(a, b, kper=8):
    """raw lanes -> logical A [32][2 kper], B [2 kper][32] of the 32x32 instructions (kper halves of a lane are used)"""
    A = np.zeros((32, 2 * kper), np.float64); B = np.zeros((2 * kper, 32), np.float64)
    for lane in range(64):
        h, r = divmod(lane, 32)
        A[r, kper * h:kper * h + kper] = a[lane, :kper].astype(np.float64)
        B[kper * h:kper * h + kper, r] = b[lane, :kper].astype(np.float64)
    return A, B


def unpack16(a, b, c, d):
    """16x16x32: A [16][32], B [32][16], C / D [16][16] (registers 0..3)"""
    A = np.zeros((16, 32), np.float64); B = np.zeros((32, 16), np.float64); C = np.zeros((16, 16), np.float32); D = np.zeros((16, 16), np.float32)
    for lane in range(64):
        q, r = divmod(lane, 16)
        A[r, 8 * q:8 * q + 8] = a[lane].astype(np.float64)
        B[8 * q:8 * q + 8, r] = b[lane].astype(np.float64)
        for v in range(4):
            C[4 * q + v, r] = c[lane, v]; D[4 * q + v, r] = d[lane, v]
    return A, B, C, D


# ---- trial generation -----------------------------------------------------------------------------------------------
def gen_trials(seed=1234):
    rng = np.random.default_rng(seed)
    trials = []          # (family, meta, A, B, C)

    def rnd16(shape, spread):
        mant = 1.0 + rng.integers(0, 1024, size=shape) / 1024.0
        ex = rng.integers(-spread, spread + 1, size=shape) if spread else np.zeros(shape, np.int64)
        sign = np.where(rng.integers(0, 2, size=shape) == 1, -1.0, 1.0)
        return sign * mant * np.exp2(ex)

    for spread in (0, 2, 6, 12):
        for cmode in (0, 1, 2):
            for rep in range(4):
                A = rnd16((32, 16), spread); B = rnd16((16, 32), spread)
                if cmode == 0: C = np.zeros((32, 32))
                else:
                    mant = 1.0 + rng.integers(0, 1 << 23, size=(32, 32)) / float(1 << 23)
                    ex = rng.integers(-spread - 2, spread + 3, size=(32, 32)) + (4 if cmode == 2 else 0)
                    C = np.where(rng.integers(0, 2, size=(32, 32)) == 1, -1.0, 1.0) * mant * np.exp2(ex)
                trials.append(("rand", (spread, cmode, rep), A, B, C))
    # group: BIG = 2^14 (f32 ulp 2^-9), small = 2^-10 * 2^-(j % 4) (half / quarter / ... ulp)
    for kb in range(16):
        for d in range(1, 16):
            A = np.zeros((32, 16)); B = np.ones((16, 32)); C = np.zeros((32, 32))
            for i in range(32):
                ka = i % 16
                k2 = (ka + d) % 16
                if i < 16: A[i, kb] = 2.0 ** 14
                else: C[i, :] = 2.0 ** 14
                # the small products (overwriting BIG when they collide with kb: that row is then a pure sum of small products)
                A[i, ka] = 2.0 ** -10; A[i, k2] = 2.0 ** -10
            for j in range(32):
                B[:, j] = 2.0 ** -(j % 4)
                B[kb, j] = 1.0
            # rows whose small slots collide with kb keep B[kb] = 1: recorded in meta, skipped by the analysis
            trials.append(("group", (kb, d), A, B, C))
    # align
    slot_sets = [(0, 1, 2), (0, 1, 8), (0, 4, 8), (3, 7, 15), (15, 0, 8), (8, 9, 0), (5, 6, 7), (0, 15, 1)]
    for (k0, k1, k2) in slot_sets:
        for cmode in range(5):          # 0: p0 in A/B; 1: C = 2^24; 2: C = 2^24 + 2; 3: C = -2^24; 4: C = 2^24 and p0 too
            A = np.zeros((32, 16)); B = np.zeros((16, 32)); C = np.zeros((32, 32))
            if cmode in (0, 4):
                A[:, k0] = 2.0 ** 10; B[k0, :] = 2.0 ** 14
            if cmode == 1 or cmode == 4: C[:, :] = 2.0 ** 24
            if cmode == 2: C[:, :] = 2.0 ** 24 + 2.0
            if cmode == 3: C[:, :] = -(2.0 ** 24)
            for i in range(32):
                A[i, k1] = 2.0 ** (4 - i) if i < 19 else 0.0
            B[k1, :] = 1.0
            A[:, k2] = 1.0
            for j in range(32):
                B[k2, j] = (1.0 if j % 2 == 0 else -1.0) * 2.0 ** (2 - j // 2)
            trials.append(("align", (k0, k1, k2, cmode), A, B, C))
    return trials


def cmd_gen(out):
    trials = gen_trials()
    n = len(trials)
    a = np.zeros((n, 64, 8), np.float16); b = np.zeros((n, 64, 8), np.float16); c = np.zeros((n, 64, 16), np.float32)
    for t, (_, _, A, B, C) in enumerate(trials):
        assert np.array_equal(A.astype(np.float16).astype(np.float64), A) and np.array_equal(B.astype(np.float16).astype(np.float64), B)
        a[t], b[t], c[t] = pack32(A, B, C.astype(np.float32))
    tmp_in, tmp_out = out + ".in.bin", out + ".out.bin"
    with open(tmp_in, "wb") as f:
        f.write(np.int32(n).tobytes()); f.write(a.tobytes()); f.write(b.tobytes()); f.write(c.tobytes())
    if not os.path.exists(PROBE):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", PROBE, PROBE + ".hip"])
    subprocess.check_call([PROBE, tmp_in, tmp_out])
    d = np.fromfile(tmp_out, np.float32).reshape(4, n, 64, 16)
    fam = np.array([t[0] for t in trials]); meta = np.array([list(t[1]) + [0] * (4 - len(t[1])) for t in trials], np.int32)
    np.savez_compressed(out, a=a.view(np.uint16), b=b.view(np.uint16), c=c, d=d, family=fam, meta=meta)
    os.remove(tmp_in); os.remove(tmp_out)
    print(f"wrote {out}: {n} trials")


# ---- candidate restatements -------------------------------------------------------------------------------------------
def f32_round(x: Fraction, mode="rne") -> np.float32:
    """Fraction -> nearest float32 (round to nearest even; 'rz' toward zero); no overflow / subnormal handling beyond numpy's"""
    if x == 0: return np.float32(0.0)
    sign = -1 if x < 0 else 1
    ax = abs(x)
    e = ax.numerator.bit_length() - ax.denominator.bit_length()
    if Fraction(2) ** e > ax: e -= 1
    if Fraction(2) ** (e + 1) <= ax: e += 1
    e = max(e, -126)
    q = ax / Fraction(2) ** (e - 23)
    fl = q.numerator // q.denominator
    rem = q - fl
    if mode == "rne":
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (fl & 1)): fl += 1
    return np.float32(sign * float(fl) * 2.0 ** (e - 23))


def trunc_to(x: Fraction, lsb_exp: int, mode: str) -> Fraction:
    """x aligned to a grid of 2^lsb_exp: 'rz' toward zero, 'floor' toward -inf (two's complement truncation), 'rne'"""
    q = x / Fraction(2) ** lsb_exp
    if mode == "floor": n = q.numerator // q.denominator
    elif mode == "rz": n = abs(q.numerator) // q.denominator * (1 if q >= 0 else -1)
    else:
        fl = q.numerator // q.denominator; rem = q - fl
        n = fl + (1 if (rem > Fraction(1, 2) or (rem == Fraction(1, 2) and fl & 1)) else 0)
    return Fraction(n) * Fraction(2) ** lsb_exp


def exp_of(x: Fraction) -> int:
    ax = abs(x)
    e = ax.numerator.bit_length() - ax.denominator.bit_length()
    if Fraction(2) ** e > ax: e -= 1
    if Fraction(2) ** (e + 1) <= ax: e += 1
    return e


def cand_exact(p, c):                       # one rounding of the exact sum
    return f32_round(sum(p, Fraction(0)) + c)


def cand_chain(p, c):                       # ascending fmaf chain
    acc = c
    for x in p: acc = Fraction(float(f32_round(acc + x)))
    return np.float32(float(acc))


def make_grouped(gs, inner="exact", c_first=True):
    """groups of gs consecutive k: each group's exact sum (rounded to f32 when inner == 'rne'), then added to the accumulator one by one"""
    def f(p, c):
        acc = c
        for g in range(0, len(p), gs):
            s = sum(p[g:g + gs], Fraction(0))
            if inner == "rne": s = Fraction(float(f32_round(s)))
            acc = Fraction(float(f32_round(acc + s)))
        return np.float32(float(acc))
    return f


def make_aligned(gs, width, tmode, with_c=True, rmode="rne"):
    """groups of gs products (+ the accumulator in the first group when with_c) are aligned to the largest exponent of the group, every addend
    truncated (tmode) to `width` bits below that exponent, summed exactly, rounded to f32; the groups chain through the accumulator"""
    def f(p, c):
        acc = c
        for g in range(0, len(p), gs):
            terms = list(p[g:g + gs]) + [acc]
            nz = [t for t in terms if t != 0]
            if not nz: acc = Fraction(0); continue
            emax = max(exp_of(t) for t in nz)
            s = sum((trunc_to(t, emax - width, tmode) for t in terms), Fraction(0))
            acc = Fraction(float(f32_round(s, rmode)))
        return np.float32(float(acc))
    return f


def candidates():
    c = {"exact_single_rounding": cand_exact, "ascending_fmaf_chain": cand_chain}
    for gs in (2, 4, 8):
        c[f"groups_of_{gs}_exact_then_chain"] = make_grouped(gs, "exact")
        c[f"groups_of_{gs}_rne_then_chain"] = make_grouped(gs, "rne")
    for gs in (4, 8, 16):
        for width in (23, 24, 25, 26, 27, 28, 30, 32):
            for tmode in ("rz", "floor"):
                c[f"aligned_g{gs}_w{width}_{tmode}"] = make_aligned(gs, width, tmode)
    return c


def cmd_fit(path, max_elems=200):
    z = np.load(path)
    a = z["a"].view(np.float16); b = z["b"].view(np.float16); c = z["c"]; d = z["d"]; fam = z["family"]; meta = z["meta"]
    cands = candidates()
    rng = np.random.default_rng(7)
    for variant, name in ((0, "32x32x16"), (2, "32x32x8")):
        kper = 8 if variant == 0 else 4
        score = {k: [0, 0] for k in cands}
        for t in range(len(fam)):
            if fam[t] != "rand": continue
            A, B = unpack32_AB(a[t], b[t], kper)
            C = unpack32_D(c[t]); D = unpack32_D(d[variant, t])
            picks = rng.integers(0, 32, size=(max(1, max_elems // 48), 2))
            for i, j in picks:
                p = [Fraction(A[i, k]) * Fraction(B[k, j]) for k in range(2 * kper)]
                cc = Fraction(float(C[i, j]))
                for k, fn in cands.items():
                    got = fn(p, cc)
                    score[k][1] += 1
                    score[k][0] += int(got.tobytes() == D[i, j].tobytes())
        print(f"== v_mfma_f32_{name}_f16, random trials: candidates by agreement")
        for k, (ok, n) in sorted(score.items(), key=lambda kv: -kv[1][0])[:12]:
            print(f"  {k:40s} {ok}/{n}")


# ---- the oracle's restatement (oracle/mfma_f16_emu.h) against the dump and against fresh device runs -------------------------------
def emu_lib():
    import ctypes as C
    path = os.path.join(HERE, "..", "oracle", "build", "libmfma_emu.so")
    if not os.path.exists(path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fopenmp", "-fPIC", "-shared", "-o", path, os.path.join(HERE, "..", "oracle", "mfma_f16_emu_capi.cpp")])
    lib = C.CDLL(path)
    lib.mfma_emu_chain_trials.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return lib


def emu_chain(a16, b16, c32, KB):
    """a16, b16: [n][KB][64][8] uint16, c32: [n][64][16] f32 -> [n][64][16] f32 by the CPU restatement"""
    lib = emu_lib()
    a16 = np.ascontiguousarray(a16, np.uint16); b16 = np.ascontiguousarray(b16, np.uint16); c32 = np.ascontiguousarray(c32, np.float32)
    n = c32.shape[0]
    out = np.zeros_like(c32)
    lib.mfma_emu_chain_trials(a16.ctypes.data, b16.ctypes.data, c32.ctypes.data, n, KB, out.ctypes.data)
    return out


def cmd_check(path):
    """the C restatement against the first dump (single issues, variant 0)"""
    z = np.load(path)
    n = z["c"].shape[0]
    got = emu_chain(z["a"].reshape(n, 1, 64, 8), z["b"].reshape(n, 1, 64, 8), z["c"], 1)
    same = got.view(np.uint32) == z["d"][0].view(np.uint32)
    for f in ("rand", "group", "align"):
        sel = z["family"] == f
        print(f"{f}: {int(same[sel].sum())} / {int(same[sel].size)} elements bit-equal")
    return bool(same.all())


def run_chain_probe(a16, b16, c32, KB, tag):
    probe = os.path.join(HERE, "probes", "mfma_f16_chain_probe")
    if not os.path.exists(probe):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", probe, probe + ".hip"])
    fin, fout = f"/tmp/mfma_chain_{tag}.in", f"/tmp/mfma_chain_{tag}.out"
    with open(fin, "wb") as f:
        f.write(np.int32(c32.shape[0]).tobytes()); f.write(np.int32(KB).tobytes())
        f.write(np.ascontiguousarray(a16, np.uint16).tobytes()); f.write(np.ascontiguousarray(b16, np.uint16).tobytes())
        f.write(np.ascontiguousarray(c32, np.float32).tobytes())
    subprocess.check_call([probe, fin, fout], stdout=subprocess.DEVNULL)
    d = np.fromfile(fout, np.float32).reshape(c32.shape)
    os.remove(fin); os.remove(fout)
    return d


def soak_families(rng):
    """(name, KB, n, generator of raw f16 value arrays [n][KB][64][8] for A and B, C [n][64][16])"""
    def h16(x): return np.asarray(x, np.float32).astype(np.float16)
    fams = []
    def gemm_like(KB, n, wstd, xkind):
        shape = (n, KB, 64, 8)
        w = h16(rng.normal(0.0, wstd, size=shape))
        if xkind == "ln": x = h16(rng.normal(0.0, 1.0, size=shape))
        elif xkind == "gelu":
            g = rng.normal(0.0, 1.0, size=shape); x = h16(0.5 * g * (1 + np.tanh(0.79788456 * (g + 0.044715 * g ** 3))))
        else: x = h16(rng.normal(0.0, 0.3, size=shape) * (rng.random(shape) < 0.5))       # half zeros
        return w, x, np.zeros((n, 64, 16), np.float32)
    fams.append(("gemm K=768 w~0.02 x~LN", 48, 96, lambda: gemm_like(48, 96, 0.02, "ln")))
    fams.append(("gemm K=768 w~0.2 x~LN", 48, 48, lambda: gemm_like(48, 48, 0.2, "ln")))
    fams.append(("gemm K=3072 w~0.02 x~GELU", 192, 48, lambda: gemm_like(192, 48, 0.02, "gelu")))
    fams.append(("gemm K=1024 w~0.05 x sparse", 64, 32, lambda: gemm_like(64, 32, 0.05, "sparse")))
    def wide(KB, n, lo, hi, cexp):
        shape = (n, KB, 64, 8)
        def v():
            mant = 1.0 + rng.integers(0, 1024, size=shape) / 1024.0
            return h16(np.where(rng.integers(0, 2, size=shape) == 1, -1.0, 1.0) * mant * np.exp2(rng.integers(lo, hi + 1, size=shape)))
        c = np.zeros((n, 64, 16), np.float32)
        if cexp is not None:
            mant = 1.0 + rng.integers(0, 1 << 23, size=c.shape) / float(1 << 23)
            c = (np.where(rng.integers(0, 2, size=c.shape) == 1, -1.0, 1.0) * mant * np.exp2(rng.integers(cexp[0], cexp[1] + 1, size=c.shape))).astype(np.float32)
        return v(), v(), c
    fams.append(("wide exponents -7..7, C -20..20", 1, 200, lambda: wide(1, 200, -7, 7, (-20, 20))))
    fams.append(("wide exponents -7..7, C -60..-20 (tiny)", 1, 100, lambda: wide(1, 100, -7, 7, (-60, -20))))
    fams.append(("wide exponents -7..7, C 20..60 (huge)", 1, 100, lambda: wide(1, 100, -7, 7, (20, 60))))
    fams.append(("wide exponents, chains of 4", 4, 100, lambda: wide(4, 100, -6, 6, (-8, 8))))
    fams.append(("narrow exponents (cancellation), chains of 8", 8, 100, lambda: wide(8, 100, 0, 0, (0, 3))))
    def subnormal(KB, n):
        shape = (n, KB, 64, 8)
        def v():
            bits = rng.integers(0, 1 << 12, size=shape).astype(np.uint16)            # exponent fields 0..3: subnormals and the smallest normals
            bits |= (rng.integers(0, 2, size=shape).astype(np.uint16) << 15)
            return bits.view(np.float16)
        big = h16(np.exp2(rng.integers(0, 15, size=shape)) * (1.0 + rng.integers(0, 1024, size=shape) / 1024.0))
        a = np.where(rng.random(shape) < 0.5, v(), big)
        b = np.where(rng.random(shape) < 0.5, v(), big)
        c = (rng.normal(0, 1e-3, size=(n, 64, 16)) * (rng.random((n, 64, 16)) < 0.7)).astype(np.float32)
        return a.astype(np.float16), b.astype(np.float16), c
    fams.append(("f16 subnormal operands", 2, 150, lambda: subnormal(2, 150)))
    def tiny_acc(n):
        a, b, _ = wide(1, n, -12, -8, None)
        c = (np.where(rng.integers(0, 2, size=(n, 64, 16)) == 1, -1.0, 1.0) * np.exp2(rng.integers(-140, -100, size=(n, 64, 16)).astype(np.float64))).astype(np.float32)
        return a, b, c
    fams.append(("f32 subnormal / tiny accumulators", 1, 60, lambda: tiny_acc(60)))
    return fams


def cmd_soak(out_txt):
    rng = np.random.default_rng(20260924)
    lines = []
    all_ok = True
    for name, KB, n, gen in soak_families(rng):
        a, b, c = gen()
        a16 = a.view(np.uint16); b16 = b.view(np.uint16)
        dev = run_chain_probe(a16, b16, c, KB, "soak")
        cpu = emu_chain(a16, b16, c, KB)
        same = dev.view(np.uint32) == cpu.view(np.uint32)
        both_nan = np.isnan(dev) & np.isnan(cpu)
        ok = same | both_nan
        line = f"{name:50s} chains of {KB:3d} issues: {int(ok.sum())} / {ok.size} elements bit-equal"
        if not ok.all():
            all_ok = False
            idx = np.argwhere(~ok)[:5]
            for (t, lane, v) in idx:
                line += f"\n    trial {t} lane {lane} reg {v}: device {float(dev[t, lane, v]).hex()} cpu {float(cpu[t, lane, v]).hex()} C {float(c[t, lane, v]).hex()}"
        print(line); lines.append(line)
    lines.append("ALL BIT-EQUAL" if all_ok else "DIFFERENCES FOUND")
    with open(out_txt, "w") as f: f.write("\n".join(lines) + "\n")
    print(lines[-1])


if __name__ == "__main__":
    if len(sys.argv) < 3: sys.exit(__doc__)
    if sys.argv[1] == "gen": cmd_gen(sys.argv[2])
    elif sys.argv[1] == "fit": cmd_fit(sys.argv[2])
    elif sys.argv[1] == "check": sys.exit(0 if cmd_check(sys.argv[2]) else 1)
    elif sys.argv[1] == "soak": cmd_soak(sys.argv[2])
    else: sys.exit(__doc__)
