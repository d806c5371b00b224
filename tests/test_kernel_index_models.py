"""Index models of the MFMA lock-step kernels (CPU, exact arithmetic): the lane -> (row, slot, chain, element) arithmetic of
gemm_slots4_kernel and gemm_slots16_kernel (bark.cpp_amd/csrc/kernels.hip) restated line by line in Python, with the matrix instruction
replaced by what the device probes established (tools/probes/mfma4x4_probe.hip, mfma16x16x4_probe.hip: one fmaf per block element / an
ascending-k fmaf chain, A index in the result register, B index in the lane).  Every output must equal the C1 dot product of
tests/test_canon_orders.py bit for bit.  This does not run the HIP code - the -m gpu route checks do that - it checks that the
decomposition the kernels implement IS C1, for shapes that exercise clamped rows, a second slot tile and more than one chunk round."""
import numpy as np

from test_canon_orders import add32, c1_dot, fma32


def _halves(row_f16, q):
    """the chunk's eight f16 values as four little-endian 32-bit words (e0 | e1 << 16, ...), as a 16-byte load sees them"""
    h = row_f16[8 * q:8 * q + 8].view(np.uint16).astype(np.uint32)
    return [int(h[0] | (h[1] << 16)), int(h[2] | (h[3] << 16)), int(h[4] | (h[5] << 16)), int(h[6] | (h[7] << 16))]


def _half_to_f32(bits16):
    return np.float32(np.array([bits16], np.uint16).view(np.float16)[0])


def _problem(M, B, K, seed):
    rng = np.random.default_rng(seed)
    W = (rng.standard_normal((M, K)) * 0.05).astype(np.float16)
    X = rng.standard_normal((B, K)).astype(np.float16)
    return W, X


def test_slots16_decomposition_is_c1():
    M, B, K = 13, 18, 256                     # one row tile with clamped rows, two slot tiles (the second with 2 live slots), 2 chunk rounds
    W, X = _problem(M, B, K, 16)
    nblk = K // 128
    for by in range(2):
        m0, s0 = 0, 16 * by
        lds = np.zeros((8, 16, 16), np.float32)
        for w in range(8):
            acc = [np.zeros((16, 16), np.float32) for _ in range(2)]          # D[row][slot] of chains 2w and 2w+1
            for i in range(nblk):
                for c in range(2):
                    q = 16 * i + 2 * w + c                                    # wrow + (i << 7) + (c << 3) with the (2 w) << 3 base
                    for second in range(2):                                   # the chunk's two MFMAs
                        A = np.zeros((16, 4), np.float32); Bm = np.zeros((4, 16), np.float32)
                        for lane in range(64):
                            g, r = lane >> 4, lane & 15
                            hi, sh = g >= 2, 16 * (g & 1)
                            ww = _halves(W[min(m0 + r, M - 1)], q); xw = _halves(X[min(s0 + r, B - 1)], q)
                            wsel = (ww[3] if hi else ww[2]) if second else (ww[1] if hi else ww[0])
                            xsel = (xw[3] if hi else xw[2]) if second else (xw[1] if hi else xw[0])
                            A[r][g] = _half_to_f32((wsel >> sh) & 0xFFFF)     # A: lane holds row lane % 16 of k = lane / 16
                            Bm[g][r] = _half_to_f32((xsel >> sh) & 0xFFFF)    # B: lane holds column lane % 16 of k = lane / 16
                        for row in range(16):
                            for col in range(16):
                                v = acc[c][row][col]
                                for k in range(4):                            # ascending-k fmaf chain (device probe)
                                    v = fma32(A[row][k], Bm[k][col], v)
                                acc[c][row][col] = v
            for row in range(16):
                for col in range(16):
                    lds[w][row][col] = add32(acc[0][row][col], acc[1][row][col])
        for tid in range(256):
            emm, en = tid & 15, (tid >> 4) & 15
            if s0 + en >= B or m0 + emm >= M:
                continue
            p = [lds[qq][emm][en] for qq in range(8)]
            v = add32(add32(add32(p[0], p[1]), add32(p[2], p[3])), add32(add32(p[4], p[5]), add32(p[6], p[7])))
            want = c1_dot(W[m0 + emm].astype(np.float32), X[s0 + en].astype(np.float32))
            assert v == want, (by, emm, en, v, want)


def test_slots4_decomposition_is_c1():
    M, B, K = 6, 11, 256                      # two row quads (the second with 2 live rows), slots 8..10 in the second wave
    W, X = _problem(M, B, K, 4)
    nblk = K // 128
    for bx in range(2):
        m0 = 4 * bx
        for w in range(4):
            if 8 * w >= B:
                continue
            acc = np.zeros((2, 16, 4, 4), np.float32)                         # [slot group][block = chain][row v][slot j]
            for i in range(nblk):
                for e in range(8):
                    for gidx in range(2):
                        for b in range(16):
                            a = [np.float32(W[min(m0 + r, M - 1)][8 * (16 * i + b) + e]) for r in range(4)]
                            xs = [np.float32(X[min(8 * w + 4 * gidx + r, B - 1)][8 * (16 * i + b) + e]) for r in range(4)]
                            for v in range(4):                                # A index in the result register, B index in the lane
                                for j in range(4):
                                    acc[gidx][b][v][j] = fma32(a[v], xs[j], acc[gidx][b][v][j])
            for gidx in range(2):
                for v in range(4):
                    for j in range(4):
                        t = [acc[gidx][b][v][j] for b in range(16)]
                        for st in (1, 2, 4, 8):                               # lane xor 4, 8, 16, 32 = chain xor 1, 2, 4, 8
                            t = [add32(t[b], t[b ^ st]) for b in range(16)]
                        n, m = 8 * w + 4 * gidx + j, m0 + v
                        if n < B and m < M:
                            want = c1_dot(W[m].astype(np.float32), X[n].astype(np.float32))
                            assert t[0] == want and all(x == t[0] for x in t), (bx, w, gidx, v, j)


def _xcd_rank(b, n):
    """device_utils.h: xcd_rank - workgroup b (XCD b % 8) -> a rank such that one XCD holds consecutive ranks"""
    q, r, x, i = n >> 3, n & 7, b & 7, b >> 3
    return (x * (q + 1) if x < r else r * (q + 1) + (x - r) * q) + i


def _panel_tile(l, nrow, ncol, pw):
    """device_utils.h: panel_tile - rank -> (row tile, column tile), column panels of width pw"""
    per_panel = nrow * pw
    p, rem = divmod(l, per_panel)
    c0 = p * pw
    w = min(pw, ncol - c0)
    row = rem // w
    return row, c0 + (rem - row * w)


def _xcd_panel_width(n_tiles, ncol):
    """kernels.hip: xcd_panel_width"""
    pw = 1
    while pw * pw < (n_tiles + 7) // 8:
        pw += 1
    return max(1, min(pw, ncol))


def test_xcd_aware_tile_order_visits_every_tile_once():
    """The renumbering the many-row products and the prefill / fine attentions apply to blockIdx.x (any grid size, also when it is not a
    multiple of 8, ragged last column panel): every output tile exactly once, and the workgroups of one XCD (b % 8) cover a compact
    block (a run of consecutive ranks: at most two column panels, share / pw rows of each) instead of tiles from all over the output."""
    for nrow, ncol in [(1, 1), (1, 7), (3, 5), (16, 36), (8, 18), (8, 24), (16, 12), (16, 17), (5, 36), (64, 18), (11, 9), (2, 48)]:
        n = nrow * ncol
        pw = _xcd_panel_width(n, ncol)
        ranks = sorted(_xcd_rank(b, n) for b in range(n))
        assert ranks == list(range(n)), (nrow, ncol)
        seen = set()
        per_xcd = {}
        for b in range(n):
            t = _panel_tile(_xcd_rank(b, n), nrow, ncol, pw)
            assert 0 <= t[0] < nrow and 0 <= t[1] < ncol and t not in seen, (nrow, ncol, b, t)
            seen.add(t)
            per_xcd.setdefault(b % 8, []).append(t)
        assert len(seen) == n
        if n >= 64:
            for tiles in per_xcd.values():
                rows, cols = {t[0] for t in tiles}, {t[1] for t in tiles}
                assert len(cols) <= 2 * pw, (nrow, ncol, pw, len(rows), len(cols))
                if (nrow, ncol) in ((16, 36), (8, 18), (8, 24), (16, 12), (64, 18)):          # the fine model's products
                    assert len(rows) + len(cols) <= (nrow + ncol) * 2 // 3, (nrow, ncol, pw, len(rows), len(cols))


def test_attention_rank_order_keeps_a_head_on_one_xcd():
    """attn_rows_kernel / attn_flash_f16_kernel: rank -> (sequence, head, query tile); the 32 query tiles of one (sequence, head) are
    consecutive ranks, i.e. they run on at most QT / (ranks per XCD) + 2 XCDs - two for the fine model's 32 tiles x 12 heads."""
    for QT, H, Z in [(32, 12, 1), (32, 12, 8), (9, 2, 3), (32, 16, 5)]:
        n = QT * H * Z
        owner = {}
        for b in range(n):
            r = _xcd_rank(b, n)
            key = (r // (QT * H), (r // QT) % H)
            owner.setdefault(key, set()).add(b % 8)
            assert 0 <= r % QT < QT
        share = max(1, n // 8)                                             # consecutive ranks one XCD holds
        assert len(owner) == H * Z and all(len(v) <= QT // share + 2 for v in owner.values())


def _window_rank(b, QT, H, Z):
    """attn_window_kernel (round 6): blockIdx.x -> rank.  xcd_rank, then - when the grid is a multiple of 8 - an XCD's run of C consecutive ranks is dealt
    over NS (window, head) units at a time: j -> (j % NS) * (C / NS) + j / NS."""
    total = QT * H * Z
    rank = _xcd_rank(b, total)
    if total % 8 == 0:
        C = total // 8
        base, j = (rank // C) * C, rank % C
        NS = 4 if (C % 4 == 0 and C // 4 >= QT) else 3 if (C % 3 == 0 and C // 3 >= QT) else 2 if (C % 2 == 0 and C // 2 >= QT) else 1
        rank = base + (j % NS) * (C // NS) + j // NS
    return rank


def test_window_attention_deals_an_xcds_run_over_several_heads_and_spreads_the_k_walks():
    """attn_window_kernel's workgroup mapping of round 6, restated: (1) still a bijection onto (window, head, query tile), every XCD keeping its own run of
    ranks (K / V of a unit stay in one L2); (2) the 32 workgroups an XCD runs side by side (consecutive b on that XCD) belong to NS different units once the
    run is long enough - one window: 1, two windows: 3, eight: 4; (3) the K walk of query tile q starts at key tile (q >> 2) & 3 and 16-d block q & 3, so the
    tiles of one unit that run side by side start from all sixteen (tile, block) positions, every (tile, block) is visited exactly once per workgroup, and
    the register sets come back in key order by the two conditional swaps."""
    for QT, H, Z, want_units in [(32, 12, 1, 1), (32, 12, 2, 3), (32, 12, 8, 4), (32, 16, 1, 2), (32, 16, 5, 4)]:
        n = QT * H * Z
        ranks = [_window_rank(b, QT, H, Z) for b in range(n)]
        assert sorted(ranks) == list(range(n))                                            # (1) bijection
        C = n // 8
        for x in range(8):
            mine = [ranks[b] for b in range(x, n, 8)]                                     # XCD x, in the order its workgroups start
            assert sorted(mine) == list(range(x * C, (x + 1) * C)), (QT, H, Z, x)         # its own run of consecutive ranks, nothing else
            first = mine[:32]                                                             # resident together (one workgroup per CU, 32 CUs per XCD)
            units = {r // QT for r in first}
            assert len(units) >= want_units, (QT, H, Z, x, len(units))                    # (2)
    # (3) rotation: block order b ^ rot_b, tile order t ^ rot_t, sets restored by swaps on bit 0 then bit 1
    starts = set()
    for q in range(32):
        rot_b, rot_t = q & 3, (q >> 2) & 3
        visited = [((n >> 2) ^ rot_t, (n & 3) ^ rot_b) for n in range(16)]                # load_blk(n): (key tile, 16-d block)
        assert sorted(visited) == [(t, b) for t in range(4) for b in range(4)]
        starts.add(visited[0])
        sets = [t ^ rot_t for t in range(4)]                                              # register set t holds key tile t ^ rot_t after the scores
        if rot_t & 1:
            sets[0], sets[1] = sets[1], sets[0]; sets[2], sets[3] = sets[3], sets[2]
        if rot_t & 2:
            sets[0], sets[2] = sets[2], sets[0]; sets[1], sets[3] = sets[3], sets[1]
        assert sets == [0, 1, 2, 3], (q, sets)
        for t in range(4):                                                                # accumulator b of a tile holds block b ^ rot_b: (a0 + a1) + (a2 + a3) is
            acc = [b ^ rot_b for b in range(4)]                                           # (c0 + c1) + (c2 + c3) up to commuted operands
            assert {frozenset(acc[0:2]), frozenset(acc[2:4])} == {frozenset((0, 1)), frozenset((2, 3))}
    assert len(starts) == 16


# ---- lock-step route for few slots (default up to 8 live slots, BARK_HIP_FEW_SLOTS): gemv_ln_slots_ps_kernel (kernels.hip) -> attn_fused_ps_kernel (attention_kernels.hip) ----
def _mul32(a, b):
    return np.float32(np.float32(a) * np.float32(b))


def _score_block(kq4, qb):
    """score_block_f4 (device_utils.h): one fmaf chain over the block's four d-quads x four components"""
    acc = np.float32(0.0)
    for i in range(4):
        for comp in range(4):
            acc = fma32(kq4[i][comp], qb[4 * i + comp], acc)
    return acc


def test_slot_partial_scores_reach_the_attention_as_the_c2_score():
    """The address arithmetic of the two kernels restated thread by thread on a small shape (2 slots with different contexts, 2 heads, block_size
    1024, keys beyond 512 so that both copies of a q block work): what copy workgroup (rep, q block) thread t writes to ps, and what thread t of the
    attention workgroup (head, slot) reads back and combines, must be C2's score ((c0 + c1) + (c2 + c3)) * 0.125 of key j for every cached key -
    and the key the step appended must be scored from its K row at position ctx - 1."""
    rng = np.random.default_rng(5)
    H, P, B = 2, 1024, 2
    E = 64 * H
    n_q, n_main, kpc, n_copy = E // 16, 3 * E // 16, 512, 2
    ctxs = [600, 37]                                           # n_past + 1 per slot
    stride = H * 16 * P * 4                                    # floats per slot of one layer's K cache [H][16][P][4]
    kc = rng.standard_normal(B * stride).astype(np.float32)
    q = rng.standard_normal((B, E)).astype(np.float32)
    ps = np.full(B * H * 4 * P, np.nan, np.float32)
    # --- producer: gemv_ln_slots_ps_kernel<PS = true>, the copies (blockIdx.x >= n_main), grid (n_main + n_copy n_q, B) ---
    for slot in range(B):
        n_past = ctxs[slot] - 1
        for bx in range(n_main, n_main + n_copy * n_q):
            rep, wg = (bx - n_main) // n_q, (bx - n_main) % n_q
            if rep * kpc >= n_past:
                continue                                       # the early exit
            m0 = wg * 16
            hq, blk = m0 >> 6, (m0 >> 4) & 3
            qb = [q[slot][m0 + i] for i in range(16)]          # qs[wave * 4 + rg] = q row m0 + 4 wave + rg
            base_f4 = slot * stride // 4 + (hq * 16 + 4 * blk) * 1024 + rep * kpc          # buf_rsrc base in float4 units
            for tid in range(256):
                for half in range(2):
                    if half == 1 and not (tid + 256 < kpc):
                        continue
                    j = rep * kpc + tid + 256 * half
                    if j >= n_past:
                        continue
                    kq = []
                    for i in range(4):                         # buf_ld_f4(kr, tid * 16 (+ 4096), i * 16384): byte offsets
                        byte = base_f4 * 16 + tid * 16 + 4096 * half + i * 16384
                        kq.append(kc[byte // 4: byte // 4 + 4])
                    ps[slot * (E >> 6) * 4 * P + (hq * 4 + blk) * P + j] = _score_block(kq, qb)
    # --- consumer: attn_fused_ps_kernel, workgroup (h, slot), thread tid owns keys tid + 256 g ---
    for slot in range(B):
        ctx = ctxs[slot]
        for h in range(H):
            psl = (slot * H + h) * 4 * P
            got = {}
            for tid in range(256):
                for g in range(4):
                    if not (g == 0 or ctx > 256 * g):
                        continue
                    if tid + 256 * g < ctx - 1:
                        p4 = [ps[psl + b * P + tid + 256 * g] for b in range(4)]
                        got[tid + 256 * g] = _mul32(add32(add32(p4[0], p4[1]), add32(p4[2], p4[3])), 0.125)
            # the appended key: lanes 0..3 of wave 3 form the blocks from the K row at ctx - 1
            cb = []
            for b in range(4):
                kq = [kc[slot * stride + ((h * 16 + 4 * b + i) * P + (ctx - 1)) * 4: slot * stride + ((h * 16 + 4 * b + i) * P + (ctx - 1)) * 4 + 4] for i in range(4)]
                cb.append(_score_block(kq, [q[slot][h * 64 + 16 * b + i] for i in range(16)]))
            got[ctx - 1] = _mul32(add32(add32(cb[0], cb[1]), add32(cb[2], cb[3])), 0.125)
            # reference: C2 on the K layout [H][16][P][4] (kc_index: ((h 16 + d / 4) P + pos) 4 + d % 4)
            assert sorted(got) == list(range(ctx))
            for j in range(ctx):
                c = []
                for b in range(4):
                    acc = np.float32(0.0)
                    for d in range(16 * b, 16 * b + 16):
                        acc = fma32(kc[slot * stride + ((h * 16 + d // 4) * P + j) * 4 + d % 4], q[slot][h * 64 + d], acc)
                    c.append(acc)
                ref = _mul32(add32(add32(c[0], c[1]), add32(c[2], c[3])), 0.125)
                assert got[j].tobytes() == ref.tobytes(), (slot, h, j)
