// fast_kernels.hip - the tolerance route of the many-row products and of the fine model's attention (BARK_HIP_FAST_GEMM=1).
//
// The canonical route (kernels.hip: gemm_kernel, attention_kernels.hip: attn_rows_kernel) reproduces ONE fixed f32 summation order on
// the f32 matrix cores (157 TFLOP/s peak); greedy ids are bit-identical to the oracle.  This file is the same operators on the f16
// matrix cores (v_mfma_f32_32x32x16_f16, 2.5 PFLOP/s peak): identical operands and rounding points for the products (R1: f16-rounded
// activations x f16 weights, exact f32 products, f32 accumulation) but the matrix core's own accumulation order, and a flash-style
// attention whose q / k / v / p are rounded to f16 (the canonical attention keeps them in f32).  Results agree with the canonical
// route to rounding noise (logits: stated tolerance 5e-3), NOT bit for bit - DESIGN.md section 3 explains why no ggml build agrees
// with another one more closely than that.  Never the route the parity tests check; tests/test_gpu_parity.py checks it against the
// canonical route with that tolerance.
//
//   gemm_f16_tile_kernel<TN, TM>   y[n][m] = epi(sum_k x[n][k] W[m][k] + b[m]): both operands K-contiguous.  256 threads = 2 x 2 waves,
//       tile TN x TM (128 x 128 or 64 x 64), K step 64.  Operand tiles are fetched as full 128-byte row pieces (8 lanes per row) into
//       registers, written to LDS with a 144-byte row stride (a ds_read_b128 of 16 consecutive rows then covers 16 disjoint groups of
//       four banks: conflict-free), two LDS buffers, ONE barrier per K step: tile t + 1 is in flight from memory while tile t is
//       multiplied, and is written behind the MFMAs into the buffer whose readers passed the previous barrier.
//   attn_flash_f16_kernel<KS>      non-causal attention over S keys (fine model), one workgroup per (head, 32 queries), KS waves that
//       split the keys.  The score tile is computed TRANSPOSED (S^T = K Q^T), so lane (half, q) holds 16 keys of ITS OWN query in its
//       accumulator: the online softmax needs one cross-lane exchange per key block (the two halves' maxima), and the exponentials are
//       already the B operand of the second product (O^T = V^T P^T) in registers - no LDS round trip, no transposition.  V is stored
//       transposed [d][key] by the QKV epilogue, with the keys of every group of 16 permuted to the order the accumulator holds them.
#include "device_utils.h"

#include <algorithm>

namespace barkhip {

typedef _Float16 half4v __attribute__((ext_vector_type(4)));
DEVINL uint4 ld_u4g(const half_t * p) { return *reinterpret_cast<const uint4 *>(p); }

constexpr int FT_LDK = 72;                       // halfs per staged row: 64 of the K step + 8 of padding (144 bytes)

// accumulator register r of lane (half, l31): x row nb + (r & 3) + 8 (r >> 2) (nb includes 4 half), weight row m
DEVINL void fast_tile_epilogue(const LinArgs & a, const floatx16 & acc, int tile_n0, int half, int m, int n_past) {
    const int nb = tile_n0 + 4 * half;
    const float bias = a.bias ? a.bias[m] : 0.0f;
    float v[16];
    #pragma unroll
    for (int r = 0; r < 16; r++) v[r] = a.bias ? acc[r] + bias : acc[r];
    switch (a.epi) {
        case EPI_RESID: {
            float old[16];
            #pragma unroll
            for (int r = 0; r < 16; r++) { const int n = nb + (r & 3) + 8 * (r >> 2); old[r] = n < a.N ? a.res[(size_t) n * a.M + m] : 0.0f; }
            #pragma unroll
            for (int r = 0; r < 16; r++) { const int n = nb + (r & 3) + 8 * (r >> 2); if (n < a.N) a.res[(size_t) n * a.M + m] = v[r] + old[r]; }   // cur + inpL (bark.cpp:1352,1388)
            break;
        }
        case EPI_GELU: {
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = nb + (r & 3) + 8 * (r >> 2);
                if (n < a.N) a.out_h[(size_t) n * a.M + m] = gelu_lut_apply(v[r], a.lut);
            }
            break;
        }
        case EPI_QKV: {
            const int E = a.E;
            const int mm = m < E ? m : m < 2 * E ? m - E : m - 2 * E;
            const int h = mm >> 6, d = mm & 63;
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = nb + (r & 3) + 8 * (r >> 2);
                if (n >= a.N) continue;
                const int zq = a.seq ? n / a.seq : 0;
                int pos = a.pos0 + n_past + (n - zq * a.seq);
                size_t zoff = (size_t) zq * a.kv_slot_stride;
                if (a.seqtab) {
                    const SeqTab t = a.seqtab[zq];
                    if (n - zq * a.seq >= t.len) continue;
                    pos = t.pos0 + (n - zq * a.seq); zoff = (size_t) t.slot * a.kv_slot_stride;
                }
                if (m < E) a.q[(size_t) n * E + m] = v[r];
                else if (m < 2 * E) a.kc[zoff + kc_index(h, d, pos, a.P)] = v[r];
                else { a.vc[zoff + vc_index(h, d, pos, a.P)] = v[r]; if (a.vt) a.vt[zoff + kc_index(h, d, pos, a.P)] = v[r]; }
            }
            break;
        }
        case EPI_QKV16: {
            // operands of attn_flash_f16_kernel: q (pre-scaled by log2(e) / sqrt(64): the scores arrive in log2 units) and k as f16 rows [n][E]; v transposed
            // [sequence][E][seq] with the keys of each group of 16 in the order 0-3, 8-11, 4-7, 12-15 - registers 0..7 of a lane are then
            // 8 consecutive positions (one 16-byte store) and exactly the 8 key slots the lane feeds to the second product
            const int E = a.E;
            if (m < 2 * E) {
                half_t * dst = m < E ? a.q16 + m : a.k16 + (m - E);
                const float sc = m < E ? 0.125f * 1.44269504088896340736f : 1.0f;
                #pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int n = nb + (r & 3) + 8 * (r >> 2);
                    if (n < a.N) dst[(size_t) n * E + (size_t) 0] = (half_t) (v[r] * sc);
                }
            } else {
                const int z = tile_n0 / a.seq, p0 = tile_n0 - z * a.seq;          // a 32-row MFMA tile never straddles two sequences (seq % 32 == 0)
                half_t * dst = a.vt16 + ((size_t) z * E + (m - 2 * E)) * a.seq + p0 + 8 * half;
                half8 lo, hi;
                #pragma unroll
                for (int r = 0; r < 8; r++) { lo[r] = (half_t) v[r]; hi[r] = (half_t) v[8 + r]; }
                if (tile_n0 < a.N)      *reinterpret_cast<half8 *>(dst) = lo;
                if (tile_n0 + 16 < a.N) *reinterpret_cast<half8 *>(dst + 16) = hi;
            }
            break;
        }
        default: {
            #pragma unroll
            for (int r = 0; r < 16; r++) { const int n = nb + (r & 3) + 8 * (r >> 2); if (n < a.N) a.out[(size_t) n * a.ld_out + m] = v[r]; }
            break;
        }
    }
}

// WM: waves along m (the waves form a 2 x WM grid: 4 or 8 per workgroup).  Eight waves (two workgroups per CU = four waves per SIMD) keep a SIMD's
// matrix core fed while some of them sit at the per-step barrier or wait for LDS - the PMC pass of the four-wave kernel showed 46 - 57 % of
// the wave cycles parked there.
template <int TN, int TM, int D, int WM>
__global__ __launch_bounds__(128 * WM, WM) void gemm_f16_tile_kernel(const LinArgs a, const int ncol, const int nrow, const int pw) {
    extern __shared__ __attribute__((aligned(16))) half_t lds_h[];
    half_t * As = lds_h;                                    // [2][TN][FT_LDK]
    half_t * Bs = lds_h + 2 * TN * FT_LDK;                  // [2][TM][FT_LDK]
    constexpr int NT = 128 * WM;                            // threads
    constexpr int IT = TN / 64, JT = TM / (32 * WM);        // 32 x 32 MFMA tiles per wave and dimension (waves as 2 x WM)
    constexpr int CA = TN * 8 / NT, CB = TM * 8 / NT;       // 16-byte chunks per thread and K step
    static_assert(JT >= 1 && CA >= 1 && CB >= 1, "tile too small for this many waves");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wn = w / WM, wm = w % WM;
    int trow, tcol;
    panel_tile(xcd_rank(blockIdx.x, ncol * nrow), nrow, ncol, pw, trow, tcol);      // XCD-aware: consecutive ranks = a compact block of tiles
    const int n0 = trow * TN, m0 = tcol * TM;
    const int K = a.K, nkt = K >> 6;
    const half_t * xsrc[CA]; const half_t * wsrc[CB];
    int adst[CA], bdst[CB];
    #pragma unroll
    for (int i = 0; i < CA; i++) {
        const int c = tid + NT * i, row = c >> 3, ch = c & 7;
        xsrc[i] = a.x_f16 + (size_t) min(n0 + row, a.N - 1) * K + ch * 8;
        adst[i] = row * FT_LDK + ch * 8;
    }
    #pragma unroll
    for (int i = 0; i < CB; i++) {
        const int c = tid + NT * i, row = c >> 3, ch = c & 7;
        wsrc[i] = a.W + (size_t) min(m0 + row, a.M - 1) * K + ch * 8;
        bdst[i] = row * FT_LDK + ch * 8;
    }
    // D register sets: tile t travels in set t % D, requested D - 1 K steps before it is written to LDS (a request to HBM takes ~2000
    // cycles, one K step of MFMAs 130 - 520); the loop is unrolled D times so that every set index is a compile-time constant
    uint4 ra[D][CA], rb[D][CB];
    floatx16 acc[IT][JT];
    #pragma unroll
    for (int i = 0; i < IT; i++)
        #pragma unroll
        for (int j = 0; j < JT; j++)
            #pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
#define FT_FETCH(SET, KT)                                                                                    \
    { _Pragma("unroll") for (int i = 0; i < CA; i++) ra[SET][i] = ld_u4g(xsrc[i] + ((KT) << 6));             \
      _Pragma("unroll") for (int i = 0; i < CB; i++) rb[SET][i] = ld_u4g(wsrc[i] + ((KT) << 6)); }
#define FT_STAGE(SET, BUF)                                                                                   \
    { _Pragma("unroll") for (int i = 0; i < CA; i++) *reinterpret_cast<uint4 *>(As + (BUF) * TN * FT_LDK + adst[i]) = ra[SET][i]; \
      _Pragma("unroll") for (int i = 0; i < CB; i++) *reinterpret_cast<uint4 *>(Bs + (BUF) * TM * FT_LDK + bdst[i]) = rb[SET][i]; }
#define FT_COMPUTE(KT)                                                                                       \
    { const half_t * Ab = As + ((KT) & 1) * TN * FT_LDK + aoff;                                              \
      const half_t * Bb = Bs + ((KT) & 1) * TM * FT_LDK + boff;                                              \
      _Pragma("unroll") for (int kk = 0; kk < 4; kk++) {                                                     \
          half8 af[IT], bf[JT];                                                                              \
          _Pragma("unroll") for (int i = 0; i < IT; i++) af[i] = *reinterpret_cast<const half8 *>(Ab + i * 32 * FT_LDK + kk * 16);      \
          _Pragma("unroll") for (int jj = 0; jj < JT; jj++) bf[jj] = *reinterpret_cast<const half8 *>(Bb + jj * 32 * FT_LDK + kk * 16);  \
          _Pragma("unroll") for (int i = 0; i < IT; i++)                                                     \
              _Pragma("unroll") for (int jj = 0; jj < JT; jj++) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[jj], acc[i][jj], 0, 0, 0); \
      } }
    const int aoff = (wn * (TN / 2) + l31) * FT_LDK + half * 8, boff = (wm * (TM / WM) + l31) * FT_LDK + half * 8;
    #pragma unroll
    for (int j = 0; j < D - 1; j++) if (j < nkt) FT_FETCH(j, j)
    FT_STAGE(0, 0)
    __syncthreads();
    // steady state: groups of D steps without a single branch between the requests and their use - hipcc places s_waitcnt per program
    // point, not per path: one `if` around a request makes every consumer wait for ALL outstanding loads, i.e. for the request of its
    // own step (measured: the pipeline depth then changes nothing).  The last 2 D - 2 steps run through a guarded tail.
    int kt = 0;
    for (; kt + 2 * D - 2 < nkt; kt += D) {
        #pragma unroll
        for (int j = 0; j < D; j++) {
            FT_FETCH((j + D - 1) % D, kt + j + D - 1)
            FT_COMPUTE(kt + j)
            FT_STAGE((j + 1) % D, (kt + j + 1) & 1)
            __syncthreads();
        }
    }
    // kt is a multiple of D here: step kt + j travels in set j % D
    #pragma unroll
    for (int j = 0; j < 2 * D - 1; j++) {
        if (kt + j < nkt) {                                     // uniform
            if (kt + j + D - 1 < nkt) FT_FETCH((j + D - 1) % D, kt + j + D - 1)
            FT_COMPUTE(kt + j)
            if (kt + j + 1 < nkt) FT_STAGE((j + 1) % D, (kt + j + 1) & 1)
            __syncthreads();
        }
    }
#undef FT_COMPUTE
#undef FT_FETCH
#undef FT_STAGE
    const int n_past = (a.epi == EPI_QKV && a.st) ? a.st->n_past : 0;
    #pragma unroll
    for (int j = 0; j < JT; j++) {
        const int m = m0 + wm * (TM / WM) + j * 32 + l31;
        if (m >= a.M) continue;
        #pragma unroll
        for (int i = 0; i < IT; i++) fast_tile_epilogue(a, acc[i][j], n0 + wn * (TN / 2) + i * 32, half, m, n_past);
    }
}

template <int TN, int TM, int D, int WM>
static void launch_tile(hipStream_t s, const LinArgs & a) {
    const size_t lds = (size_t) 2 * (TN + TM) * FT_LDK * sizeof(half_t);
    const int ncol = (a.M + TM - 1) / TM, nrow = (a.N + TN - 1) / TN;
    hipLaunchKernelGGL((gemm_f16_tile_kernel<TN, TM, D, WM>), dim3(ncol * nrow), dim3(128 * WM), lds, s, a, ncol, nrow, xcd_panel_width(ncol * nrow, ncol));
}

void launch_linear_fast(hipStream_t s, const LinArgs & a) {
    if (!a.x_f16 || !a.W || (a.K & 63) != 0 || a.N < 1) kernel_fail("bark-hip: the f16 tile product takes f16 rows and f16 weights, K %% 64 == 0");
    if (a.epi == EPI_QKV16 && (!a.q16 || !a.k16 || !a.vt16 || a.seq <= 0 || (a.seq & 31) || a.N % a.seq)) kernel_fail("bark-hip: QKV16 epilogue needs whole sequences of a multiple of 32 rows");
    // three register sets (a tile is requested two K steps ahead): 1.34 / 1.12 / 1.12 ms per fine pass at D = 2 / 3 / 4 (profiles/r03_fine_ab_fast.txt)
    const long tiles128 = (long) ((a.M + 127) / 128) * ((a.N + 127) / 128);
    // 128 x 128 tiles: eight waves (2 x 4), four per SIMD with two workgroups per CU - 1.37 -> 1.01 ms per fine pass, 0.67 -> 0.50 ms per window
    // with eight side by side against the four-wave form (profiles/r03_fine_ab_fast.txt); 64 x 64 tiles (fewer than 128 large tiles): four waves
    if (tiles128 >= 128) launch_tile<128, 128, 3, 4>(s, a); else launch_tile<64, 64, 3, 2>(s, a);
}

// ------------------------------------------------------------------------------------------------
// flash-style attention on the f16 matrix cores (see the file header).  Lane (half, l31) of a wave:
//   first product   S^T[key][q] = sum_d K[key][d] Q[q][d]:  A = K row key0 + l31, B = Q row q0 + l31, both the 8 d at 16 s + 8 half;
//                   accumulator register r = key (r & 3) + 8 (r >> 2) + 4 half of the block, query l31
//   second product  O^T[d][q] = sum_key V^T[d][key] P^T[key][q]:  B = the lane's own 8 exponentials of key step ks (registers 8 ks ..
//                   8 ks + 7), A = V^T row d = 32 t + l31 at positions key0 + 16 ks + 8 half .. + 7 (the permuted order of the epilogue)
// ------------------------------------------------------------------------------------------------
template <int KS, int NB>
__global__ __launch_bounds__(64 * KS) void attn_flash_f16_kernel(const AttnFlashArgs a) {
    __shared__ float comb[KS > 1 ? (KS - 1) * 34 * 64 : 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    // XCD-aware: the S / 32 query tiles of one (sequence, head) hold consecutive ranks, so its K and V^T (256 KB) stay in ONE L2
    const int QT = a.S >> 5, rank = xcd_rank(blockIdx.x, QT * a.H * a.Z);
    const int q0 = (rank % QT) * 32, h = (rank / QT) % a.H, z = rank / (QT * a.H);
    const int E = a.E, S = a.S;
    const half_t * Qp = a.q16 + ((size_t) z * S + q0 + l31) * E + h * 64 + half * 8;
    const half_t * Kp = a.k16 + ((size_t) z * S + l31) * E + h * 64 + half * 8;
    const half_t * Vp = a.vt16 + ((size_t) z * E + h * 64 + l31) * S + half * 8;
    half8 qf[4];
    #pragma unroll
    for (int s = 0; s < 4; s++) qf[s] = ld_half8(Qp + s * 16);
    const int per = S / KS, kbeg = w * per, kend = kbeg + per;
    floatx16 o[2];
    #pragma unroll
    for (int t = 0; t < 2; t++)
        #pragma unroll
        for (int r = 0; r < 16; r++) o[t][r] = 0.0f;
    // running maximum in log2 units; it is raised lazily: while the block maximum stays below mrun + 8 the exponentials are formed
    // against the old maximum (at most 2^8: harmless in f32 and in the f16 operand) and the 32 accumulators are not rescaled
    float mrun = -INFINITY, lrun = 0.0f;
    constexpr float LAZY = 8.0f;
    // NB register sets: key block b travels in set b % NB and is requested NB - 1 blocks ahead (the operands were written by the
    // previous kernel on other XCDs: every request goes to the memory side)
    half8 kr[NB][4], vr[NB][2][2];
#define FA_LOAD(SET, KEY0)                                                                                   \
    { _Pragma("unroll") for (int s = 0; s < 4; s++) kr[SET][s] = ld_half8(Kp + (size_t) (KEY0) * E + s * 16); \
      _Pragma("unroll") for (int t = 0; t < 2; t++)                                                          \
          _Pragma("unroll") for (int ks = 0; ks < 2; ks++) vr[SET][t][ks] = ld_half8(Vp + (size_t) t * 32 * S + (KEY0) + ks * 16); }
    auto block = [&](const half8 (&kk_)[4], const half8 (&vv_)[2][2]) {
        floatx16 sc;
        #pragma unroll
        for (int r = 0; r < 16; r++) sc[r] = 0.0f;
        #pragma unroll
        for (int s = 0; s < 4; s++) sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kk_[s], qf[s], sc, 0, 0, 0);
        float t2[16];                                                   // scores in log2 units (q carries log2(e) / 8)
        #pragma unroll
        for (int r = 0; r < 16; r++) t2[r] = sc[r];
        float mx = fmaxf(fmaxf(fmaxf(t2[0], t2[1]), fmaxf(t2[2], t2[3])), fmaxf(fmaxf(t2[4], t2[5]), fmaxf(t2[6], t2[7])));
        mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(t2[8], t2[9]), fmaxf(t2[10], t2[11])), fmaxf(fmaxf(t2[12], t2[13]), fmaxf(t2[14], t2[15]))));
        // the two halves of a query exchange their maxima only when one of them outgrows the running maximum (mrun is the same in both halves
        // at all times): the cross-lane round trip is off the common path
        if (__builtin_amdgcn_ballot_w64(mx > mrun + LAZY) != 0) {       // some query of the wave needs a new maximum (rare after the first blocks)
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(mrun, mx);
            const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
            lrun *= alpha;
            #pragma unroll
            for (int t = 0; t < 2; t++)
                #pragma unroll
                for (int r = 0; r < 16; r++) o[t][r] *= alpha;
            mrun = mnew;
        }
        float p[16];
        #pragma unroll
        for (int r = 0; r < 16; r++) p[r] = __builtin_amdgcn_exp2f(t2[r] - mrun);
        lrun += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])) + (((p[8] + p[9]) + (p[10] + p[11])) + ((p[12] + p[13]) + (p[14] + p[15])));
        half8 pf[2];
        #pragma unroll
        for (int ks = 0; ks < 2; ks++)
            #pragma unroll
            for (int e = 0; e < 8; e++) pf[ks][e] = (half_t) p[ks * 8 + e];
        #pragma unroll
        for (int t = 0; t < 2; t++)
            #pragma unroll
            for (int ks = 0; ks < 2; ks++) o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vv_[t][ks], pf[ks], o[t], 0, 0, 0);
    };
    #pragma unroll
    for (int j = 0; j < NB - 1; j++) if (kbeg + 32 * j < kend) FA_LOAD(j, kbeg + 32 * j)
    // steady state without branches between the requests and their use (see gemm_f16_tile_kernel), guarded tail of 2 NB - 2 blocks
    int key0 = kbeg;
    for (; key0 + 32 * (2 * NB - 2) < kend; key0 += 32 * NB) {
        #pragma unroll
        for (int j = 0; j < NB; j++) {
            FA_LOAD((j + NB - 1) % NB, key0 + 32 * (j + NB - 1))
            block(kr[j], vr[j]);
        }
    }
    #pragma unroll
    for (int j = 0; j < 2 * NB - 1; j++) {
        if (key0 + 32 * j < kend) {                             // uniform
            if (key0 + 32 * (j + NB - 1) < kend) FA_LOAD((j + NB - 1) % NB, key0 + 32 * (j + NB - 1))
            block(kr[j % NB], vr[j % NB]);
        }
    }
#undef FA_LOAD
    lrun = lrun + __shfl_xor(lrun, 32, 64);                 // the two halves hold disjoint keys of the same query
    if constexpr (KS > 1) {
        // the waves' partial results meet in LDS: (m, l, O) of waves 1 .. KS-1, merged by wave 0
        if (w > 0) {
            float * c = comb + (size_t) (w - 1) * 34 * 64 + lane;
            c[0] = mrun; c[64] = lrun;
            #pragma unroll
            for (int t = 0; t < 2; t++)
                #pragma unroll
                for (int r = 0; r < 16; r++) c[(2 + t * 16 + r) * 64] = o[t][r];
        }
        __syncthreads();
        if (w > 0) return;
        #pragma unroll
        for (int ww = 1; ww < KS; ww++) {
            const float * c = comb + (size_t) (ww - 1) * 34 * 64 + lane;
            const float m2 = c[0], l2 = c[64];
            const float mnew = fmaxf(mrun, m2);
            const float a1 = __builtin_amdgcn_exp2f(mrun - mnew), a2 = __builtin_amdgcn_exp2f(m2 - mnew);
            lrun = lrun * a1 + l2 * a2;
            #pragma unroll
            for (int t = 0; t < 2; t++)
                #pragma unroll
                for (int r = 0; r < 16; r++) o[t][r] = o[t][r] * a1 + c[(2 + t * 16 + r) * 64] * a2;
            mrun = mnew;
        }
    }
    const float inv = 1.0f / lrun;
    half_t * out = a.att + ((size_t) z * S + q0 + l31) * a.ld_att + h * 64 + 4 * half;
    #pragma unroll
    for (int t = 0; t < 2; t++)
        #pragma unroll
        for (int g = 0; g < 4; g++) {
            half4v v4;
            #pragma unroll
            for (int e = 0; e < 4; e++) v4[e] = (half_t) (o[t][4 * g + e] * inv);
            *reinterpret_cast<half4v *>(out + t * 32 + 8 * g) = v4;
        }
}

void launch_attn_flash(hipStream_t s, const AttnFlashArgs & a) {
    if (a.S <= 0 || (a.S & 127) || a.Z < 1 || (a.E & 7) || (a.ld_att & 3)) kernel_fail("bark-hip: flash attention takes whole sequences of a multiple of 128 keys");
    // two waves per (window, head, 32 queries) split the keys, three register sets per wave: KS = 1 / 2 / 4 and NB = 2 / 3 / 4 measured within
    // 2 % of each other once the pipeline's steady state is branch-free (profiles/r03_fine_ab_fast.txt)
    hipLaunchKernelGGL((attn_flash_f16_kernel<2, 3>), dim3(a.S / 32 * a.H * a.Z), dim3(128), 0, s, a);
}

void init_fast_attributes() {
    const int lds = 2 * (128 + 128) * FT_LDK * (int) sizeof(half_t);
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16_tile_kernel<128, 128, 3, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

}  // namespace barkhip
