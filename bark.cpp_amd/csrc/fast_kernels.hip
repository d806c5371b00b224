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
                const int pos = a.pos0 + n_past + n;
                if (m < E) a.q[(size_t) n * E + m] = v[r];
                else if (m < 2 * E) a.kc[kc_index(h, d, pos, a.P)] = v[r];
                else { a.vc[vc_index(h, d, pos, a.P)] = v[r]; if (a.vt) a.vt[kc_index(h, d, pos, a.P)] = v[r]; }
            }
            break;
        }
        case EPI_QKV16: {
            // operands of attn_flash_f16_kernel: q (pre-scaled by 1/sqrt(64): exact in f16) and k as f16 rows [n][E]; v transposed
            // [sequence][E][seq] with the keys of each group of 16 in the order 0-3, 8-11, 4-7, 12-15 - registers 0..7 of a lane are then
            // 8 consecutive positions (one 16-byte store) and exactly the 8 key slots the lane feeds to the second product
            const int E = a.E;
            if (m < 2 * E) {
                half_t * dst = m < E ? a.q16 + m : a.k16 + (m - E);
                const float sc = m < E ? 0.125f : 1.0f;
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

template <int TN, int TM>
__global__ __launch_bounds__(256) void gemm_f16_tile_kernel(const LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) half_t lds_h[];
    half_t * As = lds_h;                                    // [2][TN][FT_LDK]
    half_t * Bs = lds_h + 2 * TN * FT_LDK;                  // [2][TM][FT_LDK]
    constexpr int IT = TN / 64, JT = TM / 64;               // 32 x 32 MFMA tiles per wave and dimension (waves as 2 x 2)
    constexpr int CA = TN / 32, CB = TM / 32;               // 16-byte chunks per thread and K step
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wn = w >> 1, wm = w & 1;
    const int n0 = blockIdx.y * TN, m0 = blockIdx.x * TM;
    const int K = a.K, nkt = K >> 6;
    const half_t * xsrc[CA]; const half_t * wsrc[CB];
    int adst[CA], bdst[CB];
    #pragma unroll
    for (int i = 0; i < CA; i++) {
        const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
        xsrc[i] = a.x_f16 + (size_t) min(n0 + row, a.N - 1) * K + ch * 8;
        adst[i] = row * FT_LDK + ch * 8;
    }
    #pragma unroll
    for (int i = 0; i < CB; i++) {
        const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
        wsrc[i] = a.W + (size_t) min(m0 + row, a.M - 1) * K + ch * 8;
        bdst[i] = row * FT_LDK + ch * 8;
    }
    uint4 ra[CA], rb[CB];
    auto fetch = [&](int kt) {
        #pragma unroll
        for (int i = 0; i < CA; i++) ra[i] = ld_u4g(xsrc[i] + (kt << 6));
        #pragma unroll
        for (int i = 0; i < CB; i++) rb[i] = ld_u4g(wsrc[i] + (kt << 6));
    };
    auto stage = [&](int buf) {
        #pragma unroll
        for (int i = 0; i < CA; i++) *reinterpret_cast<uint4 *>(As + buf * TN * FT_LDK + adst[i]) = ra[i];
        #pragma unroll
        for (int i = 0; i < CB; i++) *reinterpret_cast<uint4 *>(Bs + buf * TM * FT_LDK + bdst[i]) = rb[i];
    };
    floatx16 acc[IT][JT];
    #pragma unroll
    for (int i = 0; i < IT; i++)
        #pragma unroll
        for (int j = 0; j < JT; j++)
            #pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
    fetch(0);
    stage(0);
    __syncthreads();
    const int aoff = (wn * (TN / 2) + l31) * FT_LDK + half * 8, boff = (wm * (TM / 2) + l31) * FT_LDK + half * 8;
    for (int kt = 0; kt < nkt; kt++) {
        if (kt + 1 < nkt) fetch(kt + 1);
        const half_t * Ab = As + (kt & 1) * TN * FT_LDK + aoff;
        const half_t * Bb = Bs + (kt & 1) * TM * FT_LDK + boff;
        #pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            half8 af[IT], bf[JT];
            #pragma unroll
            for (int i = 0; i < IT; i++) af[i] = *reinterpret_cast<const half8 *>(Ab + i * 32 * FT_LDK + kk * 16);
            #pragma unroll
            for (int j = 0; j < JT; j++) bf[j] = *reinterpret_cast<const half8 *>(Bb + j * 32 * FT_LDK + kk * 16);
            #pragma unroll
            for (int i = 0; i < IT; i++)
                #pragma unroll
                for (int j = 0; j < JT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nkt) stage((kt + 1) & 1);
        __syncthreads();
    }
    const int n_past = (a.epi == EPI_QKV && a.st) ? a.st->n_past : 0;
    #pragma unroll
    for (int j = 0; j < JT; j++) {
        const int m = m0 + wm * (TM / 2) + j * 32 + l31;
        if (m >= a.M) continue;
        #pragma unroll
        for (int i = 0; i < IT; i++) fast_tile_epilogue(a, acc[i][j], n0 + wn * (TN / 2) + i * 32, half, m, n_past);
    }
}

void launch_linear_fast(hipStream_t s, const LinArgs & a) {
    if (!a.x_f16 || !a.W || (a.K & 63) != 0 || a.N < 1) kernel_fail("bark-hip: the f16 tile product takes f16 rows and f16 weights, K %% 64 == 0");
    if (a.epi == EPI_QKV16 && (!a.q16 || !a.k16 || !a.vt16 || a.seq <= 0 || (a.seq & 31) || a.N % a.seq)) kernel_fail("bark-hip: QKV16 epilogue needs whole sequences of a multiple of 32 rows");
    const long tiles128 = (long) ((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (tiles128 >= 128) {
        const size_t lds = (size_t) 2 * (128 + 128) * FT_LDK * sizeof(half_t);
        hipLaunchKernelGGL((gemm_f16_tile_kernel<128, 128>), dim3((a.M + 127) / 128, (a.N + 127) / 128), dim3(256), lds, s, a);
    } else {
        const size_t lds = (size_t) 2 * (64 + 64) * FT_LDK * sizeof(half_t);
        hipLaunchKernelGGL((gemm_f16_tile_kernel<64, 64>), dim3((a.M + 63) / 64, (a.N + 63) / 64), dim3(256), lds, s, a);
    }
}

// ------------------------------------------------------------------------------------------------
// flash-style attention on the f16 matrix cores (see the file header).  Lane (half, l31) of a wave:
//   first product   S^T[key][q] = sum_d K[key][d] Q[q][d]:  A = K row key0 + l31, B = Q row q0 + l31, both the 8 d at 16 s + 8 half;
//                   accumulator register r = key (r & 3) + 8 (r >> 2) + 4 half of the block, query l31
//   second product  O^T[d][q] = sum_key V^T[d][key] P^T[key][q]:  B = the lane's own 8 exponentials of key step ks (registers 8 ks ..
//                   8 ks + 7), A = V^T row d = 32 t + l31 at positions key0 + 16 ks + 8 half .. + 7 (the permuted order of the epilogue)
// ------------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(64 * KS) void attn_flash_f16_kernel(const AttnFlashArgs a) {
    __shared__ float comb[KS > 1 ? (KS - 1) * 34 * 64 : 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int q0 = blockIdx.x * 32, h = blockIdx.y, z = blockIdx.z;
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
    float mrun = -INFINITY, lrun = 0.0f;
    constexpr float L2E = 1.44269504088896340736f;
    half8 kf[4], vf[2][2], kn[4], vn[2][2];
    auto load_block = [&](half8 (&kk)[4], half8 (&vv)[2][2], int key0) {
        #pragma unroll
        for (int s = 0; s < 4; s++) kk[s] = ld_half8(Kp + (size_t) key0 * E + s * 16);
        #pragma unroll
        for (int t = 0; t < 2; t++)
            #pragma unroll
            for (int ks = 0; ks < 2; ks++) vv[t][ks] = ld_half8(Vp + (size_t) t * 32 * S + key0 + ks * 16);
    };
    auto block = [&](const half8 (&kk)[4], const half8 (&vv)[2][2]) {
        floatx16 sc;
        #pragma unroll
        for (int r = 0; r < 16; r++) sc[r] = 0.0f;
        #pragma unroll
        for (int s = 0; s < 4; s++) sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kk[s], qf[s], sc, 0, 0, 0);
        float mx = sc[0];
        #pragma unroll
        for (int r = 1; r < 16; r++) mx = fmaxf(mx, sc[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * L2E);
        const float moff = -mnew * L2E;
        float p[16], ls = 0.0f;
        #pragma unroll
        for (int r = 0; r < 16; r++) { p[r] = __builtin_amdgcn_exp2f(fmaf(sc[r], L2E, moff)); ls += p[r]; }
        lrun = lrun * alpha + ls;
        mrun = mnew;
        #pragma unroll
        for (int t = 0; t < 2; t++)
            #pragma unroll
            for (int r = 0; r < 16; r++) o[t][r] *= alpha;
        half8 pf[2];
        #pragma unroll
        for (int ks = 0; ks < 2; ks++)
            #pragma unroll
            for (int e = 0; e < 8; e++) pf[ks][e] = (half_t) p[ks * 8 + e];
        #pragma unroll
        for (int t = 0; t < 2; t++)
            #pragma unroll
            for (int ks = 0; ks < 2; ks++) o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vv[t][ks], pf[ks], o[t], 0, 0, 0);
    };
    load_block(kf, vf, kbeg);
    for (int key0 = kbeg; key0 < kend; key0 += 64) {
        const bool more = key0 + 32 < kend;
        if (more) load_block(kn, vn, key0 + 32);
        block(kf, vf);
        if (more) {
            if (key0 + 64 < kend) load_block(kf, vf, key0 + 64);
            block(kn, vn);
        }
    }
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
            const float a1 = __builtin_amdgcn_exp2f((mrun - mnew) * L2E), a2 = __builtin_amdgcn_exp2f((m2 - mnew) * L2E);
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
    static const int ks = getenv("BARK_HIP_FLASH_KS") ? atoi(getenv("BARK_HIP_FLASH_KS")) : 2;      // waves per (head, 32-query) tile: A/B on the device
    const dim3 grid(a.S / 32, a.H, a.Z);
    if (ks >= 4)      hipLaunchKernelGGL((attn_flash_f16_kernel<4>), grid, dim3(256), 0, s, a);
    else if (ks == 2) hipLaunchKernelGGL((attn_flash_f16_kernel<2>), grid, dim3(128), 0, s, a);
    else              hipLaunchKernelGGL((attn_flash_f16_kernel<1>), grid, dim3(64), 0, s, a);
}

void init_fast_attributes() {
    (void) hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16_tile_kernel<128, 128>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               2 * (128 + 128) * FT_LDK * (int) sizeof(half_t));
}

}  // namespace barkhip
