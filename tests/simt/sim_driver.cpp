// Host-side execution of the opt-in few-slot lock-step kernels next to the kernels they must equal, work-item for work-item (see hip/hip_runtime.h in this
// directory and tests/test_simt_emulation.py, which builds this file against patched copies of the kernel sources).  Test infrastructure only.
#include "kernels_sim.hip"
#include "attention_kernels_sim.hip"
#include "fast_kernels_sim.hip"

namespace barkhip {
// the other kernel files' entry points kernels.hip refers to: never reached here
void init_quant_attributes() {}
void launch_linear_q(hipStream_t, const LinArgs &) { kernel_fail("sim: quantised products are not emulated"); }
void launch_linear_w32(hipStream_t, const LinArgs &) { kernel_fail("sim: f32 products are not emulated"); }
}

using namespace barkhip;

namespace {
template <int NBLK> void qkv_slots(const LinArgs & a, int B) {
    const int n_main = (a.M + 15) / 16, n_q = a.E / 16;
    sim::launch(dim3(n_main + 2 * n_q, B), 256, [&] { gemv_ln_slots_ps_kernel<NBLK, true, true>(a.W, a.x_f32, a.ln_g, a.ln_b, a.kc, a.st, a.M, a.E, 512, a); });
}
template <int NBLK> void qkv_single(const LinArgs & a) {
    const int n_main = (a.M + 15) / 16, n_q = a.E / 16;
    sim::launch(dim3(n_main + 2 * n_q), 256, [&] { gemv_ln_wg_kernel<NBLK, true, true>(a.W, a.x_f32, a.ln_g, a.ln_b, a.kc, a.st, a.M, 0, a.E, 512, a); });
}
template <int NBLK> void fc_slots(const LinArgs & a, int B) {
    sim::launch(dim3((a.M + 15) / 16, B), 256, [&] { gemv_ln_slots_ps_kernel<NBLK, true, false>(a.W, a.x_f32, a.ln_g, a.ln_b, a.kc, a.st, a.M, a.E, 0, a); });
}
template <int NBLK> void fc_single(const LinArgs & a) {
    sim::launch(dim3((a.M + 15) / 16), 256, [&] { gemv_ln_wg_kernel<NBLK, true, false>(a.W, a.x_f32, a.ln_g, a.ln_b, a.kc, a.st, a.M, 0, a.E, 0, a); });
}
template <int NBLK> void proj_slots(const LinArgs & a, int B) {
    sim::launch(dim3((a.M + 3) / 4, B), 64, [&] { gemv_slots_kernel<NBLK>(a.W, a.x_f16, a.M, a); });
}
template <int NBLK> void proj_single(const LinArgs & a) {
    sim::launch(dim3((a.M + 3) / 4), 64, [&] { gemv_kernel<NBLK>(a.W, a.x_f16, a.M, 0, a); });
}
template <int NBLK> void qkv_slots16(const LinArgs & a, int B) {
    sim::launch(dim3((a.M + 15) / 16, (B + 15) / 16), 256, [&] { gemm_slots16_kernel<NBLK, true>(a.W, a.x_f16, a.M, 0, a); });
}
#define BY_NBLK(F, K, ...) switch ((K) >> 7) { case 1: F<1>(__VA_ARGS__); break; case 2: F<2>(__VA_ARGS__); break; case 4: F<4>(__VA_ARGS__); break; case 8: F<8>(__VA_ARGS__); break; default: return -1; }
}  // namespace

extern "C" {

// QKV of one layer for B slots.  route 0: the experimental kernel with a slot dimension; route 1: the single-utterance kernel, slot after slot.
// x [B][E] f32, W [3E][E] f16, kc / vc [B][stride] f32 (K [H][16][P][4], V [H][P][64]), q [B][E], ps [B][H][4][P], st [B]
int sim_qkv(int route, const void * W, const float * x, const float * ln_g, const float * ln_b, const float * bias, float * kc, float * vc, float * q, float * ps,
            StepState * st, int E, int B, long stride) {
    LinArgs a;
    a.W = (const half_t *) W; a.M = 3 * E; a.K = E; a.N = 1; a.ln_g = ln_g; a.ln_b = ln_b; a.bias = bias; a.epi = EPI_QKV; a.E = E; a.P = 1024; a.pos0 = 0;
    if (route == 0 || route == 2) {
        a.batched = 1; a.nbatch = B; a.kv_slot_stride = (size_t) stride; a.x_f32 = x; a.q = q; a.kc = kc; a.vc = vc; a.st = st;
        if (route == 0) { a.ps = ps; BY_NBLK(qkv_slots, E, a, B) }
        else switch (E >> 7) {                                                                 // the default lock step: matrix cores, LayerNorm fused (n_embd % 256 == 0)
            case 2: qkv_slots16<2>(a, B); break; case 4: qkv_slots16<4>(a, B); break; case 8: qkv_slots16<8>(a, B); break; default: return -1; }
    } else {
        for (int b = 0; b < B; b++) {
            a.x_f32 = x + (size_t) b * E; a.q = q + (size_t) b * E; a.kc = kc + (size_t) b * stride; a.vc = vc + (size_t) b * stride; a.st = st + b;
            a.ps = ps + (size_t) b * (E / 64) * 4 * 1024; a.ng = 4;
            BY_NBLK(qkv_single, E, a)
        }
    }
    return 0;
}

// decode attention of B slots at their own context lengths.  route 0: attn_fused_ps_kernel on the partial scores; route 1: attn_fused_kernel (the default lock step)
int sim_attention(int route, int vs, const float * q, const float * kc, const float * vc, const float * ps, const StepState * st, void * att, int H, int B, long stride) {
    AttnDecodeArgs a;
    a.q = q; a.kc = kc; a.vc = vc; a.H = H; a.P = 1024; a.st = st; a.att = (half_t *) att; a.nbatch = B; a.kv_slot_stride = (size_t) stride;
    if (route == 0) {
        a.ps = ps;
        if (vs == 2) sim::launch(dim3(H, B, 2), 256, [&] { attn_fused_ps_kernel<2>(a); }); else sim::launch(dim3(H, B), 256, [&] { attn_fused_ps_kernel<1>(a); });
    } else {
        if (vs == 2) sim::launch(dim3(H, B, 2), 256, [&] { attn_fused_kernel<2>(a); }); else sim::launch(dim3(H, B), 256, [&] { attn_fused_kernel<1>(a); });
    }
    return 0;
}

// The single-utterance decode step's own pair: gemv_ln_wg_kernel<PS> (with the fixed-address copy of the appended K row and the K-layout copy of V) and
// attn_ps_kernel on its partial scores.  One sequence: x [E], caches of one layer, vt = V in the K layout [H][16][P][4]; att [E] f16
int sim_decode_attention(const void * W, const float * x, const float * ln_g, const float * ln_b, const float * bias, float * kc, float * vc, float * vt, float * q, float * ps,
                         float * knew, StepState * st, void * att, int E, int ng) {
    LinArgs a;
    a.W = (const half_t *) W; a.M = 3 * E; a.K = E; a.N = 1; a.ln_g = ln_g; a.ln_b = ln_b; a.bias = bias; a.epi = EPI_QKV; a.E = E; a.P = 1024; a.pos0 = 0;
    a.x_f32 = x; a.q = q; a.kc = kc; a.vc = vc; a.vt = vt; a.st = st; a.ps = ps; a.knew = knew; a.ng = ng;
    // copies as launch_gemv_n sizes them (kernels.hip)
    const int n_main = (a.M + 15) / 16, n_q = E / 16, keys = 256 * std::max(1, std::min(ng, 4));
    const int fit = std::max(1, std::min(2, (256 - n_main) / n_q));
    const int n_copy = std::max((keys + 511) / 512, std::min(fit, keys / 256));
    const int kpc = ((keys + n_copy - 1) / n_copy + 127) / 128 * 128;
    switch (E >> 7) {
        case 1: sim::launch(dim3(n_main + n_copy * n_q), 256, [&] { gemv_ln_wg_kernel<1, true, true>(a.W, a.x_f32, a.ln_g, a.ln_b, a.kc, a.st, a.M, 0, a.E, kpc, a); }); break;
        case 2: sim::launch(dim3(n_main + n_copy * n_q), 256, [&] { gemv_ln_wg_kernel<2, true, true>(a.W, a.x_f32, a.ln_g, a.ln_b, a.kc, a.st, a.M, 0, a.E, kpc, a); }); break;
        default: return -1;
    }
    AttnDecodeArgs at;
    at.q = q; at.kc = kc; at.vc = vc; at.H = E / 64; at.P = 1024; at.st = st; at.att = (half_t *) att; at.ps = ps; at.knew = knew; at.ng = ng; at.vt = vt;
    sim::launch(dim3(8 * 16 * ((at.H + 7) / 8)), 1024, [&] { attn_ps_kernel(at.ps, at.vt, at.st, at.knew, at.q, at.H, std::max(1, std::min(ng, 4)), at); });
    return st->fault ? -2 : 0;
}

// LayerNorm + FC + GELU table for B slots: out [B][M] f16.  route 0: the per-slot kernel without copies; route 1: the single-utterance kernel per slot
int sim_fc(int route, const void * W, const float * x, const float * ln_g, const float * ln_b, const float * bias, const uint16_t * lut, void * out, int E, int M, int B) {
    LinArgs a;
    a.W = (const half_t *) W; a.M = M; a.K = E; a.N = 1; a.ln_g = ln_g; a.ln_b = ln_b; a.bias = bias; a.epi = EPI_GELU; a.lut = lut; a.E = E;
    if (route == 0) { a.batched = 1; a.nbatch = B; a.x_f32 = x; a.out_h = (half_t *) out; BY_NBLK(fc_slots, E, a, B) }
    else for (int b = 0; b < B; b++) { a.x_f32 = x + (size_t) b * E; a.out_h = (half_t *) out + (size_t) b * M; BY_NBLK(fc_single, E, a) }
    return 0;
}

// out-projection + residual for B slots: res [B][M] f32 updated in place, xh [B][K] f16.  route 0: gemv_slots_kernel; route 1: gemv_kernel per slot
int sim_proj(int route, const void * W, const void * xh, const float * bias, float * res, int K, int M, int B) {
    LinArgs a;
    a.W = (const half_t *) W; a.M = M; a.K = K; a.N = 1; a.bias = bias; a.epi = EPI_RESID;
    if (route == 0) { a.batched = 1; a.nbatch = B; a.x_f16 = (const half_t *) xh; a.res = res; BY_NBLK(proj_slots, K, a, B) }
    else for (int b = 0; b < B; b++) { a.x_f16 = (const half_t *) xh + (size_t) b * K; a.res = res + (size_t) b * M; BY_NBLK(proj_single, K, a) }
    return 0;
}

// N rows through the prefill product (gemm_kernel: persistent workgroups, C1 chains on v_mfma_f32_32x32x2_f32): out [N][M] f32 = x [N][K] f16 x W [M][K] f16 + bias
// wide: the 32-row x 96-column tile shape (round 6) instead of 64 x 64
int sim_gemm(const void * W, const void * xh, const float * bias, float * out, int N, int K, int M, int grid, int wide) {
    LinArgs a;
    a.W = (const half_t *) W; a.M = M; a.K = K; a.N = N; a.x_f16 = (const half_t *) xh; a.bias = bias; a.epi = EPI_LOGITS; a.out = out; a.ld_out = M;
    if (wide) {
        const int ncol = (M + 95) / 96, nrow = (N + 31) / 32;
        sim::launch(dim3(std::min(grid, ncol * nrow)), 512, [&] { gemm_kernel<32, 96>(a, ncol, nrow, xcd_panel_width(ncol * nrow, ncol)); });
    } else {
        const int ncol = (M + 63) / 64, nrow = (N + 63) / 64;
        sim::launch(dim3(std::min(grid, ncol * nrow)), 512, [&] { gemm_kernel<64, 64>(a, ncol, nrow, xcd_panel_width(ncol * nrow, ncol)); });
    }
    return 0;
}

// N rows through the fine model's product (gemm_f16_tile_kernel: canonical order C1m, the f16 matrix cores' own accumulation): out [N][M] f32.  tile: 128 or 64
int sim_gemm_f16(const void * W, const void * xh, const float * bias, float * out, int N, int K, int M, int tile) {
    LinArgs a;
    a.W = (const half_t *) W; a.M = M; a.K = K; a.N = N; a.x_f16 = (const half_t *) xh; a.bias = bias; a.epi = EPI_LOGITS; a.out = out; a.ld_out = M;
    const int ncol = (M + tile - 1) / tile, nrow = (N + tile - 1) / tile, pw = xcd_panel_width(ncol * nrow, ncol);
    if (tile == 128) sim::launch(dim3(ncol * nrow), 512, [&] { gemm_f16_tile_kernel<128, 128, 3, 4>(a, ncol, nrow, pw); });
    else if (tile == 64) sim::launch(dim3(ncol * nrow), 256, [&] { gemm_f16_tile_kernel<64, 64, 3, 2>(a, ncol, nrow, pw); });
    else return -1;
    return 0;
}

// N queries at positions n_past .. against the cache of one sequence (prefill: causal; fine model: whole windows, N == 1024, n_past == 0, not causal).
// kernel 0: attn_rows_kernel (scores in an LDS tile), 1: attn_window_kernel<causal> (scores in registers).  q [N][E] f32, att [N][E] f16
int sim_attention_rows(int kernel, const float * q, const float * kc, const float * vc, void * att, int H, int N, int n_past, int causal) {
    AttnPrefillArgs a;
    a.q = q; a.ldq = H * 64; a.kc = kc; a.vc = vc; a.H = H; a.P = 1024; a.N = N; a.n_past = n_past; a.causal = causal; a.att = (half_t *) att; a.ld_att = H * 64;
    if (kernel == 1) {
        if (causal) {                                          // round 6: the register-resident kernel with the causal mask (any N, continues a cache at n_past)
            if (n_past + N > 1024) return -1;
            sim::launch(dim3((N + 31) / 32 * H), 512, [&] { attn_window_kernel<true>(a); });
        } else {
            if (N != 1024 || n_past != 0) return -1;
            sim::launch(dim3(32 * H), 512, [&] { attn_window_kernel<false>(a); });
        }
    } else sim::launch(dim3((N + 31) / 32 * H), 512, [&] { attn_rows_kernel(a); });
    return 0;
}

int sim_state_size() { return (int) sizeof(StepState); }

}  // extern "C"
