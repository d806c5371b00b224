// engine_codec.hip - EnCodec decode (encodec_decompress_audio call site, /root/reference/bark.cpp:2143-2167) for one utterance or
// for all utterances of a lock-step batch at once.  Architecture: HF modeling_encodec.py:316-347 (decoder stack), :236-249 (LSTM +
// skip), :252-282 (residual blocks), :381-448 (RVQ de-embedding); kernels in codec_kernels.hip, the LSTM input projection in kernels.hip.
#include "engine_internal.h"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>

using namespace barkhip;
using namespace barkhip::detail;

namespace barkhip {

// codes[b]: [n_q][T[b]] ids of utterance b.  Every activation is time-major [row][C] with the utterances' rows back to back (CodecBatch,
// kernels.h): one launch per operator for the whole batch, the T + 1 strictly sequential LSTM launches of an utterance shared by all.
// tap_stage >= 0 (one utterance only): *tap receives the activation after that stage (0 first conv, 1 LSTM + skip, 2..5 up-blocks).
std::vector<std::vector<float>> engine_codec_decode_many(bark_context * c, const std::vector<const int32_t *> & codes, int n_q, const std::vector<int> & T,
                                                         int tap_stage, std::vector<float> * tap) {
    HIP_OK(hipSetDevice(c->device));
    CodecModel & cm = c->codec;
    const int B = (int) T.size();
    if (B < 1 || B > 32 || codes.size() != T.size()) throw std::runtime_error("codec: 1..32 utterances per call");
    if (tap_stage >= 0 && B != 1) throw std::runtime_error("codec: activation taps take one utterance");
    if (n_q <= 0 || n_q > cm.n_q) throw std::runtime_error("codec: bad code matrix shape");
    std::vector<int> Tpre((size_t) B + 1, 0);
    int Tmax = 0;
    for (int b = 0; b < B; b++) {
        if (T[(size_t) b] <= 0 || T[(size_t) b] > 4096) throw std::runtime_error("codec: bad code matrix shape");
        for (size_t i = 0; i < (size_t) n_q * T[(size_t) b]; i++) if (codes[(size_t) b][i] < 0 || codes[(size_t) b][i] >= cm.hp.n_bins) throw std::runtime_error("codec: code out of range");
        Tpre[(size_t) b + 1] = Tpre[(size_t) b] + T[(size_t) b];
        Tmax = std::max(Tmax, T[(size_t) b]);
    }
    const int Ts = Tpre[(size_t) B];                              // frames of the whole batch
    hipStream_t s = c->stream;
    const int D = cm.D;
    // largest activation: rows x channels at every stage (time-major; the hidden channels of a residual block are half its width)
    size_t need = (size_t) std::max(cm.hp.hidden_dim, D) * Ts;
    { int ch = D; size_t tt = (size_t) Ts; for (auto & b : cm.blocks) { ch = b.up.cout; tt *= b.up.stride; need = std::max(need, (size_t) ch * tt); } }
    if (need > c->cbuf_elems) {
        for (auto & b : c->cbuf) b = dev_alloc<float>(c, need);
        for (auto & b : c->cbuf_hh) b = dev_alloc<half_t>(c, need);
        c->cbuf_elems = need;
    }
    if ((size_t) Ts > c->c_T) {
        c->c_gi = dev_alloc<float>(c, (size_t) Ts * 4 * D);
        c->c_hseq_h = dev_alloc<half_t>(c, (size_t) Ts * D);
        c->c_xt_h = dev_alloc<half_t>(c, (size_t) Ts * D);
        c->c_hseq2_h = dev_alloc<half_t>(c, (size_t) Ts * D);
        c->c_T = (size_t) Ts;
    }
    if (!c->c_cell) { c->c_cell = dev_alloc<float>(c, (size_t) 32 * D); c->c_cell2 = dev_alloc<float>(c, (size_t) 32 * D); c->d_codec_T = dev_alloc<int>(c, 80); }
    if ((size_t) n_q * Ts > c->d_codes_elems) { c->d_codes = dev_alloc<int32_t>(c, (size_t) n_q * Ts); c->d_codes_elems = (size_t) n_q * Ts; }
    for (int b = 0; b < B; b++)
        HIP_OK(hipMemcpyAsync(c->d_codes + (size_t) n_q * Tpre[(size_t) b], codes[(size_t) b], (size_t) n_q * T[(size_t) b] * 4, hipMemcpyHostToDevice, s));
    // device copies of the frame counts and their prefix sums: [0, 32) T, [40, 73) Tpre
    {
        int hdr[80] = {};
        for (int b = 0; b < B; b++) hdr[b] = T[(size_t) b];
        for (int b = 0; b <= B; b++) hdr[40 + b] = Tpre[(size_t) b];
        HIP_OK(hipMemcpyAsync(c->d_codec_T, hdr, sizeof(hdr), hipMemcpyHostToDevice, s));
        HIP_OK(hipStreamSynchronize(s));                        // hdr is a stack object
    }
    CodecBatch cb; cb.T = c->d_codec_T; cb.Tpre = c->d_codec_T + 40; cb.B = B;
    // time-major buffers: three f32 (A / Bf / R) and three f16 (H0 / H1 / H2)
    float * A = c->cbuf[0], * Bf = c->cbuf[1], * R = c->cbuf[2];
    half_t * H0 = c->cbuf_hh[0], * H1 = c->cbuf_hh[1], * H2 = c->cbuf_hh[2];

    // one convolution over time-major rows: xh [rows_in][cin] f16 -> any of y (f32), yh_raw, yh_elu (f16: what the next operators consume)
    auto conv_args = [&](const half_t * wm, const float * w32, const float * bias, int cin, int cout, int K, int stride, const half_t * xh, int tm_in) {
        ConvTmArgs a;
        a.W = wm; a.w32 = w32; a.bias = bias; a.cin = cin; a.cout = cout; a.cout32 = (cout + 31) & ~31; a.K = K;
        a.convT = stride > 0 ? 1 : 0; a.nphase = stride > 0 ? stride : 1;
        a.kd = (stride > 0 ? 2 : K) * cin; a.kd16 = (a.kd + 15) & ~15;
        a.xh = xh; a.rows_in = Ts * tm_in; a.tm_in = tm_in; a.cb = cb;
        return a;
    };
    auto conv = [&](const CodecModel::Conv & cv, const half_t * xh, int tm, const float * add, float * y, half_t * yh_raw, half_t * yh_elu) {
        ConvTmArgs a = conv_args(cv.wm, cv.w32, cv.b, cv.cin, cv.cout, cv.k, 0, xh, tm);
        a.add = add; a.y = y; a.yh_raw = yh_raw; a.yh_elu = yh_elu;
        launch_conv_tm(s, a);
    };
    // RVQ de-embedding, first conv
    launch_rvq_gather(s, cm.codebooks, cm.hp.n_bins, cm.hp.hidden_dim, c->d_codes, n_q, Tmax, Ts, A, cb);
    launch_act_round(s, A, (size_t) cm.hp.hidden_dim * Ts, 0, H0);
    conv(cm.init, H0, 1, nullptr, Bf, c->c_xt_h, nullptr);      // Bf = x [row][D]; its f16 image is the LSTM's input
    // 2-layer LSTM + skip (modeling_encodec.py:236-249), both layers as a wave front: launch i = layer 1 at step i + layer 2 at step
    // i - 1 (its input projection formed in the same kernel): T + 1 strictly sequential launches instead of 2 T, for all utterances at
    // once.  64 of them are captured once as a hipGraph whose nodes take their launch index from a device counter, so one graph
    // serves every T (and is re-captured only when the batch size or a buffer changes).
    {
        LinArgs g;
        g.W = cm.lstm[0].w_ih; g.M = 4 * D; g.K = D; g.N = Ts; g.x_f16 = c->c_xt_h; g.epi = EPI_LOGITS; g.out = c->c_gi; g.ld_out = 4 * D;
        launch_linear(s, g);
    }
    LstmPairArgs a;
    a.gi1 = c->c_gi; a.w_hh1 = cm.lstm[0].w_hh; a.b_ih1 = cm.lstm[0].b_ih; a.b_hh1 = cm.lstm[0].b_hh; a.c1 = c->c_cell; a.h1 = c->c_hseq_h;
    a.w_ih2 = cm.lstm[1].w_ih; a.w_hh2 = cm.lstm[1].w_hh; a.b_ih2 = cm.lstm[1].b_ih; a.b_hh2 = cm.lstm[1].b_hh; a.c2 = c->c_cell2; a.h2 = c->c_hseq2_h;
    a.out2 = R; a.T = Tmax; a.D = D; a.cb = cb;
    if (!c->use_graph) {
        for (int i = 0; i <= Tmax; i++) { a.t = i; launch_lstm_pair_step(s, a); }
    } else {
        constexpr int kBlock = 64;
        auto & slot = c->lstm_graph;
        if (slot.exec && (slot.out != R || slot.gi != c->c_gi || slot.B != B)) { (void) hipGraphExecDestroy(slot.exec); slot.exec = nullptr; }
        if (!slot.exec) {
            hipGraph_t graph = nullptr;
            HIP_OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            try {
                a.t_base = c->d_lstm_t;
                for (int i = 0; i < kBlock; i++) { a.t = i; launch_lstm_pair_step(s, a); }
                launch_add_int(s, c->d_lstm_t, kBlock);
            } catch (...) { hipGraph_t g2 = nullptr; (void) hipStreamEndCapture(s, &g2); if (g2) (void) hipGraphDestroy(g2); throw; }
            HIP_OK(hipStreamEndCapture(s, &graph));
            HIP_OK(hipGraphInstantiate(&slot.exec, graph, nullptr, nullptr, 0));
            (void) hipGraphDestroy(graph);
            slot.out = R; slot.gi = c->c_gi; slot.B = B;
        }
        const int hdr[2] = {0, Tmax};
        HIP_OK(hipMemcpyAsync(c->d_lstm_t, hdr, sizeof(hdr), hipMemcpyHostToDevice, s));
        HIP_OK(hipStreamSynchronize(s));                         // hdr is a stack object
        for (int i0 = 0; i0 <= Tmax; i0 += kBlock) HIP_OK(hipGraphLaunch(slot.exec, s));
    }
    // parity taps are handed out channel-major [C][T'] (one utterance)
    auto grab = [&](int stage, const float * buf, int C, size_t rows) {
        if (tap_stage != stage || !tap) return;
        std::vector<float> tm((size_t) C * rows);
        HIP_OK(hipMemcpyAsync(tm.data(), buf, tm.size() * 4, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        tap->resize(tm.size());
        for (size_t r = 0; r < rows; r++) for (int ch = 0; ch < C; ch++) (*tap)[(size_t) ch * rows + r] = tm[r * (size_t) C + ch];
    };
    grab(0, Bf, D, (size_t) Ts);
    // skip connection + the four upsampling blocks + final conv: ~25 launches, replayed from a hipGraph captured per list of frame
    // counts (the buffers are the context's own, so a graph stays valid until they are re-allocated for a longer input)
    int tmul = 1;
    float * pcm_dev = nullptr;
    auto tail = [&](bool taps) {
        launch_add(s, R, Bf, (size_t) D * Ts, A);                    // y + x
        if (taps) grab(1, A, D, (size_t) Ts);
        launch_act_round(s, A, (size_t) D * Ts, 1, H0);             // H0 = f16(ELU(x))
        tmul = 1;
        for (int b = 0; b < 4; b++) {
            const CodecModel::Block & bl = cm.blocks[b];
            // upsampling of f16(ELU(x)) (H0): the next operators consume only the f16 images of its result - as it is (shortcut input, H1) and
            // after the ELU (conv1 input, H2)
            ConvTmArgs u = conv_args(bl.up.wm, bl.up.w32, bl.up.b, bl.up.cin, bl.up.cout, bl.up.k, bl.up.stride, H0, tmul);
            u.yh_raw = H1; u.yh_elu = H2;
            launch_conv_tm(s, u);
            tmul *= bl.up.stride;
            // residual block: shortcut(x) + conv2(elu(conv1(elu(x))))   (modeling_encodec.py:252-282)
            conv(bl.c1, H2, tmul, nullptr, nullptr, nullptr, H0);           // H0 = f16(ELU(conv1(...)))
            conv(bl.c2, H0, tmul, nullptr, R, nullptr, nullptr);            // R = r (f32: it is added, not multiplied)
            conv(bl.sc, H1, tmul, R, taps ? A : nullptr, nullptr, H0);      // shortcut(x) + r; H0 = f16(ELU(..)): the next block's / the last conv's input
            if (taps) grab(2 + b, A, bl.up.cout, (size_t) Ts * tmul);
        }
        conv(cm.fin, H0, tmul, nullptr, Bf, nullptr, nullptr);
        pcm_dev = Bf;
    };
    if (c->use_graph && tap_stage < 0) {
        auto & cg = c->codec_graph;
        if (cg.exec && (cg.T != T || cg.buf != A)) { (void) hipGraphExecDestroy(cg.exec); cg.exec = nullptr; }
        if (!cg.exec) {
            hipGraph_t graph = nullptr;
            HIP_OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            try { tail(false); }
            catch (...) { hipGraph_t g2 = nullptr; (void) hipStreamEndCapture(s, &g2); if (g2) (void) hipGraphDestroy(g2); throw; }
            HIP_OK(hipStreamEndCapture(s, &graph));
            HIP_OK(hipGraphInstantiate(&cg.exec, graph, nullptr, nullptr, 0));
            (void) hipGraphDestroy(graph);
            cg.T = T; cg.buf = A; cg.out = pcm_dev; cg.tmul = tmul;
        }
        HIP_OK(hipGraphLaunch(cg.exec, s));
        c->stats.graph_replays++;
        pcm_dev = cg.out; tmul = cg.tmul;
    } else {
        tail(true);
    }
    std::vector<float> all((size_t) Ts * tmul);                  // the final conv has one output channel: utterance b = samples [tmul Tpre[b], tmul Tpre[b + 1])
    HIP_OK(hipMemcpyAsync(all.data(), pcm_dev, all.size() * 4, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    std::vector<std::vector<float>> pcm((size_t) B);
    for (int b = 0; b < B; b++) pcm[(size_t) b].assign(all.begin() + (size_t) tmul * Tpre[(size_t) b], all.begin() + (size_t) tmul * Tpre[(size_t) b + 1]);
    return pcm;
}

std::vector<float> engine_codec_decode(bark_context * c, const int32_t * codes, int n_q, int T, int tap_stage, std::vector<float> * tap) {
    return std::move(engine_codec_decode_many(c, {codes}, n_q, {T}, tap_stage, tap)[0]);
}

}  // namespace barkhip
