/*
 * bark.h - drop-in C API of the MI355X-native Bark engine.
 *
 * This header is the ABI contract of the reference's public interface
 * (/root/reference/bark.h:37-240): same enums, same `bark_context_params` field order (the
 * struct is passed BY VALUE, so field order and types are ABI), same entry points with C
 * linkage.  A caller written against the reference header (examples/main/main.cpp,
 * examples/server/server.cpp, ...) compiles and links against libbark.so unchanged.
 * The implementation behind it is a from-scratch HIP engine for gfx950; nothing of ggml exists
 * here - the three ggml/encodec headers included below are thin shims that only provide the few
 * names callers of this header use (enum ggml_ftype, ggml_time_us, ...).
 *
 * Engine extensions (device selection, stage-level entry points, batches) live in
 * bark_mi355x.h and never change anything in this file.
 */
/*
 * The declarations below restate the public interface of PABannier/bark.cpp's bark.h, which is distributed under the
 * ISC licence; its notice is reproduced here as that licence requires:
 *
 *   Copyright 2024 Pierre-Antoine Bannier
 *
 *   Permission to use, copy, modify, and/or distribute this software for any purpose with or without fee is hereby
 *   granted, provided that the above copyright notice and this permission notice appear in all copies.
 *
 *   THE SOFTWARE IS PROVIDED "AS IS" AND THE AUTHOR DISCLAIMS ALL WARRANTIES WITH REGARD TO THIS SOFTWARE INCLUDING ALL
 *   IMPLIED WARRANTIES OF MERCHANTABILITY AND FITNESS. IN NO EVENT SHALL THE AUTHOR BE LIABLE FOR ANY SPECIAL, DIRECT,
 *   INDIRECT, OR CONSEQUENTIAL DAMAGES OR ANY DAMAGES WHATSOEVER RESULTING FROM LOSS OF USE, DATA OR PROFITS, WHETHER IN
 *   AN ACTION OF CONTRACT, NEGLIGENCE OR OTHER TORTIOUS ACTION, ARISING OUT OF OR IN CONNECTION WITH THE USE OR
 *   PERFORMANCE OF THIS SOFTWARE.
 */
#pragma once

#include "encodec.h"
#include "ggml-backend.h"
#include "ggml.h"

#include <stdbool.h>
#include <stdint.h>

#ifndef BARK_API
#  define BARK_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* reference bark.h:37-41 */
enum bark_verbosity_level { LOW = 0, MEDIUM = 1, HIGH = 2 };

/* reference bark.h:43-47 */
enum bark_encoding_step { SEMANTIC = 0, COARSE = 1, FINE = 2 };

struct bark_context;
struct bark_model;
struct bark_vocab;
struct gpt_model;

/* reference bark.h:58 - invoked on the calling thread, once per decode step */
typedef void (*bark_progress_callback)(struct bark_context * bctx, enum bark_encoding_step step, int progress,
                                       void * user_data);

/* reference bark.h:60-79 */
struct bark_statistics {
    int64_t t_load_us;          /* model load                      */
    int64_t t_eval_us;          /* whole bark_generate_audio call  */
    int64_t t_semantic_us;      /* semantic stage                  */
    int64_t t_coarse_us;        /* coarse stage                    */
    int64_t t_fine_us;          /* fine stage                      */
    int32_t n_sample_semantic;  /* tokens sampled per stage        */
    int32_t n_sample_coarse;
    int32_t n_sample_fine;
};

/* reference bark.h:81-141 - field order is ABI */
struct bark_context_params {
    enum bark_verbosity_level verbosity;
    float   temp;                       /* semantic + coarse sampling temperature (0 => greedy) */
    float   fine_temp;                  /* fine sampling temperature (0 => greedy)              */
    float   min_eos_p;                  /* semantic early-stop probability                      */
    int32_t sliding_window_size;        /* coarse: new tokens per window                        */
    int32_t max_coarse_history;         /* coarse: history tokens re-fed per window             */
    int32_t sample_rate;
    int32_t target_bandwidth;
    int32_t cls_token_id;
    int32_t sep_token_id;
    int32_t n_steps_text_encoder;       /* max semantic tokens                                  */
    int32_t text_pad_token;
    int32_t text_encoding_offset;
    float   semantic_rate_hz;
    int32_t semantic_pad_token;
    int32_t semantic_vocab_size;
    int32_t semantic_infer_token;
    float   coarse_rate_hz;
    int32_t coarse_infer_token;
    int32_t coarse_semantic_pad_token;
    int32_t n_coarse_codebooks;
    int32_t n_fine_codebooks;
    int32_t codebook_size;
    bark_progress_callback progress_callback;
    void *  progress_callback_user_data;
};

/* reference bark.h:148 ; defaults bark.cpp:2202-2232 */
BARK_API struct bark_context_params bark_context_default_params(void);

/* reference bark.h:158-161 ; returns NULL on failure (diagnostics on stderr) */
BARK_API struct bark_context * bark_load_model(const char * model_path, struct bark_context_params params,
                                               uint32_t seed);

/* reference bark.h:171-174 ; n_threads is a CPU-backend hint and is ignored by the HIP engine */
BARK_API bool bark_generate_audio(struct bark_context * bctx, const char * text, int n_threads);

/* reference bark.h:182-192 ; buffer owned by the context, valid until the next generate/free */
BARK_API float * bark_get_audio_data(struct bark_context * bctx);
BARK_API int     bark_get_audio_data_size(struct bark_context * bctx);

/* reference bark.h:200-219 */
BARK_API int64_t bark_get_load_time(struct bark_context * bctx);
BARK_API int64_t bark_get_eval_time(struct bark_context * bctx);
BARK_API void    bark_reset_statistics(struct bark_context * bctx);

/* reference bark.h:229-232 - rewrites an f16 / f32 model file with the GPT matrices in a ggml block format (q4_0, q4_1, q5_0,
 * q5_1, q8_0: the types of examples/quantize); host-only, needs no GPU; false + a message on stderr for anything else */
BARK_API bool bark_model_quantize(const char * fname_inp, const char * fname_out, enum ggml_ftype ftype);

/* reference bark.h:239-240 ; NULL-safe */
BARK_API void bark_free(struct bark_context * bctx);

#ifdef __cplusplus
}
#endif
