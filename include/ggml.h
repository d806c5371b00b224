/*
 * ggml.h - SHIM.  The reference's bark.h includes <ggml.h> (bark.h:22) and its callers use a
 * handful of names from it.  This file provides exactly those names so that unmodified callers
 * (examples/main/main.cpp:26-27,81 ; examples/quantize/main.cpp:30-36,69-71) compile against the
 * MI355X engine.  There is no tensor library behind it.
 */
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* numeric values follow ggml's public enum */
enum ggml_ftype {
    GGML_FTYPE_UNKNOWN     = -1,
    GGML_FTYPE_ALL_F32     = 0,
    GGML_FTYPE_MOSTLY_F16  = 1,
    GGML_FTYPE_MOSTLY_Q4_0 = 2,
    GGML_FTYPE_MOSTLY_Q4_1 = 3,
    GGML_FTYPE_MOSTLY_Q8_0 = 7,
    GGML_FTYPE_MOSTLY_Q5_0 = 8,
    GGML_FTYPE_MOSTLY_Q5_1 = 9,
};

enum ggml_type {
    GGML_TYPE_F32  = 0,
    GGML_TYPE_F16  = 1,
    GGML_TYPE_Q4_0 = 2,
    GGML_TYPE_Q4_1 = 3,
    GGML_TYPE_Q5_0 = 6,
    GGML_TYPE_Q5_1 = 7,
    GGML_TYPE_Q8_0 = 8,
};

struct ggml_context;
struct ggml_init_params { size_t mem_size; void * mem_buffer; bool no_alloc; };

__attribute__((visibility("default"))) void    ggml_time_init(void);
__attribute__((visibility("default"))) int64_t ggml_time_us(void);
__attribute__((visibility("default"))) int64_t ggml_time_ms(void);
__attribute__((visibility("default"))) struct ggml_context * ggml_init(struct ggml_init_params params);
__attribute__((visibility("default"))) void    ggml_free(struct ggml_context * ctx);

#ifdef __cplusplus
}
#endif
