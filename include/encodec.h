/* encodec.h - SHIM (reference bark.h:20 includes it; callers of bark.h use nothing from it).
 * The EnCodec decoder itself is part of the HIP engine (bark.cpp_amd/csrc/codec_kernels.hip, engine_codec.hip). */
#pragma once
#include "ggml.h"
