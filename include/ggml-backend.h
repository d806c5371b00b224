/* ggml-backend.h - SHIM (reference bark.h:21 includes it; callers use nothing from it). */
#pragma once
#include "ggml.h"
