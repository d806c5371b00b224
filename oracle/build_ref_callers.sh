#!/bin/bash
# Builds the reference's OWN callers of bark.h - examples/main, examples/server, examples/quantize - unmodified, from the
# sources where they lie under /root/reference, against this repository's libbark.so.  Outputs go to oracle/_ref/ only
# (git-ignored, but shipped to the GPU box by gpurun), where the GPU tests run them end to end: the reference's CLI and HTTP
# server serving from the MI355X engine.  The reference LIBRARY itself (bark.cpp) cannot be built here: its ggml / encodec.cpp
# submodule is absent from the checkout; these binaries are the reference-side half of the drop-in boundary.
# No reference source is copied into the repository.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(dirname "$HERE")"
REF="${BARK_REFERENCE:-/root/reference}"
OUT="$HERE/_ref"
[ -d "$REF/examples/main" ] || { echo "no reference checkout at $REF: nothing to build"; exit 0; }
[ -f "$ROOT/bark.cpp_amd/lib/libbark.so" ] || { echo "libbark.so is not built yet"; exit 1; }
mkdir -p "$OUT"
CXX="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-std=c++17 -O1 -w -I$ROOT/include -I$REF/examples -L$ROOT/bark.cpp_amd/lib -lbark -lpthread -Wl,-rpath,\$ORIGIN/../../bark.cpp_amd/lib"
newer() { [ ! -f "$1" ] || [ "$2" -nt "$1" ] || [ "$ROOT/include/bark.h" -nt "$1" ]; }
if newer "$OUT/bark_main" "$REF/examples/main/main.cpp"; then
    $CXX "$REF/examples/main/main.cpp" "$REF/examples/common.cpp" $FLAGS -o "$OUT/bark_main" &
fi
if newer "$OUT/bark_quantize" "$REF/examples/quantize/main.cpp"; then
    $CXX "$REF/examples/quantize/main.cpp" $FLAGS -o "$OUT/bark_quantize" &
fi
if newer "$OUT/bark_server" "$REF/examples/server/server.cpp"; then
    $CXX "$REF/examples/server/server.cpp" "$REF/examples/common.cpp" -I"$REF/examples/server" $FLAGS -o "$OUT/bark_server" &
fi
wait
ls -la "$OUT"
