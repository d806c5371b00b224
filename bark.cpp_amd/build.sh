#!/bin/bash
# Builds bark.cpp_amd/lib/libbark.so for gfx950 (cross-compiles without a GPU).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/csrc"
OUT="$HERE/lib"
mkdir -p "$OUT" "$OUT/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -mllvm -amdgpu-kernarg-preload-count=16 -Wall -Wno-unused-function -I$HERE/../include -I$SRC"
# explicit object list: stale objects of renamed / split sources are never linked
OBJS=()
pids=()
for f in kernels.hip fast_kernels.hip quant_kernels.hip attention_kernels.hip misc_kernels.hip codec_kernels.hip engine_load.hip engine.hip engine_codec.hip engine_batch.hip engine_timing.hip api.hip batcher.hip; do
    o="$OUT/obj/${f%.*}.o"
    OBJS+=("$o")
    if [ ! -f "$o" ] || [ "$SRC/$f" -nt "$o" ] || [ -n "$(find "$SRC" "$HERE/../include" -name '*.h' -newer "$o" 2>/dev/null | head -1)" ]; then
        $HIPCC $FLAGS -c "$SRC/$f" -o "$o" &
        pids+=($!)
    fi
done
for f in model_file.cpp tokenizer.cpp; do
    o="$OUT/obj/${f%.*}.o"
    OBJS+=("$o")
    if [ ! -f "$o" ] || [ "$SRC/$f" -nt "$o" ] || [ -n "$(find "$SRC" -name '*.h' -newer "$o" 2>/dev/null | head -1)" ]; then
        g++ -O2 -std=c++17 -fPIC -fvisibility=hidden -Wall -I"$SRC" -c "$SRC/$f" -o "$o" &
        pids+=($!)
    fi
done
# quantize.cpp uses _Float16 conversions: ROCm's clang has the type on the host, gcc 11 does not
o="$OUT/obj/quantize.o"
OBJS+=("$o")
if [ ! -f "$o" ] || [ "$SRC/quantize.cpp" -nt "$o" ] || [ -n "$(find "$SRC" -name '*.h' -newer "$o" 2>/dev/null | head -1)" ]; then
    /opt/rocm/lib/llvm/bin/clang++ -O2 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wall -I"$SRC" -c "$SRC/quantize.cpp" -o "$o" &
    pids+=($!)
fi
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libbark.so" "${OBJS[@]}"
echo "built $OUT/libbark.so"
# native batching HTTP front end (examples/batch_server.cpp): same protocol as the reference's example server, requests travel as lock-step batches
g++ -O2 -std=c++17 -Wall -I"$HERE/../include" -I"$HERE/examples" "$HERE/examples/batch_server.cpp" -L"$OUT" -lbark -lpthread -Wl,-rpath,'$ORIGIN' -o "$OUT/bark_batch_server"
echo "built $OUT/bark_batch_server"
