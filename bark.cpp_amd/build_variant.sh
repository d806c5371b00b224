#!/bin/bash
# Builds an alternate libbark (diagnostic / A-B variants) next to the product library:
#   build_variant.sh <suffix> [extra hipcc flags...]   ->  lib/libbark_<suffix>.so   (select it with BARK_HIP_LIBRARY)
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SUF="$1"; shift
SRC="$HERE/csrc"; OUT="$HERE/lib"; OBJ="$OUT/obj_$SUF"
mkdir -p "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -mllvm -amdgpu-kernarg-preload-count=16 -Wall -Wno-unused-function -Wno-pass-failed -I$HERE/../include -I$SRC $*"
pids=()
for f in kernels.hip fast_kernels.hip quant_kernels.hip attention_kernels.hip misc_kernels.hip codec_kernels.hip engine_load.hip engine.hip engine_codec.hip engine_batch.hip engine_timing.hip api.hip batcher.hip; do
    $HIPCC $FLAGS -c "$SRC/$f" -o "$OBJ/${f%.*}.o" & pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libbark_$SUF.so" "$OBJ"/*.o "$OUT/obj/model_file.o" "$OUT/obj/tokenizer.o" "$OUT/obj/quantize.o"
echo "built $OUT/libbark_$SUF.so"
