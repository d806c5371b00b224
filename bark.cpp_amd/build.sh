#!/bin/bash
# Builds bark.cpp_amd/lib/libbark.so for gfx950 (cross-compiles without a GPU).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/csrc"
OUT="$HERE/lib"
mkdir -p "$OUT" "$OUT/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -mllvm -amdgpu-kernarg-preload-count=16 -Wall -Wno-unused-function -I$HERE/../include -I$SRC"
# --variant <suffix> <extra flags...>: a DIAGNOSTIC library that differs from the product in one object - misc_kernels.hip compiled with the extra flags,
# linked with the product's other objects (same flags, same object list: nothing to drift) into lib/diag/libbark_<suffix>.so, away from the product library
# (tools/state_race_demo.sh: the sampler with an injected lag, with and without round 4's race)
if [ "$1" = "--variant" ]; then
    SUF="$2"; shift 2
    [ -f "$OUT/libbark.so" ] || "$0" > /dev/null
    mkdir -p "$OUT/diag/obj"
    $HIPCC $FLAGS "$@" -c "$SRC/misc_kernels.hip" -o "$OUT/diag/obj/misc_kernels_$SUF.o"
    VOBJS=()
    for f in kernels fast_kernels quant_kernels attention_kernels codec_kernels engine_load engine engine_codec engine_batch engine_timing api batcher model_file tokenizer quantize; do VOBJS+=("$OUT/obj/$f.o"); done
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/diag/libbark_$SUF.so" "${VOBJS[@]}" "$OUT/diag/obj/misc_kernels_$SUF.o"
    echo "built $OUT/diag/libbark_$SUF.so"
    exit 0
fi
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
# diagnostic variants (--variant) link the product's objects: one that is older than an object it was made from no longer matches the library (a missing export
# shows only at load time) - remove it, tools/state_race_demo.sh --build-only (run by __graft_entry__.build()) makes it again and the test that needs it skips meanwhile
for v in "$OUT"/diag/libbark_*.so; do
    [ -f "$v" ] || continue
    for o in "${OBJS[@]}"; do if [ "$o" -nt "$v" ]; then rm -f "$v"; echo "removed stale $v"; break; fi; done
done
# native batching HTTP front end (examples/batch_server.cpp): same protocol as the reference's example server, requests travel as lock-step batches
g++ -O2 -std=c++17 -Wall -I"$HERE/../include" -I"$HERE/examples" "$HERE/examples/batch_server.cpp" -L"$OUT" -lbark -lpthread -Wl,-rpath,'$ORIGIN' -o "$OUT/bark_batch_server"
echo "built $OUT/bark_batch_server"
