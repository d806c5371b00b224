#!/bin/bash
# Builds the two diagnostic variants of tools/state_race_demo.py and prints, from the compiled code of sample_greedy_kernel, where the load of st->step
# sits relative to the barriers, the injected lag (s_sleep) and thread 0's store of the state: the evidence behind DESIGN.md section 10.  CPU only.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
# the two diagnostic libraries differ from the product in ONE object (misc_kernels.o); bark.cpp_amd/build.sh --variant compiles it with the product's own
# flags and links it with the product's own object list into lib/diag/ (nothing is duplicated here)
"$R/bark.cpp_amd/build.sh" --variant lag_old -DBARK_DIAG_PLAIN_STATE_LOADS -DBARK_DIAG_LAG_WAVES=4
"$R/bark.cpp_amd/build.sh" --variant lag_fix -DBARK_DIAG_LAG_WAVES=4
[ "$1" = "--build-only" ] && exit 0
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -mllvm -amdgpu-kernarg-preload-count=16 -I$R/include -I$R/bark.cpp_amd/csrc -S --cuda-device-only"
show() {
    echo "== $1: sample_greedy_kernel, line of the kernel's assembly: instruction   (flags: ${2:-none})"
    /opt/rocm/bin/hipcc $FLAGS $2 "$R/bark.cpp_amd/csrc/misc_kernels.hip" -o /tmp/_misc_$$.s 2>/dev/null
    awk '/^_ZN7barkhip20sample_greedy_kernel/ {on=1; n=0; next} /^\.Lfunc_end/ {on=0} on {n++; if ($0 ~ /s_barrier|s_sleep|s_load_dword s[0-9]+, s\[[0-9:]+\], 0x8$|flat_load_dword.*offset:8|global_store_dwordx3/) print "   " n ": " $0}' /tmp/_misc_$$.s | head -12
    rm -f /tmp/_misc_$$.s
}
show "round-4 form of the kernel (plain loads of the state)" "-DBARK_DIAG_PLAIN_STATE_LOADS"
show "product build (volatile loads: the fix)" ""
show "lag_old (plain loads + waves 1..15 sleep behind the second barrier)" "-DBARK_DIAG_PLAIN_STATE_LOADS -DBARK_DIAG_LAG_WAVES=4"
show "lag_fix (volatile loads + the same sleep)" "-DBARK_DIAG_LAG_WAVES=4"
echo "The first two s_barrier lines are the fast path's only barriers (the others sit on the exact path); global_store_dwordx3 is thread 0 writing {n_past, cur_token, step}."
echo "Round-4 form: the scalar load of st->step (offset 0x8) is issued BEHIND both barriers; nothing orders it against the store of a wave that is ahead."
