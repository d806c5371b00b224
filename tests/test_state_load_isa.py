"""The compiled sampler must read the stage state before its first barrier (regression guard for the root cause of round 4's red GPU suite).

sample_greedy_kernel reads `st->step` / `st->n_past` at its top and thread 0 rewrites the state at its end; on the fast path no barrier lies
between the other waves' last use of `step` (the coarse stage's codebook parity, reference /root/reference/bark.cpp:1829-1841) and that store.
With plain loads the compiler sank the scalar load of `st->step` behind the second __syncthreads(): a wave that fell behind wave 0 read the
advanced step and embedded the next token from the wrong codebook's row (tools/state_race_demo.sh prints the same listing,
tools/state_race_demo.py forces the race on a GPU).  The loads are volatile now; this test compiles the kernel for gfx950 (hipcc cross-compiles,
no GPU) and checks the assembly: the state is read by a load in front of the FIRST s_barrier and no scalar load of offset 8 follows it."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sampler_asm(extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
           "-fvisibility=hidden", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "bark.cpp_amd", "csrc"),
           "-S", "--cuda-device-only", *extra, os.path.join(ROOT, "bark.cpp_amd", "csrc", "misc_kernels.hip"), "-o", "-"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN7barkhip20sample_greedy_kernel"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return [l.strip() for l in lines[start:end]]


def test_sampler_reads_the_stage_state_before_its_first_barrier():
    body = _sampler_asm()
    first_barrier = next(i for i, l in enumerate(body) if l.startswith("s_barrier"))
    early = [l for l in body[:first_barrier] if re.match(r"(flat|global)_load_dword v\d+, .*offset:8\b", l)]
    assert early, "no load of st->step (offset 8 of the state) in front of the first barrier"
    late = [l for l in body[first_barrier:] if re.match(r"s_load_dword s\d+, s\[\d+:\d+\], 0x8$", l)]
    assert not late, f"a scalar load of offset 8 behind the first barrier: {late}"


def test_the_round_4_form_of_the_kernel_had_the_load_behind_the_barriers():
    """The same check on the kernel as it was (-DBARK_DIAG_PLAIN_STATE_LOADS keeps that form for tools/state_race_demo.py): the guard above must
    be able to see the defect.  Should a future compiler place this load early of its own accord, this test - not the product - needs attention."""
    body = _sampler_asm(["-DBARK_DIAG_PLAIN_STATE_LOADS"])
    barriers = [i for i, l in enumerate(body) if l.startswith("s_barrier")]
    late = [i for i, l in enumerate(body) if i > barriers[1] and re.match(r"s_load_dword s\d+, s\[\d+:\d+\], 0x8$", l)]
    assert late, "the plain-load form no longer shows the sunk load (compiler changed?)"


def test_the_state_is_rewritten_behind_the_kernels_last_barrier():
    """Round 6 (advisor): the ordering is structural as well - thread 0's stores of the state {n_past, cur_token, step} (one global_store_dwordx3 through the
    state's scalar base) follow the LAST s_barrier of the kernel, which every wave reaches only after its last use of `step` / `np_next`; the volatile
    loads checked above stay as the second guard."""
    body = _sampler_asm()
    last_barrier = max(i for i, l in enumerate(body) if l.startswith("s_barrier"))
    x3 = [i for i, l in enumerate(body) if re.match(r"global_store_dwordx3 v\d+, v\[\d+:\d+\], s\[\d+:\d+\]$", l)]
    assert x3, "the store of {n_past, cur_token, step} was not found (did the layout of StepState change?)"
    assert all(i > last_barrier for i in x3), (last_barrier, x3)
    # and the next step's embedding row (the other waves' last use of the loaded state) is written in front of that barrier
    assert any(re.match(r"global_store_dword v", l) for l in body[last_barrier - 12:last_barrier])
