#!/usr/bin/env python3
"""build_engine.py OUT_DIR - the WHOLE engine (bark.cpp_amd/csrc) compiled for the host against the stand-in for the HIP runtime in this directory:
libbark_sim.so exports the same C ABI as libbark.so, every kernel runs work-item for work-item on fibers (hip/hip_runtime.h), memory is host memory, a
captured graph is the list of what was captured.  TEST INFRASTRUCTURE: it exists so that the -m "not gpu" suite can run the product's own source - host
control flow, graphs, lock-step jobs, every kernel - against the oracle; the product is libbark.so and has no CPU path.  The sources are patched
textually in OUT_DIR (inline asm, the buffer-load intrinsic binding, dynamic LDS); nothing under bark.cpp_amd/ is modified."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.environ.get("BARK_SIM_CSRC", os.path.join(ROOT, "bark.cpp_amd", "csrc"))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
HIP_SOURCES = ["kernels.hip", "fast_kernels.hip", "quant_kernels.hip", "attention_kernels.hip", "misc_kernels.hip", "codec_kernels.hip", "engine_load.hip", "engine.hip",
               "engine_codec.hip", "engine_batch.hip", "engine_timing.hip", "api.hip", "batcher.hip"]
CPP_SOURCES = ["model_file.cpp", "tokenizer.cpp", "quantize.cpp"]


def patch(text: str) -> str:
    # the LLVM buffer-load intrinsics bound by name: plain loads from the descriptor's base address
    text = re.sub(r'__device__ float4v llvm_amdgcn_raw_buffer_load_v4f32\([^;]*;',
                  'inline float4v llvm_amdgcn_raw_buffer_load_v4f32(int4v rsrc, int voffset, int soffset, int) { const char * b = reinterpret_cast<const char *>('
                  '((unsigned long long) (unsigned) rsrc.y << 32) | (unsigned) rsrc.x); float4v r; memcpy(&r, b + voffset + soffset, 16); return r; }', text)
    text = re.sub(r'__device__ float   llvm_amdgcn_raw_buffer_load_f32\([^;]*;',
                  'inline float llvm_amdgcn_raw_buffer_load_f32(int4v rsrc, int voffset, int soffset, int) { const char * b = reinterpret_cast<const char *>('
                  '((unsigned long long) (unsigned) rsrc.y << 32) | (unsigned) rsrc.x); float r; memcpy(&r, b + voffset + soffset, 4); return r; }', text)
    text = text.replace('asm("" : "+v"(v));', '')                                   # the f16 rounding point: no fused conversion to keep apart on the host
    # weight-prefetch requests: nothing to load, but every address must lie inside an allocation (the plan's slicing is checked end to end)
    request = ('{ (void) after; (void) sink; if (!sim::inside_an_allocation(p + off, 4)) { fprintf(stderr, "sim: a prefetch request outside every allocation\\n"); abort(); } '
               'sim::prefetch_requests()++; }')
    text = re.sub(r'asm volatile\("global_load_dword.*?\)\);', lambda m: request, text)
    text = text.replace('asm volatile("; NWPF hold %0" :: "v"(sink));', '(void) sink;')
    text = re.sub(r'extern __shared__ (__attribute__\(\(aligned\(16\)\)\) )?(\w+) (\w+)\[\];', r'static \2 \3[65536];', text)      # dynamic LDS
    return text


def build(out_dir: str) -> str:
    os.makedirs(out_dir, exist_ok=True)
    src = os.path.join(out_dir, "src")
    os.makedirs(src, exist_ok=True)
    for name in os.listdir(CSRC):
        if name.endswith((".h", ".hip", ".cpp")):
            open(os.path.join(src, name), "w").write(patch(open(os.path.join(CSRC, name)).read()))
    flags = ["-std=c++20", "-O1", "-mfma", "-mf16c", "-mavx2", "-ffp-contract=off", "-pthread", "-fPIC", "-fvisibility=hidden", "-Wno-everything",
             "-I", HERE, "-I", src, "-I", os.path.join(ROOT, "include")]
    procs, objs = [], []
    for name in HIP_SOURCES + CPP_SOURCES:
        o = os.path.join(out_dir, name.rsplit(".", 1)[0] + ".o")
        objs.append(o)
        procs.append((name, subprocess.Popen([CLANG, "-x", "c++"] + flags + ["-c", os.path.join(src, name), "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for name, p in procs:
        log = p.communicate()[0].decode()
        if p.returncode != 0:
            raise RuntimeError(f"{name}: {log[-3000:]}")
    so = os.path.join(out_dir, "libbark_sim.so")
    r = subprocess.run([CLANG, "-shared", "-pthread", "-o", so] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    return so


if __name__ == "__main__":
    print(build(sys.argv[1]))
