#!/usr/bin/env python3
"""In-kernel time line of attn_window_kernel (the fine model's whole-window attention): per phase, microseconds averaged over workgroups and waves.
Needs the diagnostic build:  bark.cpp_amd/build_variant.sh attnw -DATTNW_STAMPS   (s_memrealtime stamps into a __device__ array, 100 MHz).
usage: BARK_HIP_LIBRARY=bark.cpp_amd/lib/libbark_attnw.so python tools/attnw_phases.py [windows]"""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BARK_HIP_LIBRARY", os.path.join(ROOT, "bark.cpp_amd", "lib", "libbark_attnw.so"))
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
pkg = load_package()
Z = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0), 0)
us, _ = ctx.time_fine_pass(4, Z)
lib = pkg.load_library()
buf = np.zeros(512 * 8 * 8, np.uint64)
assert lib.bark_hip_debug_attnw_stamps(C.c_void_p(buf.ctypes.data), buf.size) == 0
st = buf.reshape(512, 8, 8).astype(np.int64)
n_wg = min(512, 32 * 12 * Z)
st = st[:n_wg]
names = ["entry -> query tile staged, first K rows landed", "scores (128 MFMA per wave)", "max exchange (barrier)", "64 exponentials + sum exchange (barrier)",
         "scale + mix (128 MFMA per wave)", "partials -> LDS + barrier", "tree over the waves + store"]
print(f"fine pass {us:.1f} us over {Z} window(s); last launch of attn_window_kernel, {n_wg} workgroups x 8 waves, 100 MHz stamps")
d = np.diff(st, axis=2) / 100.0
for i, nme in enumerate(names):
    print(f"  {nme:58s} mean {d[:, :, i].mean():6.2f} us   min {d[:, :, i].min():6.2f}   max {d[:, :, i].max():6.2f}")
cy = np.zeros(512 * 8 * 8, np.uint64)
if hasattr(lib, "bark_hip_debug_attnw_cycles") and lib.bark_hip_debug_attnw_cycles(C.c_void_p(cy.ctypes.data), cy.size) == 0:
    cy = cy.reshape(512, 8, 8).astype(np.int64)[:n_wg]
    dc = np.diff(cy, axis=2).astype(np.float64)
    for i in (1, 4):
        print(f"  shader clock during '{names[i]}': {dc[:, :, i].mean():9.0f} cycles -> {dc[:, :, i].mean() / max(d[:, :, i].mean(), 1e-9) / 1000.0:.2f} GHz"
              f"   (128 MFMA x 128 cycles at two waves per SIMD = 16384 cycles of matrix-core time)")
tot = (st[:, :, 7] - st[:, :, 0]) / 100.0
print(f"  workgroup life (entry -> end)                              mean {tot.mean():6.2f} us   min {tot.min():6.2f}   max {tot.max():6.2f}")
span = (st[:, :, 7].max() - st[:, :, 0].min()) / 100.0
print(f"  launch span over the recorded workgroups: {span:.1f} us")
ctx.free()
