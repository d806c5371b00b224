import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bark_amd_loader import load_package
from tools.make_synth_model import ensure_model
pkg = load_package()
c = pkg.BarkContext.load_model(ensure_model("small", 0), pkg.default_params(temp=0.0, fine_temp=0.0), 0)
codes = np.random.default_rng(0).integers(0, 1024, (8, 384)).astype(np.int32)
for _ in range(3):
    c.codec_decode(codes)
import time
t0 = time.perf_counter()
for _ in range(20):
    c.codec_decode(codes)
print("codec decode T=384: %.3f ms per call (host wall clock incl. the PCM copy)" % ((time.perf_counter() - t0) / 20 * 1e3))
c.free()
