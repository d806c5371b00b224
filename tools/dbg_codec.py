import sys, numpy as np
sys.path.insert(0, '.')
from bark_amd_loader import load_package
from oracle.pyoracle import Oracle
from tools.make_synth_model import ensure_model
pkg = load_package()
path = ensure_model("toy", 0)
orc = Oracle(path, 4)
for T in (50, 60, 64, 65, 96, 128, 129, 150, 205, 50):
    ctx = pkg.BarkContext.load_model(path, pkg.default_params(temp=0.0, fine_temp=0.0), 0)
    codes = np.random.default_rng(T).integers(0, 1024, (8, T)).astype(np.int32)
    g = ctx.codec_tap(codes, 1); r = orc.codec_tap(codes, 1)
    bad = np.flatnonzero(g != r)
    print("T", T, "stage1 mismatch", bad.size, "of", g.size, "first t", (bad % T)[:6], "ch", (bad // T)[:6], flush=True)
    ctx.free()
