#!/usr/bin/env python3
"""Deterministic demonstration of the root cause of round 4's red GPU suite.

sample_greedy_kernel read the stage state (`st->step`, the coarse stage's codebook parity) with plain loads at its top and thread 0 rewrites the
state at its end; on the fast path no barrier lies in between, and the compiler had sunk the scalar load of `st->step` BELOW the second
__syncthreads().  A wave that falls behind wave 0 by more than the tail of the kernel then reads the step thread 0 has already advanced, takes
the other codebook's token offset and writes its 64 elements of the next token's embedding from the wrong row.  Under concurrent contexts that
happened about once per 10^6 lock steps (tools/clone_stress.py); here it is forced: two diagnostic builds of the library in which waves 1 .. 15
of the sampler sleep ~16 us behind the second barrier (-DBARK_DIAG_LAG_WAVES=4),

    lib/diag/libbark_lag_old.so   the state loads as they were (-DBARK_DIAG_PLAIN_STATE_LOADS; tools/state_race_demo.sh checks the ISA: s_load behind the lag)
    lib/diag/libbark_lag_fix.so   the volatile loads of the fix (issued before the first barrier)

each generating one toy utterance (single-utterance path) and one 5-utterance lock-step job, compared with the oracle.
Expected: lag_old - coarse ids differ; lag_fix - everything equal.   python tools/state_race_demo.py <variant>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
variant = sys.argv[1]
os.environ["BARK_HIP_LIBRARY"] = os.path.join(ROOT, "bark.cpp_amd", "lib", "diag", "libbark_%s.so" % variant)

import numpy as np                                   # noqa: E402
import bench                                         # noqa: E402
from bark_amd_loader import load_package             # noqa: E402
from oracle.pyoracle import Oracle                   # noqa: E402
from tools.make_synth_model import ensure_model      # noqa: E402

pkg = load_package()
path = ensure_model("toy", 0)
texts = bench.synth_prompts(5)
orc = Oracle(path, n_threads=4)
want = []
for t in texts:
    orc.seed(0)
    want.append(orc.generate(t, orc.params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=30)))
orc.close()
ctx = pkg.BarkContext.load_model(path, pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=30), 0)
bad = 0
assert ctx.generate_audio(texts[0])
for name, got in (("semantic", ctx.semantic_tokens()), ("coarse", ctx.coarse_tokens())):
    d = int((np.asarray(got).ravel() != np.asarray(want[0][name]).ravel()).sum()) if np.asarray(got).size == np.asarray(want[0][name]).size else -1
    print(f"{variant}: single utterance {name}: {d} ids differ from the oracle")
    bad += d != 0
res = ctx.generate_batch(texts)
for i, r in enumerate(res):
    for name in ("semantic", "coarse"):
        d = int((np.asarray(r[name]).ravel() != np.asarray(want[i][name]).ravel()).sum()) if np.asarray(r[name]).size == np.asarray(want[i][name]).size else -1
        if d:
            print(f"{variant}: lock-step job utterance {i} {name}: {d} ids differ from the oracle")
        bad += d != 0
ctx.free()
print(f"{variant}: {'DIFFERS from the oracle' if bad else 'equal to the oracle'} ({bad} arrays)")
