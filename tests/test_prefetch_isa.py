"""Static check of the opt-in weight-prefetch kernels (BARK_HIP_WPREFETCH; NextWeights in kernels.h, prefetch_next_weights in device_utils.h).

The prefetch requests are loads whose result nobody waits for.  That is safe only if the VGPR they return into is never handed to another
value while a request may be in flight - a property of the COMPILED code, so it is checked on the gfx950 assembly hipcc produces for the two
files that hold such kernels (no GPU needed; ~1 minute of cross-compilation)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _flags():
    # the product's own flags (bark.cpp_amd/build.sh), so that the checked code is the shipped code
    txt = open(os.path.join(ROOT, "bark.cpp_amd", "build.sh")).read()
    line = next(l for l in txt.splitlines() if l.startswith("FLAGS="))
    flags = line[len("FLAGS="):].strip().strip('"')
    flags = flags.replace("$HERE/../include", os.path.join(ROOT, "include")).replace("$SRC", os.path.join(ROOT, "bark.cpp_amd", "csrc"))
    return flags.split()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_prefetch_sink_register_is_never_reused(tmp_path):
    outs = []
    procs = []
    for f in ("kernels.hip", "attention_kernels.hip"):
        out = str(tmp_path / (f + ".s"))
        outs.append(out)
        procs.append(subprocess.Popen([HIPCC] + _flags() + ["-S", "--cuda-device-only", os.path.join(ROOT, "bark.cpp_amd", "csrc", f), "-o", out],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        log = p.communicate()[0].decode()
        assert p.returncode == 0, log
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_prefetch_isa.py")] + outs, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    # every decode kernel that can carry a request was seen: 8 + 8 x 4 GEMV instantiations would be 40, the LayerNorm-fused ones stop at
    # K = 1024 (5 widths x 4), plus the attention kernel
    last = r.stdout.strip().splitlines()[-1]
    assert last.endswith("0 violations") and int(last.split()[0]) >= 29, r.stdout
