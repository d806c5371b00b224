"""The -m gpu parity tests, run on a CPU: tests/simt/build_engine.py compiles the WHOLE engine (bark.cpp_amd/csrc: host control flow, graph capture and
replay, stage loops, every kernel) for the host against the stand-in for the HIP runtime of tests/simt (work-items as fibers, matrix-core instructions as
wave-wide rendezvous with the arithmetic the device probes established), and BARK_HIP_LIBRARY points the Python binding at the result.  A selection of the
GPU suite then runs unchanged in a process of its own - the product's source against the oracle, bit for bit, without a GPU.  Test infrastructure: the product
is libbark.so and has no CPU path ("Parity without a GPU")."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

# toy-model tests that take seconds each under emulation: tokenizer and hyper-parameters through the C ABI, prompt passes and decode steps of the semantic
# and the coarse model (ragged prompt lengths: every tile shape of the prefill product and attention), fine forward passes (the C1m tile product and the
# whole-window attention), the codec at several lengths (RVQ, the matrix-core convolutions, the LSTM wave front), loader refusals
SELECTION = ("test_library_describes_a_gfx950_device or test_hparams or test_tokenizer or test_semantic_eval_prefill_and_decode or test_coarse_eval_prefill_and_decode "
             "or test_coarse_prefill_ragged_lengths or test_fine_eval or test_codec_decode or refused or fails_cleanly")
# minutes each under emulation (BARK_SIM_FULL=1): the five block formats (v_dot4 decode, i8 matrix cores for rows), graph replay against eager launches,
# the matrix-core product against the one-row-per-wave kernel.  All of them passed on the emulated engine when this file was written.
SLOW_SELECTION = "test_other_block_formats_toy or test_q4_0_logits_toy or test_graph_and_eager_agree or test_mfma_gemm_matches_row_kernel"


@pytest.fixture(scope="module")
def sim_engine(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("ROCm's clang is not installed")
    sys.path.insert(0, os.path.join(ROOT, "tests", "simt"))
    import build_engine
    return build_engine.build(str(tmp_path_factory.mktemp("sim_engine")))


def _pytest_on(sim_engine, files, k, workers=4, timeout=1500, env_add=None):
    env = dict(os.environ); env["BARK_HIP_LIBRARY"] = sim_engine
    env.update(env_add or {})
    cmd = [sys.executable, "-m", "pytest"] + files + ["-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-n", str(workers), "-k", k]
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpu_parity_tests_pass_on_the_host_emulated_engine(sim_engine):
    r = _pytest_on(sim_engine, ["tests/test_gpu_parity.py", "tests/test_gpu_loader.py"], SELECTION)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:]
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in tail and "failed" not in tail, tail
    assert int(tail.split(" passed")[0].split()[-1]) >= 30, tail


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get("BARK_SIM_FULL") != "1", reason="minutes of emulation: set BARK_SIM_FULL=1")
def test_slower_gpu_parity_tests_pass_on_the_host_emulated_engine(sim_engine):
    r = _pytest_on(sim_engine, ["tests/test_gpu_parity.py"], SLOW_SELECTION, workers=8, timeout=3000)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


_PREFETCH_CHILD = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from bark_amd_loader import load_package
from oracle.pyoracle import Oracle
pkg = load_package()
n = 16
ctx = pkg.BarkContext.load_model(%r, pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=n), 0)
orc = Oracle(%r, n_threads=4)
p = orc.params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=n)
prompt = orc.tokenize("hello world , the water is cold today and the river runs fast !")
so = orc.semantic(prompt, p)
assert np.array_equal(ctx.semantic(prompt), so), "semantic tokens"
assert np.array_equal(ctx.coarse(so), orc.coarse(so, p)), "coarse tokens"
print("PREFETCH_SIM_OK")
ctx.free(); orc.close()
"""


_SLOT_JOB_CHILD = r"""
import sys, json, hashlib
sys.path.insert(0, %r)
import numpy as np
import bench
from bark_amd_loader import load_package
pkg = load_package()
ctx = pkg.BarkContext.load_model(%r, pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=12), 0)
ctx.reserve_batch(8)
texts = bench.synth_prompts(4, seed=3)
reqs = [ctx.request_params(n_steps_text_encoder=3 + (5 * i) %% 12, temp=0.7 if i %% 4 == 1 else 0.0, seed=50 + i) for i in range(4)]
h = hashlib.sha256()
for r in ctx.generate_batch(texts, params=reqs):
    for k in ("semantic", "coarse", "fine", "pcm"):
        h.update(np.ascontiguousarray(r[k]).tobytes())
print("RESULT " + h.hexdigest())
ctx.free()
"""


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get("BARK_SIM_FULL") != "1", reason="about five minutes per arm under emulation (the fine stage): set BARK_SIM_FULL=1")
def test_few_slot_experiment_gives_the_default_job_on_the_emulated_engine(sim_engine, toy_model):
    """The few-slot lock-step route (BARK_HIP_FEW_SLOTS) end to end on the emulated engine: a ragged job of four utterances (one sampled) on
    eight slots - host plumbing, graph capture per live slot count, the per-slot kernels - gives the ids and the PCM of the default lock-step route."""
    got = []
    procs = []
    for arm in ({"BARK_HIP_FEW_SLOTS": "0"}, {}):
        env = dict(os.environ, BARK_HIP_LIBRARY=sim_engine); env.update(arm)
        procs.append(subprocess.Popen([sys.executable, "-c", _SLOT_JOB_CHILD % (ROOT, toy_model)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=3000)
        line = [l for l in out.splitlines() if l.startswith("RESULT ")]
        assert p.returncode == 0 and line, (out[-500:], err[-1500:])
        got.append(line[0])
    assert got[0] == got[1]
