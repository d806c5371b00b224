"""pytest configuration: markers, shared fixtures (synthetic model files, CPU oracle)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")
    # a fresh checkout has no binaries (they are git-ignored): build the product library and the oracle once, exactly as
    # __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU).  A failed build fails the run - no fallback.
    import subprocess
    lib = os.path.join(ROOT, "bark.cpp_amd", "lib", "libbark.so")
    if not os.path.exists(lib):
        subprocess.check_call([os.path.join(ROOT, "bark.cpp_amd", "build.sh")])
    if not os.path.exists(os.path.join(ROOT, "oracle", "build", "libbark_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])


# Order of the GPU suite (the driver runs `pytest tests -x -q -m gpu`): a failure must not hide the tests that pin the hot path row by row, so the
# per-row parity tests run FIRST and the widest, most concurrent tests LAST (round 4 ended red on a concurrency test that was collected first and
# hid ~100 row-level tests behind `-x`).  Groups, in order; inside a group the file order is kept:
#   0 loader / device           1 per-row parity against the oracle (formats; bark-small / bark-large shapes)
#   2 boundary binaries, ranks  3 lock-step jobs (N1)          4 concurrency: clones on host threads, request collector, servers, soaks, opt-in experiments
_GPU_GROUP_BY_NAME = {
    2: ("test_reference_cli_binary", "test_reference_http_server_binary", "test_two_ranks_", "test_config5_rank_shard"),
    3: ("test_in_engine_batch", "test_larger_lock_step_batches", "test_batch_with_unequal_lengths", "test_lock_step_batch_", "test_q4_0_generate_and_lock_step_batch",
        "test_randomised_lock_step_jobs", "test_job_larger_than_the_slots", "test_job_tail_on_a_second_stream", "test_small_ragged_job", "test_lock_step_time_line_hook",
        "test_ragged_job_on_quantised", "test_few_slot_"),
    4: ("test_cloned_contexts_", "test_request_batcher_", "test_request_collector_", "test_native_batch_server", "test_device_and_host_sampling_agree_on_many",
        "test_concurrent_"),
}


def _gpu_group(item):
    name = item.originalname if getattr(item, "originalname", None) else item.name
    for grp, prefixes in _GPU_GROUP_BY_NAME.items():
        if name.startswith(prefixes):
            return grp
    return 0 if "test_gpu_loader" in item.nodeid or name in ("test_library_describes_a_gfx950_device", "test_hparams") else 1


def pytest_collection_modifyitems(config, items):
    gpu = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu:
        return
    order = {id(it): (_gpu_group(it), k) for k, it in enumerate(gpu)}
    gpu_sorted = sorted(gpu, key=lambda it: order[id(it)])
    rest = [it for it in items if not it.get_closest_marker("gpu")]
    items[:] = rest + gpu_sorted


@pytest.fixture(scope="session")
def toy_model():
    from tools.make_synth_model import ensure_model
    return ensure_model("toy", 0)


@pytest.fixture(scope="session")
def mini_model():
    from tools.make_synth_model import ensure_model
    return ensure_model("mini", 0)


@pytest.fixture(scope="session")
def small_model():
    from tools.make_synth_model import ensure_model
    return ensure_model("small", 0)


@pytest.fixture(scope="session")
def toy_oracle(toy_model):
    from oracle.pyoracle import Oracle
    o = Oracle(toy_model, n_threads=4)
    yield o
    o.close()


@pytest.fixture(scope="session")
def mini_oracle(mini_model):
    from oracle.pyoracle import Oracle
    o = Oracle(mini_model, n_threads=4)
    yield o
    o.close()


GGML_FTYPE = {"q4_0": 2, "q4_1": 3, "q8_0": 7, "q5_0": 8, "q5_1": 9}     # enum ggml_ftype (include/ggml.h)


def _quantized(src: str, fmt: str = "q4_0") -> str:
    """bark_model_quantize (native writer, no GPU needed) -> <src>_<fmt>.bin next to the f16 file, made once."""
    dst = src[:-4] + "_%s.bin" % fmt
    if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
        from bark_amd_loader import load_package
        lib = load_package().load_library()
        tmp = dst + ".tmp%d" % os.getpid()
        assert lib.bark_model_quantize(src.encode(), tmp.encode(), GGML_FTYPE[fmt])
        os.replace(tmp, dst)
    return dst


@pytest.fixture(scope="session")
def quantized_model():
    """factory: quantized_model(path_of_f16_file, "q5_1") -> path of the quantised file"""
    return _quantized


@pytest.fixture(scope="session")
def toy_q4_model(toy_model):
    return _quantized(toy_model)


@pytest.fixture(scope="session")
def mini_q4_model(mini_model):
    return _quantized(mini_model)


@pytest.fixture(scope="session")
def small_q4_model(small_model):
    return _quantized(small_model)


@pytest.fixture(scope="session")
def toy_q4_oracle(toy_q4_model):
    from oracle.pyoracle import Oracle
    o = Oracle(toy_q4_model, n_threads=4)
    yield o
    o.close()


@pytest.fixture(scope="session")
def mini_q4_oracle(mini_q4_model):
    from oracle.pyoracle import Oracle
    o = Oracle(mini_q4_model, n_threads=4)
    yield o
    o.close()


@pytest.fixture(scope="session")
def toy_f32_model(toy_model):
    """the same synthetic draws written as an f32 file (convert.py without --use-f16: every tensor f32, codec included)"""
    from tools.make_synth_model import write_model
    dst = os.path.join(os.path.dirname(toy_model), "bark_toy_s0_f32.bin")
    if not os.path.exists(dst):
        tmp = dst + ".tmp%d" % os.getpid()
        write_model(tmp, "toy", 0, use_f16=False)
        os.replace(tmp, dst)
    return dst
