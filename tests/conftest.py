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
