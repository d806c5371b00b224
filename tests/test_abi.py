"""CPU tests of the drop-in boundary: libbark.so loads without a GPU, exports every symbol the headers
declare, keeps the reference's by-value parameter struct layout, and (when /root/reference is present)
compiles and links the reference's own example caller unmodified."""
import ctypes as C
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
REF = "/root/reference"


@pytest.fixture(scope="module")
def pkg():
    from bark_amd_loader import load_package
    p = load_package()
    if not os.path.exists(p.library_path()):
        p.build_library()
    return p


def _declared_symbols():
    names = set()
    for fn in ("bark.h", "bark_mi355x.h"):
        text = open(os.path.join(INC, fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#"))
        for m in re.finditer(r"BARK_API\s+[^;(]*?\b(\w+)\s*\(", text):
            names.add(m.group(1))
    text = open(os.path.join(INC, "ggml.h")).read()
    for m in re.finditer(r'visibility\("default"\)\)\)\s+[^;(]*?\b(\w+)\s*\(', text):
        names.add(m.group(1))
    return names


def test_every_declared_symbol_is_exported(pkg):
    lib = pkg.load_library()
    declared = _declared_symbols()
    assert {"bark_load_model", "bark_generate_audio", "bark_get_audio_data", "bark_hip_gpt_eval", "ggml_time_us"} <= declared
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    assert set(pkg.api.EXPORTS) <= declared | {"bark_hip_codec_tap"}


def test_param_struct_layout_matches_reference_header(pkg):
    # 23 four-byte fields, then two pointers (bark.h:81-141); x86-64: 4 bytes of padding before the pointers
    P = pkg.BarkContextParams
    assert C.sizeof(P) == 112
    assert P.progress_callback.offset == 96 and P.progress_callback_user_data.offset == 104
    assert P.temp.offset == 4 and P.codebook_size.offset == 88


def test_default_params_are_the_reference_defaults(pkg):
    p = pkg.default_params()          # bark.cpp:2202-2232
    got = {k: getattr(p, k) for k, _ in p._fields_[:-2]}
    want = dict(verbosity=0, temp=0.7, fine_temp=0.5, min_eos_p=0.2, sliding_window_size=60, max_coarse_history=630,
                sample_rate=24000, target_bandwidth=6, cls_token_id=101, sep_token_id=102, n_steps_text_encoder=768,
                text_pad_token=129595, text_encoding_offset=10048, semantic_rate_hz=49.9, semantic_pad_token=10000,
                semantic_vocab_size=10000, semantic_infer_token=129599, coarse_rate_hz=75.0, coarse_infer_token=12050,
                coarse_semantic_pad_token=12048, n_coarse_codebooks=2, n_fine_codebooks=8, codebook_size=1024)
    for k, v in want.items():
        assert got[k] == pytest.approx(v, rel=1e-6), k


def test_null_safe_getters_and_failures(pkg):
    lib = pkg.load_library()
    assert lib.bark_get_audio_data_size(None) == 0
    assert lib.bark_get_load_time(None) == 0 and lib.bark_get_eval_time(None) == 0
    lib.bark_free(None)
    lib.bark_reset_statistics(None)
    assert not lib.bark_generate_audio(None, b"x", 1)
    assert not lib.bark_model_quantize(b"a", b"b", 2)
    assert lib.ggml_time_us() > 0
    # no GPU in the build container / wrong path on the GPU box: the loader must fail loudly, never fall back
    assert not lib.bark_load_model(b"/nonexistent/model.bin", pkg.default_params(), 0)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "examples", "main")), reason="reference checkout not present")
def test_reference_example_main_compiles_and_links_unmodified(pkg, tmp_path):
    """examples/main/main.cpp + examples/common.cpp are the reference's own callers of bark.h."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "main"
    cmd = [hipcc, "-std=c++17", "-O1", "-I", INC, "-I", os.path.join(REF, "examples"),
           os.path.join(REF, "examples", "main", "main.cpp"), os.path.join(REF, "examples", "common.cpp"),
           "-L", os.path.dirname(pkg.library_path()), "-lbark", "-Wl,-rpath," + os.path.dirname(pkg.library_path()), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    h = subprocess.run([str(exe), "-h"], capture_output=True, text=True)
    assert "usage" in (h.stdout + h.stderr).lower()


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "examples", "quantize")), reason="reference checkout not present")
def test_reference_example_quantize_runs_unmodified(pkg, toy_model, tmp_path):
    """examples/quantize/main.cpp drives bark_model_quantize (bark.h:229-232); all five of its types must be written."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "quantize"
    cmd = [hipcc, "-std=c++17", "-O1", "-I", INC, os.path.join(REF, "examples", "quantize", "main.cpp"),
           "-L", os.path.dirname(pkg.library_path()), "-lbark", "-Wl,-rpath," + os.path.dirname(pkg.library_path()), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    from tests.test_quantize import FORMATS, parse_quantized
    for fmt, (ftype, ttype, _) in FORMATS.items():
        dst = tmp_path / f"toy_{fmt}.bin"
        q = subprocess.run([str(exe), toy_model, str(dst), fmt], capture_output=True, text=True)
        assert q.returncode == 0, q.stderr[-2000:]
        secs, _ = parse_quantized(str(dst))
        assert secs[0][0][9] == 2000 + ftype
        assert secs[2][1]["model/h0/mlp/c_fc/w"][0] == ttype
