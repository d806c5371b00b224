"""GPU tests of the loader's validation (engine_load.hip): a model file whose container parses but whose shapes / hparams
do not fit the kernels must be refused by bark_load_model (nullptr + stderr, reference behaviour bark.cpp:1174-1177) -
never crash the process or reach a kernel."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import model_patch as mp      # noqa: E402


def _pkg():
    from bark_amd_loader import load_package
    return load_package()


def _load_fails(path):
    pkg = _pkg()
    with pytest.raises(RuntimeError):
        pkg.BarkContext.load_model(path, pkg.default_params(temp=0.0, fine_temp=0.0), seed=0)


@pytest.fixture(scope="module")
def toy_bytes(toy_model):
    return open(toy_model, "rb").read()


def _damaged(tmp_path, toy_bytes, name, edit):
    buf = bytearray(toy_bytes)
    edit(buf, mp.walk(buf))
    p = tmp_path / name
    p.write_bytes(buf)
    return str(p)


def test_fine_model_with_six_heads_is_refused(tmp_path, toy_bytes):
    # the fine stage indexes lm_heads[nn - 1] for nn up to 7 (bark.cpp:1573): 6 heads would dereference a missing one
    _load_fails(_damaged(tmp_path, toy_bytes, "six_heads.bin", lambda b, w: mp.poke_i32(b, w["gpt"][2]["hp_off"]["n_lm_heads"], 6)))


def test_vocabulary_beyond_the_sampler_is_refused(tmp_path, toy_bytes):
    _load_fails(_damaged(tmp_path, toy_bytes, "big_vocab.bin", lambda b, w: mp.poke_i32(b, w["gpt"][0]["hp_off"]["n_out_vocab"], 20000)))
    _load_fails(_damaged(tmp_path, toy_bytes, "neg_vocab.bin", lambda b, w: mp.poke_i32(b, w["gpt"][1]["hp_off"]["n_in_vocab"], -5)))


def test_embedding_width_without_a_kernel_is_refused(tmp_path, toy_bytes):
    # 384 is a multiple of 128 but no decode GEMV is instantiated for it; the loader must say so instead of aborting later
    _load_fails(_damaged(tmp_path, toy_bytes, "e384.bin", lambda b, w: mp.poke_i32(b, w["gpt"][0]["hp_off"]["n_embd"], 384)))


def test_codec_convolution_that_does_not_chain_is_refused(tmp_path, toy_bytes):
    def edit(b, w):
        r = w["codec"]["decoder.model.0.conv.conv.weight"]
        k, cin, cout = r["dims"]
        mp.poke_i32(b, r["dims_off"] + 4, cin // 2)          # same element count: the container still parses
        mp.poke_i32(b, r["dims_off"] + 8, cout * 2)
    _load_fails(_damaged(tmp_path, toy_bytes, "codec_cin.bin", edit))


def test_vector_with_a_matrix_shape_is_refused(tmp_path, toy_bytes):
    def edit(b, w):
        r = w["gpt"][0]["tensors"]["model/wpe"]              # [E, block] -> [E / 2, 2 * block]: ne[0] no longer n_embd
        e, blk = r["dims"]
        mp.poke_i32(b, r["dims_off"], e // 2)
        mp.poke_i32(b, r["dims_off"] + 4, blk * 2)
    _load_fails(_damaged(tmp_path, toy_bytes, "wpe.bin", edit))


def test_coarse_window_outside_the_vocabulary_fails_cleanly(toy_model):
    # semantic_vocab_size moves the coarse logit window (bark.cpp:1829-1835); a window past the LM head must fail the call,
    # single and lock-step paths alike, instead of reading rows out of bounds
    pkg = _pkg()
    ctx = pkg.BarkContext.load_model(toy_model, pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=8, semantic_vocab_size=11000), seed=0)
    try:
        with pytest.raises(RuntimeError):
            ctx.coarse(np.arange(8, dtype=np.int32))
        assert not ctx.generate_audio("hello world")
        res = ctx.generate_batch(["hello world", "the river runs fast"])
        assert all(r is None for r in res)
    except RuntimeError:
        pass      # generate_batch may report the failure as an error code
    finally:
        ctx.free()
