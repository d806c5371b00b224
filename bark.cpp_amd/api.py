"""ctypes mirror of libbark.so (include/bark.h + include/bark_mi355x.h).

Names, argument meaning and error behaviour follow the reference C API
(/root/reference/bark.h:148-240): `BarkContext.load_model` <-> bark_load_model,
`generate_audio` <-> bark_generate_audio, `audio_data` <-> bark_get_audio_data[_size], ...
There is NO fallback: if the HIP library is missing or no GPU is present, loading raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

PROGRESS_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_void_p)


class BarkContextParams(C.Structure):
    """struct bark_context_params (bark.h:81-141) - field order is ABI."""
    _fields_ = [
        ("verbosity", C.c_int), ("temp", C.c_float), ("fine_temp", C.c_float), ("min_eos_p", C.c_float),
        ("sliding_window_size", C.c_int32), ("max_coarse_history", C.c_int32), ("sample_rate", C.c_int32),
        ("target_bandwidth", C.c_int32), ("cls_token_id", C.c_int32), ("sep_token_id", C.c_int32),
        ("n_steps_text_encoder", C.c_int32), ("text_pad_token", C.c_int32), ("text_encoding_offset", C.c_int32),
        ("semantic_rate_hz", C.c_float), ("semantic_pad_token", C.c_int32), ("semantic_vocab_size", C.c_int32),
        ("semantic_infer_token", C.c_int32), ("coarse_rate_hz", C.c_float), ("coarse_infer_token", C.c_int32),
        ("coarse_semantic_pad_token", C.c_int32), ("n_coarse_codebooks", C.c_int32), ("n_fine_codebooks", C.c_int32),
        ("codebook_size", C.c_int32), ("progress_callback", PROGRESS_CB), ("progress_callback_user_data", C.c_void_p),
    ]


class BarkHipRequestParams(C.Structure):
    """struct bark_hip_request_params (bark_mi355x.h): what an utterance of a lock-step job may set for itself."""
    _fields_ = [("temp", C.c_float), ("fine_temp", C.c_float), ("min_eos_p", C.c_float), ("n_steps_text_encoder", C.c_int32), ("seed", C.c_uint32)]


class BarkHipStats(C.Structure):
    _fields_ = [
        ("t_load_us", C.c_int64), ("t_eval_us", C.c_int64), ("t_semantic_us", C.c_int64), ("t_coarse_us", C.c_int64),
        ("t_fine_us", C.c_int64), ("t_codec_us", C.c_int64), ("n_sample_semantic", C.c_int64), ("n_sample_coarse", C.c_int64),
        ("n_sample_fine", C.c_int64), ("n_semantic", C.c_int32), ("n_frames", C.c_int32), ("n_samples", C.c_int32),
        ("n_near_tie", C.c_int32), ("graph_replays", C.c_int32), ("n_prefix_rows_reused", C.c_int32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def library_path() -> str:
    # BARK_HIP_LIBRARY: another build of the same library (e.g. a host-AddressSanitizer build used while debugging)
    return os.environ.get("BARK_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libbark.so")


def build_library(force: bool = False) -> str:
    """Compile csrc/ for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["rm", "-rf", os.path.join(_HERE, "lib", "obj")])
    subprocess.check_call([os.path.join(_HERE, "build.sh")])
    return library_path()


_LIB = None

EXPORTS = [
    # bark.h
    "bark_context_default_params", "bark_load_model", "bark_generate_audio", "bark_get_audio_data", "bark_get_audio_data_size",
    "bark_get_load_time", "bark_get_eval_time", "bark_reset_statistics", "bark_model_quantize", "bark_free",
    # ggml.h shim
    "ggml_time_init", "ggml_time_us", "ggml_time_ms", "ggml_init", "ggml_free",
    # bark_mi355x.h
    "bark_hip_hparams", "bark_hip_set_params", "bark_hip_tokenize", "bark_hip_bert_tokenize", "bark_hip_gpt_eval",
    "bark_hip_fine_eval", "bark_hip_semantic", "bark_hip_coarse", "bark_hip_fine", "bark_hip_fine_many", "bark_hip_codec_decode", "bark_hip_codec_tap",
    "bark_hip_clone_context", "bark_hip_generate_audio_batch", "bark_hip_generate_batch", "bark_hip_generate_batch_seeded", "bark_hip_generate_batch_ex", "bark_hip_reserve_batch", "bark_hip_profile_lock_step", "bark_hip_batch_audio", "bark_hip_batch_tokens", "bark_hip_get_semantic_tokens", "bark_hip_get_coarse_tokens", "bark_hip_get_fine_tokens", "bark_hip_get_stats",
    "bark_hip_time_decode_step", "bark_hip_time_gemv", "bark_hip_time_slots", "bark_hip_time_fine_pass", "bark_hip_time_fine_passes", "bark_hip_describe", "bark_hip_set_fine_order", "bark_hip_load_model_on_device", "bark_hip_batcher_create_multi",
    "bark_hip_batcher_create", "bark_hip_batcher_create_ex", "bark_hip_batcher_submit", "bark_hip_batcher_submit_ex", "bark_hip_batcher_wait", "bark_hip_batcher_stats", "bark_hip_batcher_admitted", "bark_hip_batcher_free",
]


def load_library() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing - run bark.cpp_amd/build.sh (there is no fallback path)")
    lib = C.CDLL(path)
    vp, ip, fp = C.c_void_p, C.c_void_p, C.c_void_p
    lib.bark_context_default_params.restype = BarkContextParams
    lib.bark_context_default_params.argtypes = []
    lib.bark_load_model.restype = vp
    lib.bark_load_model.argtypes = [C.c_char_p, BarkContextParams, C.c_uint32]
    lib.bark_generate_audio.restype = C.c_bool
    lib.bark_generate_audio.argtypes = [vp, C.c_char_p, C.c_int]
    lib.bark_get_audio_data.restype = C.POINTER(C.c_float)
    lib.bark_get_audio_data.argtypes = [vp]
    lib.bark_get_audio_data_size.restype = C.c_int
    lib.bark_get_audio_data_size.argtypes = [vp]
    lib.bark_get_load_time.restype = C.c_int64
    lib.bark_get_load_time.argtypes = [vp]
    lib.bark_get_eval_time.restype = C.c_int64
    lib.bark_get_eval_time.argtypes = [vp]
    lib.bark_reset_statistics.argtypes = [vp]
    lib.bark_model_quantize.restype = C.c_bool
    lib.bark_model_quantize.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    lib.bark_free.argtypes = [vp]
    lib.ggml_time_us.restype = C.c_int64
    lib.bark_hip_hparams.argtypes = [vp, C.c_int, ip]
    lib.bark_hip_set_params.argtypes = [vp, BarkContextParams]
    lib.bark_hip_tokenize.argtypes = [vp, C.c_char_p, ip]
    lib.bark_hip_bert_tokenize.argtypes = [vp, C.c_char_p, ip, C.c_int]
    lib.bark_hip_gpt_eval.argtypes = [vp, C.c_int, ip, C.c_int, C.c_int, C.c_int, fp]
    lib.bark_hip_fine_eval.argtypes = [vp, ip, C.c_int, fp]
    lib.bark_hip_semantic.argtypes = [vp, ip, ip, C.c_int, fp]
    lib.bark_hip_coarse.argtypes = [vp, ip, C.c_int, ip, C.c_int]
    lib.bark_hip_fine.argtypes = [vp, ip, C.c_int, ip, C.c_int]
    lib.bark_hip_fine_many.argtypes = [vp, ip, ip, C.c_int, ip, C.c_int]
    lib.bark_hip_codec_decode.argtypes = [vp, ip, C.c_int, C.c_int, fp, C.c_int]
    lib.bark_hip_codec_tap.argtypes = [vp, ip, C.c_int, C.c_int, C.c_int, fp, C.c_int]
    lib.bark_hip_generate_batch.argtypes = [vp, C.POINTER(C.c_char_p), C.c_int]
    lib.bark_hip_generate_batch_seeded.argtypes = [vp, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_uint32)]
    lib.bark_hip_generate_batch_ex.argtypes = [vp, C.POINTER(C.c_char_p), C.c_int, C.POINTER(BarkHipRequestParams)]
    lib.bark_hip_reserve_batch.argtypes = [vp, C.c_int]
    lib.bark_hip_set_fine_order.argtypes = [vp, C.c_int]
    lib.bark_hip_profile_lock_step.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
    lib.bark_hip_batch_audio.argtypes = [vp, C.c_int, C.POINTER(C.POINTER(C.c_float))]
    lib.bark_hip_batch_tokens.argtypes = [vp, C.c_int, C.c_int, ip, C.c_int]
    lib.bark_hip_clone_context.restype = vp
    lib.bark_hip_clone_context.argtypes = [vp, C.c_uint32]
    lib.bark_hip_generate_audio_batch.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_char_p), C.c_int]
    lib.bark_hip_get_semantic_tokens.argtypes = [vp, ip, C.c_int]
    lib.bark_hip_get_coarse_tokens.argtypes = [vp, ip, C.c_int]
    lib.bark_hip_get_fine_tokens.argtypes = [vp, ip, C.c_int]
    lib.bark_hip_get_stats.argtypes = [vp, C.POINTER(BarkHipStats)]
    lib.bark_hip_time_decode_step.restype = C.c_double
    lib.bark_hip_time_decode_step.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    lib.bark_hip_time_gemv.restype = C.c_double
    lib.bark_hip_time_gemv.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    lib.bark_hip_time_slots.restype = C.c_double
    lib.bark_hip_time_slots.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.bark_hip_time_fine_pass.restype = C.c_double
    lib.bark_hip_time_fine_pass.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    lib.bark_hip_batcher_create.restype = vp
    lib.bark_hip_batcher_create.argtypes = [vp, C.c_int, C.c_int]
    lib.bark_hip_batcher_create_ex.restype = vp
    lib.bark_hip_batcher_create_ex.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.bark_hip_load_model_on_device.restype = vp
    lib.bark_hip_load_model_on_device.argtypes = [C.c_char_p, BarkContextParams, C.c_uint32, C.c_int]
    lib.bark_hip_batcher_create_multi.restype = vp
    lib.bark_hip_batcher_create_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]
    lib.bark_hip_batcher_submit.restype = C.c_int64
    lib.bark_hip_batcher_submit.argtypes = [vp, C.c_char_p, C.c_uint32]
    lib.bark_hip_batcher_submit_ex.restype = C.c_int64
    lib.bark_hip_batcher_submit_ex.argtypes = [vp, C.c_char_p, C.POINTER(BarkHipRequestParams)]
    lib.bark_hip_batcher_wait.argtypes = [vp, C.c_int64, fp, C.c_int]
    lib.bark_hip_batcher_stats.restype = None
    lib.bark_hip_batcher_stats.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.bark_hip_batcher_admitted.argtypes = [vp]
    lib.bark_hip_batcher_free.restype = None
    lib.bark_hip_batcher_free.argtypes = [vp]
    lib.bark_hip_time_fine_passes.restype = C.c_double
    lib.bark_hip_time_fine_passes.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double)]
    lib.bark_hip_describe.restype = C.c_char_p
    lib.bark_hip_describe.argtypes = [vp]
    _LIB = lib
    return lib


def default_params(**overrides) -> BarkContextParams:
    p = load_library().bark_context_default_params()
    for k, v in overrides.items():
        setattr(p, k, v)
    return p


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class BarkContext:
    """Owner of a `struct bark_context *`."""

    def __init__(self, handle, lib):
        self._h = handle
        self._lib = lib
        self._cb = None
        self._params = None

    # ---- bark.h ---------------------------------------------------------------------------------
    @classmethod
    def load_model(cls, model_path: str, params: BarkContextParams | None = None, seed: int = 0, device: int | None = None) -> "BarkContext":
        """device None: bark_load_model (BARK_HIP_DEVICE / the current device); an ordinal: bark_hip_load_model_on_device (one process, several GPUs)"""
        lib = load_library()
        params = params if params is not None else default_params()
        if device is None:
            h = lib.bark_load_model(os.fsencode(model_path), params, seed)
        else:
            h = lib.bark_hip_load_model_on_device(os.fsencode(model_path), params, seed, int(device))
        if not h:
            raise RuntimeError(f"bark_load_model failed for {model_path}")
        ctx = cls(h, lib)
        ctx._cb = params.progress_callback      # keep the callback object alive
        ctx._params = params
        return ctx

    def generate_audio(self, text: str, n_threads: int = 4) -> bool:
        return bool(self._lib.bark_generate_audio(self._h, text.encode("utf-8"), n_threads))

    def audio_data(self) -> np.ndarray:
        n = self._lib.bark_get_audio_data_size(self._h)
        p = self._lib.bark_get_audio_data(self._h)
        if n <= 0 or not p:
            return np.zeros(0, np.float32)
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def load_time_us(self) -> int:
        return int(self._lib.bark_get_load_time(self._h))

    def eval_time_us(self) -> int:
        return int(self._lib.bark_get_eval_time(self._h))

    def reset_statistics(self):
        self._lib.bark_reset_statistics(self._h)

    def free(self):
        if self._h:
            self._lib.bark_free(self._h)
            self._h = None

    close = free

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    # ---- bark_mi355x.h --------------------------------------------------------------------------
    def describe(self) -> str:
        return self._lib.bark_hip_describe(self._h).decode()

    def hparams(self, which: int) -> dict:
        out = np.zeros(10, np.int32)
        if self._lib.bark_hip_hparams(self._h, which, out.ctypes.data) != 0:
            raise RuntimeError("bark_hip_hparams failed")
        keys = ["n_layer", "n_head", "n_embd", "block_size", "bias", "n_in", "n_out", "n_lm_heads", "n_wtes", "ftype"]
        return dict(zip(keys, (int(v) for v in out)))

    def set_params(self, params: BarkContextParams):
        self._cb = params.progress_callback
        self._params = params
        self._lib.bark_hip_set_params(self._h, params)

    def tokenize(self, text: str) -> np.ndarray:
        out = np.zeros(513, np.int32)
        n = self._lib.bark_hip_tokenize(self._h, text.encode("utf-8"), out.ctypes.data)
        if n != 513:
            raise RuntimeError("bark_hip_tokenize failed")
        return out

    def bert_tokenize(self, text: str, n_max: int = 256) -> np.ndarray:
        out = np.zeros(n_max, np.int32)
        n = self._lib.bark_hip_bert_tokenize(self._h, text.encode("utf-8"), out.ctypes.data, n_max)
        if n < 0:
            raise RuntimeError("bark_hip_bert_tokenize failed")
        return out[:n]

    def gpt_eval(self, which: int, tokens, n_past: int, merge_ctx: bool):
        tokens = _i32(tokens)
        logits = np.zeros(self.hparams(which)["n_out"], np.float32)
        r = self._lib.bark_hip_gpt_eval(self._h, which, tokens.ctypes.data, len(tokens), n_past, int(merge_ctx), logits.ctypes.data)
        if r < 0:
            raise RuntimeError("bark_hip_gpt_eval failed")
        return logits, r

    def fine_eval(self, tokens_8x1024, nn: int) -> np.ndarray:
        tokens = _i32(tokens_8x1024).reshape(8, 1024)
        logits = np.zeros((1024, self.hparams(2)["n_out"]), np.float32)
        if self._lib.bark_hip_fine_eval(self._h, tokens.ctypes.data, nn, logits.ctypes.data) != 0:
            raise RuntimeError("bark_hip_fine_eval failed")
        return logits

    def semantic(self, prompt513, want_eos_trace: bool = False):
        prompt = _i32(prompt513)
        assert prompt.shape == (513,)
        out = np.zeros(1024, np.int32)
        tr = np.zeros(1025, np.float32)
        n = self._lib.bark_hip_semantic(self._h, prompt.ctypes.data, out.ctypes.data, 1024, tr.ctypes.data if want_eos_trace else None)
        if n < 0:
            raise RuntimeError("bark_hip_semantic failed")
        return (out[:n].copy(), tr) if want_eos_trace else out[:n].copy()

    def coarse(self, semantic) -> np.ndarray:
        sem = _i32(semantic)
        p = self._params if self._params is not None else default_params()      # T = floor(n_sem * coarse_rate / semantic_rate) with the context's own rates
        cap = int(len(sem) * max(p.coarse_rate_hz, 1e-3) / max(p.semantic_rate_hz, 1e-3)) + 8
        out = np.zeros((cap, 2), np.int32)
        T = self._lib.bark_hip_coarse(self._h, sem.ctypes.data, len(sem), out.ctypes.data, cap)
        if T < 0:
            raise RuntimeError("bark_hip_coarse failed")
        return out[:T].copy()

    def fine(self, coarse_Tx2) -> np.ndarray:
        co = _i32(coarse_Tx2).reshape(-1, 2)
        out = np.zeros((max(len(co), 1), 8), np.int32)
        T = self._lib.bark_hip_fine(self._h, co.ctypes.data, len(co), out.ctypes.data, len(out))
        if T < 0:
            raise RuntimeError("bark_hip_fine failed")
        return out[:T].copy()

    def fine_many(self, coarse_list) -> list:
        """The fine stage of several utterances, their windows side by side in every forward pass (bark_hip_fine_many)."""
        cos = [_i32(x).reshape(-1, 2) for x in coarse_list]
        T = _i32([len(x) for x in cos])
        cat = np.ascontiguousarray(np.concatenate(cos, axis=0))
        out = np.zeros((len(cat), 8), np.int32)
        n = self._lib.bark_hip_fine_many(self._h, cat.ctypes.data, T.ctypes.data, len(cos), out.ctypes.data, len(out))
        if n < 0:
            raise RuntimeError("bark_hip_fine_many failed")
        res, off = [], 0
        for t in T:
            res.append(out[off:off + int(t)].copy()); off += int(t)
        return res

    def codec_decode(self, codes_qxT) -> np.ndarray:
        codes = _i32(codes_qxT)
        n_q, T = codes.shape
        pcm = np.zeros(T * 320, np.float32)
        n = self._lib.bark_hip_codec_decode(self._h, codes.ctypes.data, n_q, T, pcm.ctypes.data, pcm.size)
        if n < 0:
            raise RuntimeError("bark_hip_codec_decode failed")
        return pcm[:n].copy()

    def codec_tap(self, codes_qxT, stage: int) -> np.ndarray:
        codes = _i32(codes_qxT)
        n_q, T = codes.shape
        out = np.zeros(T * 320 * 64, np.float32)
        n = self._lib.bark_hip_codec_tap(self._h, codes.ctypes.data, n_q, T, stage, out.ctypes.data, out.size)
        if n < 0:
            raise RuntimeError("bark_hip_codec_tap failed")
        return out[:n].copy()

    def request_params(self, **over) -> BarkHipRequestParams:
        """The context's own values of the per-utterance parameters, with overrides (temp, fine_temp, min_eos_p, n_steps_text_encoder, seed)."""
        p = self._params if self._params is not None else default_params()
        r = BarkHipRequestParams(p.temp, p.fine_temp, p.min_eos_p, p.n_steps_text_encoder, 0)
        known = {name for name, _ in BarkHipRequestParams._fields_}
        for k, v in over.items():
            if k not in known:             # setattr on a ctypes.Structure accepts any name silently: a typo would run with the context's value
                raise TypeError(f"request_params: unknown parameter {k!r} (one of {sorted(known)})")
            setattr(r, k, v)
        return r

    def reserve_batch(self, slots: int):
        if self._lib.bark_hip_reserve_batch(self._h, slots) != 0:
            raise RuntimeError("bark_hip_reserve_batch failed")

    def generate_batch(self, texts, seeds=None, params=None) -> list:
        """In-engine batching (bark_hip_generate_batch[_seeded|_ex]): returns one dict per utterance (or None if it failed).
        params: one BarkHipRequestParams per utterance (request_params(...))."""
        n = len(texts)
        ts = (C.c_char_p * n)(*[t.encode("utf-8") for t in texts])
        if params is not None:
            assert len(params) == n and seeds is None
            good = self._lib.bark_hip_generate_batch_ex(self._h, ts, n, (BarkHipRequestParams * n)(*params))
        elif seeds is None:
            good = self._lib.bark_hip_generate_batch(self._h, ts, n)
        else:
            assert len(seeds) == n
            good = self._lib.bark_hip_generate_batch_seeded(self._h, ts, n, (C.c_uint32 * n)(*[int(v) for v in seeds]))
        if good < 0:
            raise RuntimeError("bark_hip_generate_batch failed")
        out = []
        for i in range(n):
            p = C.POINTER(C.c_float)()
            ns = self._lib.bark_hip_batch_audio(self._h, i, C.byref(p))
            if ns < 0:
                out.append(None)
                continue
            d = {"pcm": np.ctypeslib.as_array(p, shape=(ns,)).copy() if ns else np.zeros(0, np.float32)}
            for stage, (name, w) in enumerate((("semantic", 1), ("coarse", 2), ("fine", 8))):
                buf = np.zeros(65536, np.int32)
                k = self._lib.bark_hip_batch_tokens(self._h, i, stage, buf.ctypes.data, buf.size)
                d[name] = buf[:max(k, 0)].copy().reshape(-1, w) if w > 1 else buf[:max(k, 0)].copy()
            out.append(d)
        return out

    def clone(self, seed: int = 0) -> "BarkContext":
        h = self._lib.bark_hip_clone_context(self._h, seed)
        if not h:
            raise RuntimeError("bark_hip_clone_context failed")
        ctx = BarkContext(h, self._lib)
        ctx._cb = self._cb                              # the clone copies the parameters (and the callback pointer) of its source
        ctx._params = self._params
        return ctx

    @staticmethod
    def generate_audio_batch(ctxs, texts) -> int:
        lib = load_library()
        n = len(ctxs)
        assert n == len(texts) and n > 0
        hs = (C.c_void_p * n)(*[c._h for c in ctxs])
        ts = (C.c_char_p * n)(*[t.encode("utf-8") for t in texts])
        return int(lib.bark_hip_generate_audio_batch(hs, ts, n))

    def semantic_tokens(self) -> np.ndarray:
        out = np.zeros(1024, np.int32)
        n = self._lib.bark_hip_get_semantic_tokens(self._h, out.ctypes.data, 1024)
        return out[:max(n, 0)].copy()

    def coarse_tokens(self) -> np.ndarray:
        out = np.zeros((8192, 2), np.int32)            # engine_fine accepts up to 8192 frames
        n = self._lib.bark_hip_get_coarse_tokens(self._h, out.ctypes.data, 8192)
        return out[:max(n, 0)].copy()

    def fine_tokens(self) -> np.ndarray:
        out = np.zeros((8192, 8), np.int32)
        n = self._lib.bark_hip_get_fine_tokens(self._h, out.ctypes.data, 8192)
        return out[:max(n, 0)].copy()

    def stats(self) -> dict:
        s = BarkHipStats()
        self._lib.bark_hip_get_stats(self._h, C.byref(s))
        return s.as_dict()

    def set_fine_order(self, order: int):
        """0: default policy (C1 for generate_audio / stage calls, C1m inside lock-step jobs and fine_many); 1: C1 everywhere; 2: C1m everywhere."""
        if self._lib.bark_hip_set_fine_order(self._h, int(order)) != 0:
            raise RuntimeError("bark_hip_set_fine_order failed")

    def time_decode_step(self, which: int, ctx: int, iters: int):
        b = C.c_double(0)
        us = self._lib.bark_hip_time_decode_step(self._h, which, ctx, iters, C.byref(b))
        if us < 0:
            raise RuntimeError("bark_hip_time_decode_step failed")
        return us, b.value

    def time_gemv(self, which: int, op: int, iters: int):
        b = C.c_double(0)
        us = self._lib.bark_hip_time_gemv(self._h, which, op, iters, C.byref(b))
        if us < 0:
            raise RuntimeError("bark_hip_time_gemv failed")
        return us, b.value

    def time_slots(self, which: int, op: int, n_slots: int, kind: int, ctx: int, iters: int) -> float:
        us = self._lib.bark_hip_time_slots(self._h, which, op, n_slots, kind, ctx, iters)
        if us < 0:
            raise RuntimeError("bark_hip_time_slots failed")
        return us

    def profile_lock_step(self, which: int, n_slots: int, ctx: int, reps: int = 20) -> list:
        """[{"site", "us"}] per launch site of one lock step in launch order (kernel + the gap in front), closed by the graph-replayed step."""
        import json
        buf = C.create_string_buffer(1 << 16)
        n = self._lib.bark_hip_profile_lock_step(self._h, which, n_slots, ctx, reps, buf, len(buf))
        if n < 0:
            raise RuntimeError("bark_hip_profile_lock_step failed")
        return json.loads(buf.value.decode())

    def time_fine_pass(self, iters: int, n_windows: int = 1):
        """us per forward pass of the fine model over n_windows windows side by side, flops of that pass"""
        f = C.c_double(0)
        if n_windows > 1:
            us = self._lib.bark_hip_time_fine_passes(self._h, n_windows, iters, C.byref(f))
        else:
            us = self._lib.bark_hip_time_fine_pass(self._h, iters, C.byref(f))
        if us < 0:
            raise RuntimeError("bark_hip_time_fine_pass failed")
        return us, f.value


class Batcher:
    """bark_hip_batcher: thread-safe submit / wait in front of the context's lock-step batches (the context is owned by the batcher's
    worker thread while it lives)."""

    def __init__(self, ctx, max_batch: int = 32, max_wait_ms: int = 2, streams: int = 1):
        """ctx: one BarkContext (streams > 1: further workers on clones of it, same GPU), or a list of contexts - one worker each, typically one per GPU
        (bark_hip_batcher_create_multi)."""
        ctxs = list(ctx) if isinstance(ctx, (list, tuple)) else [ctx]
        self._lib = ctxs[0]._lib
        self._ctx = ctxs                                # the worker threads run on these contexts: they must outlive the batcher
        if len(ctxs) > 1:
            arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
            self._b = self._lib.bark_hip_batcher_create_multi(arr, len(ctxs), max_batch, max_wait_ms)
        else:
            self._b = self._lib.bark_hip_batcher_create_ex(ctxs[0]._h, max_batch, max_wait_ms, streams)     # streams > 1: further workers on clones of ctx
        if not self._b:
            raise RuntimeError("bark_hip_batcher_create failed")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def submit(self, text: str, seed: int = 0, params: "BarkHipRequestParams | None" = None) -> int:
        if params is not None:
            t = self._lib.bark_hip_batcher_submit_ex(self._b, text.encode("utf-8"), C.byref(params))
        else:
            t = self._lib.bark_hip_batcher_submit(self._b, text.encode("utf-8"), seed)
        if t <= 0:
            raise RuntimeError("bark_hip_batcher_submit failed")
        return t

    def wait(self, ticket: int) -> np.ndarray:
        n = self._lib.bark_hip_batcher_wait(self._b, ticket, None, 0)          # probe: -(2 + samples)
        if n > -2:
            raise RuntimeError("generation failed")
        pcm = np.zeros(-n - 2, np.float32)
        n = self._lib.bark_hip_batcher_wait(self._b, ticket, pcm.ctypes.data, pcm.size)
        if n < 0:
            raise RuntimeError("bark_hip_batcher_wait failed")
        return pcm[:n]

    def stats(self) -> dict:
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        self._lib.bark_hip_batcher_stats(self._b, C.byref(a), C.byref(b), C.byref(c))
        return {"n_batches": a.value, "n_requests": b.value, "largest_batch": c.value, "n_admitted": int(self._lib.bark_hip_batcher_admitted(self._b))}

    def free(self):
        if self._b:
            self._lib.bark_hip_batcher_free(self._b)
            self._b = None
        self._ctx = None
