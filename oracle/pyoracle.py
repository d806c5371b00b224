"""ctypes binding of the CPU oracle (oracle/bark_oracle.cpp).

TEST INFRASTRUCTURE.  Imported only by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py — never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "build", "libbark_oracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("bark_oracle.cpp", "mfma_f16_emu.h")]
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return LIB_PATH


class OrcParams(C.Structure):
    _fields_ = [("temp", C.c_float), ("fine_temp", C.c_float), ("min_eos_p", C.c_float),
                ("sliding_window_size", C.c_int32), ("max_coarse_history", C.c_int32), ("n_steps_text_encoder", C.c_int32)]


class OrcResult(C.Structure):
    _fields_ = [("n_semantic", C.c_int32), ("n_frames", C.c_int32), ("n_samples", C.c_int32),
                ("t_eval_us", C.c_int64), ("t_semantic_us", C.c_int64), ("t_coarse_us", C.c_int64), ("t_fine_us", C.c_int64),
                ("t_codec_us", C.c_int64),
                ("t_predict_semantic_us", C.c_int64), ("t_predict_coarse_us", C.c_int64), ("t_predict_fine_us", C.c_int64),
                ("n_sample_semantic", C.c_int64), ("n_sample_coarse", C.c_int64), ("n_sample_fine", C.c_int64)]


# Which order the fine model's weight products follow when an Oracle instance has not been told explicitly (set_fine_mfma): False = C1, the
# restatement of the reference's arithmetic (what bark_generate_audio and the stage-level entry points of the engine compute); True = C1m, the f16
# matrix cores' accumulation order that lock-step jobs run their fine passes in.  tests/conftest.py switches it for tests marked `lock_step_job`.
JOB_ORDER = False


class job_order:
    """with pyoracle.job_order(): ... - oracle calls inside compute what a lock-step job of the engine computes (fine products in C1m)."""
    def __init__(self, on: bool = True):
        self.on = on

    def __enter__(self):
        global JOB_ORDER
        self.prev, JOB_ORDER = JOB_ORDER, self.on
        return self

    def __exit__(self, *exc):
        global JOB_ORDER
        JOB_ORDER = self.prev


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Oracle:
    def __init__(self, model_path: str, n_threads: int = 4):
        build()
        self.lib = lib = C.CDLL(LIB_PATH)
        lib.orc_open.restype = C.c_void_p
        lib.orc_open.argtypes = [C.c_char_p]
        lib.orc_close.argtypes = [C.c_void_p]
        lib.orc_set_numerics.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.orc_set_dot_order.argtypes = [C.c_void_p, C.c_int]
        lib.orc_set_fine_mfma.argtypes = [C.c_void_p, C.c_int]
        lib.orc_set_codec_mfma.argtypes = [C.c_void_p, C.c_int]
        lib.orc_seed.argtypes = [C.c_void_p, C.c_uint32]
        lib.orc_gelu_table.restype = C.POINTER(C.c_uint16)
        lib.orc_gelu_table.argtypes = [C.c_void_p]
        lib.orc_hparams.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.orc_tokenize.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        lib.orc_bert_tokenize.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        lib.orc_gpt_eval.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        lib.orc_fine_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        lib.orc_semantic.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.orc_coarse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        lib.orc_fine.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        lib.orc_codec_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        lib.orc_codec_tap.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        lib.orc_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.h = lib.orc_open(model_path.encode())
        if not self.h:
            raise RuntimeError(f"oracle: failed to load {model_path}")
        self.n_threads = n_threads
        self._fine_mfma = None                      # None: follow pyoracle.JOB_ORDER at every call

    def _sync_order(self):
        self.lib.orc_set_fine_mfma(self.h, int(JOB_ORDER if self._fine_mfma is None else self._fine_mfma))

    def close(self):
        if self.h:
            self.lib.orc_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration --------------------------------------------------------------
    def set_numerics(self, act_round_f16: bool = True, gelu_mode: int = 0):
        self.lib.orc_set_numerics(self.h, int(act_round_f16), gelu_mode)

    def set_dot_order(self, order: int = 0):
        """0 canonical (what the engine reproduces), 1 ggml's AVX2 order, 2 one sequential chain - study modes (Numerics::dot_order)."""
        self.lib.orc_set_dot_order(self.h, int(order))

    def set_fine_mfma(self, on: bool = True):
        """The fine model's weight products as C1 chains (False: the restated reference order, the default) or in the f16 matrix cores' order C1m (True: what
        lock-step jobs of the engine compute; Numerics::fine_mfma).  None: back to following pyoracle.JOB_ORDER."""
        self._fine_mfma = None if on is None else bool(on)
        self._sync_order()

    def set_codec_mfma(self, on: bool = True):
        """The codec's convolutions in the f16 matrix cores' order over kd = k * cin + ci (default, C9m) or as (ci, k) fmaf chains (C9)."""
        self.lib.orc_set_codec_mfma(self.h, int(on))

    def seed(self, s: int):
        self.lib.orc_seed(self.h, s)

    def gelu_table(self) -> np.ndarray:
        p = self.lib.orc_gelu_table(self.h)
        return np.ctypeslib.as_array(p, shape=(65536,)).copy()

    def hparams(self, which: int) -> dict:
        out = np.zeros(10, np.int32)
        self.lib.orc_hparams(self.h, which, out.ctypes.data)
        keys = ["n_layer", "n_head", "n_embd", "block_size", "bias", "n_in", "n_out", "n_lm_heads", "n_wtes", "ftype"]
        return dict(zip(keys, (int(v) for v in out)))

    @staticmethod
    def params(temp=0.0, fine_temp=0.0, min_eos_p=0.2, sliding_window_size=60, max_coarse_history=630,
               n_steps_text_encoder=768) -> OrcParams:
        return OrcParams(temp, fine_temp, min_eos_p, sliding_window_size, max_coarse_history, n_steps_text_encoder)

    # -- pieces -----------------------------------------------------------------------
    def tokenize(self, text: str) -> np.ndarray:
        out = np.zeros(513, np.int32)
        n = self.lib.orc_tokenize(self.h, text.encode("utf-8"), out.ctypes.data)
        assert n == 513
        return out

    def bert_tokenize(self, text: str, n_max: int = 256) -> np.ndarray:
        out = np.zeros(n_max, np.int32)
        n = self.lib.orc_bert_tokenize(self.h, text.encode("utf-8"), out.ctypes.data, n_max)
        return out[:n]

    def gpt_eval(self, which: int, tokens, n_past: int, merge_ctx: bool):
        tokens = _i32(tokens)
        n_out = self.hparams(which)["n_out"]
        logits = np.zeros(n_out, np.float32)
        np_new = self.lib.orc_gpt_eval(self.h, which, tokens.ctypes.data, len(tokens), n_past, int(merge_ctx),
                                       logits.ctypes.data, self.n_threads)
        if np_new < 0:
            raise RuntimeError("oracle gpt_eval failed")
        return logits, np_new

    def fine_eval(self, tokens_8x1024, nn: int) -> np.ndarray:
        tokens = _i32(tokens_8x1024).reshape(8, 1024)
        n_out = self.hparams(2)["n_out"]
        logits = np.zeros((1024, n_out), np.float32)
        self._sync_order()
        if self.lib.orc_fine_eval(self.h, tokens.ctypes.data, nn, logits.ctypes.data, self.n_threads) != 0:
            raise RuntimeError("oracle fine_eval failed")
        return logits

    def semantic(self, prompt513, p: OrcParams, want_eos_trace: bool = False):
        prompt = _i32(prompt513)
        out = np.zeros(1024, np.int32)
        tr = np.zeros(1024, np.float32)
        n = self.lib.orc_semantic(self.h, C.byref(p), prompt.ctypes.data, out.ctypes.data,
                                  tr.ctypes.data if want_eos_trace else None, self.n_threads)
        if n < 0:
            raise RuntimeError("oracle semantic stage failed")
        return (out[:n].copy(), tr) if want_eos_trace else out[:n].copy()

    def coarse(self, semantic, p: OrcParams) -> np.ndarray:
        sem = _i32(semantic)
        out = np.zeros((4096, 2), np.int32)
        T = self.lib.orc_coarse(self.h, C.byref(p), sem.ctypes.data, len(sem), out.ctypes.data, self.n_threads)
        if T < 0:
            raise RuntimeError("oracle coarse stage failed")
        return out[:T].copy()

    def fine(self, coarse_Tx2, p: OrcParams) -> np.ndarray:
        co = _i32(coarse_Tx2).reshape(-1, 2)
        out = np.zeros((max(len(co), 1), 8), np.int32)
        self._sync_order()
        T = self.lib.orc_fine(self.h, C.byref(p), co.ctypes.data, len(co), out.ctypes.data, self.n_threads)
        if T < 0:
            raise RuntimeError("oracle fine stage failed")
        return out[:T].copy()

    def codec_decode(self, codes_qxT) -> np.ndarray:
        codes = _i32(codes_qxT)
        n_q, T = codes.shape
        pcm = np.zeros(T * 320, np.float32)
        n = self.lib.orc_codec_decode(self.h, codes.ctypes.data, n_q, T, pcm.ctypes.data, self.n_threads)
        if n < 0:
            raise RuntimeError("oracle codec decode failed")
        return pcm[:n].copy()

    def codec_tap(self, codes_qxT, stage: int) -> np.ndarray:
        codes = _i32(codes_qxT)
        n_q, T = codes.shape
        out = np.zeros(T * 320 * 64, np.float32)
        n = self.lib.orc_codec_tap(self.h, codes.ctypes.data, n_q, T, stage, out.ctypes.data, out.size, self.n_threads)
        if n < 0:
            raise RuntimeError("oracle codec tap failed")
        return out[:n].copy()

    def generate(self, text: str, p: OrcParams) -> dict:
        sem = np.zeros(1024, np.int32)
        co = np.zeros((4096, 2), np.int32)
        fi = np.zeros((4096, 8), np.int32)
        pcm = np.zeros(4096 * 320, np.float32)
        res = OrcResult()
        self._sync_order()
        rc = self.lib.orc_generate(self.h, C.byref(p), text.encode("utf-8"), sem.ctypes.data, co.ctypes.data,
                                   fi.ctypes.data, pcm.ctypes.data, C.byref(res), self.n_threads)
        if rc != 0:
            raise RuntimeError(f"oracle generate failed rc={rc}")
        d = {k: getattr(res, k) for k, _ in OrcResult._fields_}
        d.update(semantic=sem[:res.n_semantic].copy(), coarse=co[:res.n_frames].copy(), fine=fi[:res.n_frames].copy(),
                 pcm=pcm[:res.n_samples].copy())
        return d
