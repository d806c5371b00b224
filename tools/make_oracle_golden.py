#!/usr/bin/env python3
"""tests/golden/oracle_small_bench_<fmt>.npz: what the CPU oracle generates for the workload bench.py times (bark-small shapes, the bench
prompt, greedy, n_steps_text_encoder = 256), for the f16 file and its q4_0 quantisation: every semantic / coarse / fine id, the SHA-256 of
the PCM bytes and every 64th sample.  The -m gpu test of that workload compares the engine with these files (the oracle needs a minute
of CPU per format, which is GPU-box time there); tests/test_oracle_behaviour.py re-derives them from the oracle on the CPU every run,
so fixture and oracle cannot drift apart unnoticed.  usage: python tools/make_oracle_golden.py [f16 q4_0]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def file_sha256(path: str) -> np.ndarray:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return np.frombuffer(h.digest(), np.uint8).copy()


def workload(fmt: str) -> dict:
    import bench
    import conftest
    from oracle.pyoracle import Oracle
    from tools.make_synth_model import ensure_model
    path = ensure_model("small", 0)
    if fmt != "f16":
        path = conftest._quantized(path, fmt)
    text = bench.synth_prompts(64)[1]                  # bench.py: prompt (warmup + 0) with the default --warmup 1
    orc = Oracle(path, n_threads=8)
    ref = orc.generate(text, orc.params(n_steps_text_encoder=256))
    orc.close()
    pcm = np.ascontiguousarray(ref["pcm"], np.float32)
    return {"model_sha256": file_sha256(path),      # the synthetic model file this was generated from (deterministic; checked before use)
            "semantic": ref["semantic"].astype(np.int32), "coarse": ref["coarse"].astype(np.int32), "fine": ref["fine"].astype(np.int32),
            "pcm_len": np.int64(pcm.size), "pcm_sha256": np.frombuffer(hashlib.sha256(pcm.tobytes()).digest(), np.uint8).copy(),
            "pcm_every_64th": pcm[::64].copy()}


def main():
    for fmt in (sys.argv[1:] or ["f16", "q4_0"]):
        out = workload(fmt)
        dst = os.path.join(ROOT, "tests", "golden", f"oracle_small_bench_{fmt}.npz")
        np.savez_compressed(dst, **out)
        print("wrote", dst, os.path.getsize(dst), "bytes;", len(out["semantic"]), "semantic ids,", out["coarse"].shape, out["fine"].shape, int(out["pcm_len"]), "samples")


if __name__ == "__main__":
    main()
