#!/usr/bin/env python3
"""In-kernel time line of one decode step (diagnostic build: bark.cpp_amd/build_variant.sh trace -DBARK_TRACE).

Every wave of the decode kernels logs s_memrealtime (100 MHz) at entry, after its first kernel argument is usable, after its
operands have arrived (dot product done) and at its end.  Per kernel of the captured step this prints, in microseconds relative to
the step's first wave: first entry, last exit, the gap to the previous kernel's last exit, and the medians of the three phases.

  python tools/trace_decode.py [small] [ctx] [out.json] [f16 | q4_0 | ...]
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BARK_HIP_LIBRARY", os.path.join(ROOT, "bark.cpp_amd", "lib", "libbark_trace.so"))

import numpy as np                                   # noqa: E402
from bark_amd_loader import load_package             # noqa: E402
from tools.make_synth_model import ensure_model      # noqa: E402


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "small"
    ctxlen = int(sys.argv[2]) if len(sys.argv) > 2 else 640
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    fmt = sys.argv[4] if len(sys.argv) > 4 else "f16"     # or a block format: the model file is quantised first
    pkg = load_package()
    lib = pkg.load_library()
    path = ensure_model(preset, 0)
    if fmt != "f16":
        qpath = path[:-4] + "_%s.bin" % fmt
        if not os.path.exists(qpath):
            assert lib.bark_model_quantize(path.encode(), qpath.encode(), {"q4_0": 2, "q4_1": 3, "q8_0": 7, "q5_0": 8, "q5_1": 9}[fmt])
        path = qpath
    ctx = pkg.BarkContext.load_model(path, pkg.default_params(temp=0.0, fine_temp=0.0, n_steps_text_encoder=32), 0)
    replays = 4
    cap = 1 << 19
    buf = np.zeros((cap, 8), np.uint64)
    lib.bark_hip_trace_decode_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    n = lib.bark_hip_trace_decode_step(ctx._h, 0, ctxlen, replays, buf.ctypes.data, cap)
    assert n > 0, n
    rec = buf[:n]
    rec = rec[rec[:, 2] != 0]                          # slots of dead lanes / early-out workgroups stay zero
    kid = (rec[:, 0] & 0xFFFF).astype(np.int64)
    xcc = ((rec[:, 0] >> 16) & 15).astype(np.int64)
    t = rec[:, 2:6].astype(np.int64)
    tab = rec[:, 6:8].astype(np.int64)
    tick = 0.01                                       # us per s_memrealtime tick
    n_k = int(kid.max()) + 1
    # split the records of each kernel into the `replays` consecutive replays (equal wave counts, ordered by entry time)
    rows = []
    for r in range(replays):
        per = []
        for k in range(n_k):
            idx = np.flatnonzero(kid == k)
            idx = idx[np.argsort(t[idx, 0], kind="stable")]
            w = len(idx) // replays
            per.append(idx[r * w:(r + 1) * w])
        rows.append(per)
    rep = rows[-1]                                    # report the last replay (steady state)
    t_first = min(t[i, 0].min() for i in rep if len(i))
    prev_end = None
    table = []
    names = {0: "ln+qkv", 1: "attn", 2: "proj", 3: "ln+fc+gelu", 4: "mproj"}
    for k, idx in enumerate(rep):
        if not len(idx):
            continue
        tt = t[idx]
        first, last = tt[:, 0].min(), tt[:, 3].max()
        row = {"kid": k, "name": names.get(k % 5, "?") if k < n_k - 2 else ("lm_head" if k == n_k - 2 else "sample+embed"),
               "waves": int(len(idx)), "xcds": int(len(set(xcc[idx].tolist()))),
               "first_entry_us": (first - t_first) * tick, "last_exit_us": (last - t_first) * tick,
               "gap_from_prev_us": None if prev_end is None else (first - prev_end) * tick,
               "entry_spread_us": (tt[:, 0].max() - first) * tick,
               "kernarg_us_med": float(np.median(tt[:, 1] - tt[:, 0])) * tick,
               "operands_us_med": float(np.median(np.maximum(tt[:, 2], tt[:, 1]) - tt[:, 1])) * tick,
               "operands_us_max": float((np.maximum(tt[:, 2], tt[:, 1]) - tt[:, 1]).max()) * tick,
               "tail_us_med": float(np.median(tt[:, 3] - np.maximum(tt[:, 2], tt[:, 1]))) * tick, "tail_us_max": float((tt[:, 3] - np.maximum(tt[:, 2], tt[:, 1])).max()) * tick,
               "wave_us_med": float(np.median(tt[:, 3] - tt[:, 0])) * tick, "wave_us_max": float((tt[:, 3] - tt[:, 0]).max()) * tick,
               "span_us": (last - first) * tick}
        ab = tab[idx]
        if ab[:, 0].max() > 0:                            # extra stamps (attention): after the score, after exp + sums, relative to kernarg-ready
            ok = ab[:, 0] > 0
            row["ta_us_med"] = float(np.median(ab[ok, 0] - tt[ok, 1])) * tick; row["ta_us_max"] = float((ab[ok, 0] - tt[ok, 1]).max()) * tick
            row["tb_us_med"] = float(np.median(ab[ok, 1] - tt[ok, 1])) * tick; row["tb_us_max"] = float((ab[ok, 1] - tt[ok, 1]).max()) * tick
        prev_end = last
        table.append(row)
    step_span = [(max(t[i, 3].max() for i in rp if len(i)) - min(t[i, 0].min() for i in rp if len(i))) * tick for rp in rows]
    inter = [(min(t[i, 0].min() for i in rows[r + 1] if len(i)) - max(t[i, 3].max() for i in rows[r] if len(i))) * tick for r in range(replays - 1)]
    print("records", n, "kernels", n_k, "step spans us", step_span, "gaps between replays us", inter)
    hdr = ("kid", "name", "waves", "first_entry_us", "gap_from_prev_us", "entry_spread_us", "kernarg_us_med", "operands_us_med", "operands_us_max", "tail_us_med", "tail_us_max", "wave_us_max", "span_us")
    print(" ".join("%14s" % h for h in hdr))
    for row in table[:12] + table[-7:]:
        print(" ".join("%14s" % (("%.2f" % row[h]) if isinstance(row[h], float) else row[h]) for h in hdr))
    agg = {}
    for row in table:
        a = agg.setdefault(row["name"], {"n": 0, "span": 0.0, "gap": 0.0, "kernarg": 0.0, "operands": 0.0, "tail": 0.0, "spread": 0.0})
        a["n"] += 1; a["span"] += row["span_us"]; a["gap"] += row["gap_from_prev_us"] or 0.0
        a["kernarg"] += row["kernarg_us_med"]; a["operands"] += row["operands_us_med"]; a["tail"] += row["tail_us_med"]; a["spread"] += row["entry_spread_us"]
    # per wave index of the workgroup (attention kernel of layer 1): which role is the late one?
    wv = ((rec[:, 1] >> 40) & 0xFF).astype(np.int64)
    idx = rep[6] if len(rep) > 6 else rep[1]
    if len(idx) and tab[idx, 0].max() > 0:
        print("kid 6 by wave index: score-ready / stats-ready / end, us after kernarg (median over workgroups)")
        for w in sorted(set(wv[idx].tolist())):
            ii = idx[wv[idx] == w]
            print("  wave %2d: %.2f %.2f %.2f" % (w, np.median(tab[ii, 0] - t[ii, 1]) * tick, np.median(tab[ii, 1] - t[ii, 1]) * tick, np.median(t[ii, 3] - t[ii, 1]) * tick))
    # QKV kernel of layer 1: the main workgroups against the partial-score copies of the q workgroups
    if len(rep) > 5 and len(rep[5]):
        idx = rep[5]
        wgid = (rec[idx, 1] & 0xFFFFFF).astype(np.int64)
        n_main = 144 if preset == "small" else 0
        for name, sel in (("main", wgid < n_main), ("copies", wgid >= n_main)):
            ii = idx[sel]
            if len(ii):
                print("kid 5 %-6s waves %4d: operands med %.2f max %.2f | end (after kernarg) med %.2f max %.2f" % (
                    name, len(ii), np.median(t[ii, 2] - t[ii, 1]) * tick, (t[ii, 2] - t[ii, 1]).max() * tick,
                    np.median(t[ii, 3] - t[ii, 1]) * tick, (t[ii, 3] - t[ii, 1]).max() * tick))
    # LayerNorm-fused kernels of layer 1 (QKV kid 5, FC kid 8): wave 0's stamps
    for k in (5, 8):
        if len(rep) > k and len(rep[k]):
            idx = rep[k]
            ii = idx[(wv[idx] == 0) & (tab[idx, 0] > 0)]
            if len(ii):
                print("kid %d wave 0 (us after kernarg): row arrived %.2f | normalised row in LDS %.2f | dot done %.2f | end %.2f" % (
                    k, np.median(tab[ii, 0] - t[ii, 1]) * tick, np.median(tab[ii, 1] - t[ii, 1]) * tick,
                    np.median(t[ii, 2] - t[ii, 1]) * tick, np.median(t[ii, 3] - t[ii, 1]) * tick))
    for row in table[:12]:
        if "ta_us_med" in row:
            print("kid %d extra stamps after kernarg: score med %.2f max %.2f | exp+sum med %.2f max %.2f | mix (operands) med %.2f" % (
                row["kid"], row["ta_us_med"], row["ta_us_max"], row["tb_us_med"], row["tb_us_max"], row["operands_us_med"]))
    print("\nper kernel type (mean over the step): span = first entry .. last exit; gap = previous last exit .. first entry")
    for name, a in agg.items():
        print("%-14s n=%2d span %.2f gap %.2f | entry spread %.2f kernarg %.2f operands %.2f tail %.2f" % (
            name, a["n"], a["span"] / a["n"], a["gap"] / a["n"], a["spread"] / a["n"], a["kernarg"] / a["n"], a["operands"] / a["n"], a["tail"] / a["n"]))
    if out_path:
        json.dump({"preset": preset, "ctx": ctxlen, "step_span_us": step_span, "replay_gaps_us": inter, "kernels": table}, open(out_path, "w"), indent=1)
    ctx.free()


if __name__ == "__main__":
    main()
