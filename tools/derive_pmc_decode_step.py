#!/usr/bin/env python3
"""rNN_pmc_decode_step.json from the two PMC summaries of tools/rocpd_pmc.py (FETCH_SIZE and WRITE_SIZE passes over tools/profile_decode.py,
which replays semantic decode steps at context 640): HBM-side traffic of ONE decode step (all 62 kernels) and of its dominant kernel (the
LayerNorm + FC GEMV) against their algorithmic bytes.  bench.py reads `step_traffic_bytes` / `traffic_bytes_per_launch` from it.
   derive_pmc_decode_step.py FETCH.json WRITE.json OUT.json [context of the profiled steps, default 640]"""
import json, os, sys

f = json.load(open(sys.argv[1])); w = json.load(open(sys.argv[2]))
def fc(rows):
    r = [x for x in rows if "gemv_ln_wg_kernelILi6ELb0ELb0E" in x["kernel"] and x["grid_threads"] == 192 * 256]
    if not r:
        raise SystemExit("the FC instance of gemv_ln_wg_kernel is not in the summary")
    return r[0]
ff, fw = fc(f), fc(w)
steps = ff["launches"] / 12.0                                    # one FC launch per layer, 12 layers
decode = lambda rows: [x for x in rows if "barkhip" in x["kernel"]]
rd = sum(x["avg"] * x["launches"] for x in decode(f)) * 1024 * 2 / steps      # gfx950: FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads
wr = sum(x["avg"] * x["launches"] for x in decode(w)) * 1024 / steps
alg_fc = 3072 * 768 * 2
ctxlen = int(sys.argv[4]) if len(sys.argv) > 4 else 640
alg_step = 12 * 12 * 768 * 768 * 2 + 10048 * 768 * 2 + 2 * ctxlen * 768 * 12 * 4      # weights + LM head + f32 K / V rows read (bark-small, semantic)
out = {
    "source": f"profiles/{os.path.basename(sys.argv[1])}, profiles/{os.path.basename(sys.argv[2])} (rocprofv3 --kernel-trace --pmc <counter> -- python tools/profile_decode.py; separate passes)",
    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950; unit KB",
    "steps_profiled": steps,
    "step_hbm_read_bytes": rd, "step_hbm_write_bytes": wr, "step_traffic_bytes": rd + wr,
    "context": ctxlen, "step_algorithmic_bytes": alg_step, "step_traffic_over_algorithmic": (rd + wr) / alg_step,
    "kernel": "gemv_ln_wg_kernel<6,false,false>, grid 192 x 256 threads: LayerNorm + FC 3072x768 f16 + GELU of the decode step (bark-small)",
    "launches": ff["launches"], "traffic_bytes_per_launch": ff["avg"] * 2048 + fw["avg"] * 1024,
    "algorithmic_bytes_per_launch": alg_fc, "traffic_over_algorithmic": (ff["avg"] * 2048 + fw["avg"] * 1024) / alg_fc,
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
