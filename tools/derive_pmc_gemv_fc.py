#!/usr/bin/env python3
"""rNN_pmc_gemv_fc.json from the two PMC summaries of tools/rocpd_pmc.py (FETCH_SIZE and WRITE_SIZE passes over tools/profile_decode.py):
HBM-side traffic per launch of the dominant decode kernel - the LayerNorm + FC GEMV, gemv_ln_wg_kernel<6,false,false> on a grid of
192 workgroups x 256 threads - against its algorithmic bytes.   derive_pmc_gemv_fc.py FETCH.json WRITE.json OUT.json"""
import json, os, sys

def pick(path):
    rows = [r for r in json.load(open(path)) if "gemv_ln_wg_kernelILi6ELb0ELb0E" in r["kernel"] and r["grid_threads"] == 192 * 256]
    if not rows:
        raise SystemExit(f"{path}: the FC instance of gemv_ln_wg_kernel is not in the summary")
    return rows[0]

f, w = pick(sys.argv[1]), pick(sys.argv[2])
alg = 3072 * 768 * 2
rd, wr = f["avg"] * 1024 * 2, w["avg"] * 1024
out = {
    "kernel": "gemv_ln_wg_kernel<6,false,false>, grid 192 x 256 threads: LayerNorm + FC 3072x768 f16 + GELU of the decode step (bark-small)",
    "source": f"profiles/{os.path.basename(sys.argv[1])}, profiles/{os.path.basename(sys.argv[2])} (rocprofv3 --kernel-trace --pmc <counter> -- python tools/profile_decode.py; separate passes)",
    "launches": f["launches"],
    "FETCH_SIZE_raw_KB_per_launch": f["avg"], "WRITE_SIZE_raw_KB_per_launch": w["avg"],
    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (counts 64 B per 128 B request); unit KB",
    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
    "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (rd + wr) / alg,
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
