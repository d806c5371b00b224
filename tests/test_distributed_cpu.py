"""The N > 1 path of bench.py on CPU: two gloo ranks shard the synthetic prompt set without overlap and
combine their timings the way the bench contract demands (MAX time, SUM work); no data-path collective."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
sys.path.insert(0, %r)
import torch.distributed as dist
import bench
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
prompts = bench.synth_prompts(64)
mine = [bench.prompt_for(s, rank, world, prompts) for s in range(4)]
dt, audio = bench.reduce_timing(1.0 + rank, 5.12 * 4, world)
allp = [None] * world
dist.all_gather_object(allp, mine)
if rank == 0:
    print(json.dumps({"dt": dt, "audio": audio, "prompts": allp}))
dist.destroy_process_group()
"""


def test_two_rank_sharding_and_reduction(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["dt"] == 2.0                                   # MAX over ranks
    assert abs(out["audio"] - 2 * 5.12 * 4) < 1e-9            # SUM over ranks
    flat = out["prompts"][0] + out["prompts"][1]
    assert len(set(flat)) == len(flat) == 8                    # disjoint shards
    import bench
    prompts = bench.synth_prompts(64)
    assert out["prompts"][0] == [prompts[0], prompts[2], prompts[4], prompts[6]]


def test_committed_bench_line_follows_the_contract():
    """profiles/r01_bench_small_n1.json is a bench.py line from the GPU box: keys and units of the driver's contract."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "r01_bench_small_n1.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["n_gpus"] == 1 and d["value"] > 10.0 and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
