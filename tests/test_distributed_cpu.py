"""The N > 1 path of bench.py (BASELINE config 5) on CPU: two gloo ranks shard the 64 synthetic prompts by length without overlap,
all_gather the per-prompt sample counts, gather the PCM on rank 0 and combine their timings the way the bench contract demands
(MAX time, SUM work); no collective inside an utterance.  The engine itself is exercised by the -m gpu twin of this test
(tests/test_gpu_parity.py::test_two_ranks_gather_the_single_process_pcm)."""
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
idx = bench.shard_prompts(prompts, rank, world)
dt, audio = bench.reduce_timing(1.0 + rank, 5.12 * 4, world)
# stand-in for the engine: utterance i "generates" 100 + 7 i samples whose values encode (i, position)
import numpy as np
pcms = [(np.arange(100 + 7 * i, dtype=np.float32) + 1000.0 * i) for i in idx]
counts, gathered = bench.gather_batch_results(pcms, idx, len(prompts), rank, world, None)
allp = [None] * world
dist.all_gather_object(allp, idx)
if rank == 0:
    ok = all(np.array_equal(gathered[i], np.arange(100 + 7 * i, dtype=np.float32) + 1000.0 * i) for i in range(len(prompts)))
    print(json.dumps({"dt": dt, "audio": audio, "shards": allp, "counts": counts.tolist(), "n_gathered": len(gathered), "pcm_ok": bool(ok)}))
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
    flat = out["shards"][0] + out["shards"][1]
    assert sorted(flat) == list(range(64)) and len(out["shards"][0]) == len(out["shards"][1]) == 32      # disjoint, complete, balanced
    sys.path.insert(0, ROOT)
    import bench
    prompts = bench.synth_prompts(64)
    lens = [[len(prompts[i]) for i in sh] for sh in out["shards"]]
    assert all(l == sorted(l) for l in lens)                   # every shard sorted by length (neighbours share a lock-step batch)
    assert abs(sum(lens[0]) - sum(lens[1])) <= max(map(max, lens))      # dealt snake-wise: no rank holds all the long utterances
    for world in (4, 8):                                       # the same properties for the node sizes the driver runs
        shards = [bench.shard_prompts(prompts, r, world) for r in range(world)]
        assert sorted(sum(shards, [])) == list(range(64)) and {len(s) for s in shards} == {64 // world}
        tot = [sum(len(prompts[i]) for i in s) for s in shards]
        assert max(tot) - min(tot) <= max(len(p) for p in prompts)
    assert out["counts"] == [100 + 7 * i for i in range(64)]   # all_gather of the sample counts
    assert out["n_gathered"] == 64 and out["pcm_ok"]           # gather of the PCM on rank 0, bit for bit


def test_committed_bench_line_follows_the_contract():
    """profiles/r0N_bench_small_n1.json is a bench.py line from the GPU box: keys and units of the driver's contract."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import glob
    newest = sorted(glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]_bench_small_n1.json")))[-1]      # the latest round's line
    d = json.load(open(newest))
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


def test_cpu_baseline_leg_reports_the_contract_keys(toy_model):
    """bench.py's cpu_baseline leg (the oracle on pinned cores, same workload as the headline) - here on the toy model"""
    import os
    import bench
    before = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    try:
        c = bench.cpu_baseline_leg(toy_model, bench.synth_prompts(2)[1], 16)
    finally:
        if before is not None:
            os.sched_setaffinity(0, before)
    assert c["kind"] == "port" and c["unit"] == "audio-s/s" and 1 <= c["cores"] <= 4 and c["value"] > 0 and "n_steps_text_encoder=16" in c["sample"]
    assert set(c["stage_ms_per_token"]) == {"semantic", "coarse", "fine"}


def test_gpus_flag_starts_its_own_ranks_or_refuses(monkeypatch, capsys):
    """`python bench.py --gpus N` is the N-rank job: without a launcher the process re-executes itself under torch.distributed.run with
    --nproc-per-node N on 127.0.0.1; when --gpus disagrees with WORLD_SIZE, or the node shows fewer GPUs than N, it exits 2 and prints NO line
    (an N-GPU request must never produce an n_gpus = 1 record).  The plain two-rank command itself runs in the GPU suite
    (tests/test_gpu_parity.py::test_two_ranks_gather_the_single_process_pcm)."""
    import pytest
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_execv(exe, argv):
        seen["exe"], seen["argv"] = exe, list(argv)
        raise SystemExit(0)
    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--backend", "gloo", "--all-ranks-on-device0"])
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    argv = seen["argv"]
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in argv
    assert argv[argv.index("--nproc-per-node") + 1] == "4" and argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    k = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[k + 1:] == ["--gpus", "4", "--steps", "2", "--backend", "gloo", "--all-ranks-on-device0"]
    # under a launcher whose world differs from --gpus: refused, nothing on stdout
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 2 and capsys.readouterr().out == ""
    monkeypatch.delenv("WORLD_SIZE")
    # a node with fewer GPUs than ranks (this container has none): refused as well
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode == 2 and r.stdout.strip() == "" and "refusing" in r.stderr
