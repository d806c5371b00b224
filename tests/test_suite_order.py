"""The GPU suite's collection order is part of the contract with the driver (`pytest tests -x -q -m gpu`): a late failure must not hide the tests
that pin the hot path row by row (round 4 ended red on a concurrency test that was collected FIRST and hid 108 row-level tests behind `-x`).
conftest.py orders the GPU tests: loader -> per-row parity -> boundary binaries / ranks -> lock-step jobs -> concurrency.  This CPU test holds it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpu_suite_runs_row_level_parity_first_and_concurrency_last():
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "--collect-only", "-q"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    ids = [l.strip() for l in r.stdout.splitlines() if "::" in l]
    assert len(ids) > 100, r.stdout[-1500:] + r.stderr[-1500:]
    pos = {}
    for k, i in enumerate(ids):
        pos.setdefault(i.split("::")[1].split("[")[0], k)

    def before(a, b):
        assert pos[a] < pos[b], f"{a} must be collected before {b}"
    assert ids[0].startswith("tests/test_gpu_loader.py")
    for row_level in ("test_semantic_eval_prefill_and_decode", "test_fine_eval", "test_codec_decode", "test_small_model_decode_and_stages", "test_large_model_shapes",
                      "test_bench_workload_matches_the_oracle", "test_q4_0_small_model_logits", "test_graph_and_eager_agree"):
        before(row_level, "test_reference_cli_binary_runs_on_this_engine")
        before(row_level, "test_randomised_lock_step_jobs_against_the_oracle")
    before("test_reference_cli_binary_runs_on_this_engine", "test_in_engine_batch_matches_oracle")
    for job in ("test_randomised_lock_step_jobs_against_the_oracle", "test_in_engine_batch_matches_oracle", "test_few_slot_route_equals_the_matrix_core_route"):
        before(job, "test_cloned_contexts_serve_jobs_from_concurrent_host_threads")
        before(job, "test_request_batcher_serves_concurrent_submitters")
    assert ids[-1].split("::")[1].startswith(("test_concurrent_", "test_native_batch_server", "test_request_batcher", "test_cloned_contexts"))
