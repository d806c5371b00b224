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


def test_tests_that_run_lock_step_jobs_say_so():
    """A GPU test that runs lock-step jobs (generate_batch / Batcher / fine_many) must carry one of the group markers: they set its place in the suite
    (conftest.py) and, for `lock_step_job` / `job_order`, the order the oracle computes the fine products in.  A new job test without a marker would fall
    into the row-level group and compare C1m ids with the oracle's C1."""
    import ast
    need = ("generate_batch", "Batcher", "fine_many")
    ok = {"lock_step_job", "concurrency", "boundary", "job_order"}
    missing = []
    for fn in ("test_gpu_parity.py", "test_gpu_batch_ragged.py"):
        src = open(os.path.join(ROOT, "tests", fn)).read()
        for node in ast.parse(src).body:
            if not (isinstance(node, ast.FunctionDef) and node.name.startswith("test_")):
                continue
            body = ast.get_source_segment(src, node)
            if not any(k in body for k in need):
                continue
            marks = {d.attr if isinstance(d, ast.Attribute) else getattr(getattr(d, "func", None), "attr", None) for d in node.decorator_list}
            if not (marks & ok) and "job_order(" not in body:            # (a test may also switch the oracle's order itself, around the job it compares)
                missing.append(f"{fn}::{node.name}")
    assert not missing, missing
