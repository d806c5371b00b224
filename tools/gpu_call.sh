#!/bin/bash
# One parameterised GPU-box script (replaces the per-call scripts of rounds 1 - 3): gpurun -- 'tools/gpu_call.sh <tag> <step> [<step> ...]'
#   steps:  test:<pytest -k expression | all>   fine_ab   batch_tp   bench   stats   soak   <any other word>: tools/<word>.py if it exists
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out
TAG=$1; shift
for step in "$@"; do
    case "$step" in
        testnx:*) timeout 1200 python -m pytest tests -m gpu -q -k "${step#testnx:}" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${TAG}_pytest.log; tail -40 gpurun_out/${TAG}_pytest.log ;;
        test:all) timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${TAG}_pytest.log; tail -5 gpurun_out/${TAG}_pytest.log ;;
        test:*)   timeout 1200 python -m pytest tests -m gpu -x -q -k "${step#test:}" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${TAG}_pytest.log; tail -25 gpurun_out/${TAG}_pytest.log ;;
        fine_ab)  timeout 400 python tools/fine_ab.py c1_default c1m:BARK_HIP_FINE_ORDER=c1m c1_rows_attn:BARK_HIP_CROSSCHECK=512 c1m_rows_attn:BARK_HIP_CROSSCHECK=512,BARK_HIP_FINE_ORDER=c1m c1m_x8_default:FINE_WINDOWS=8 c1_x8:BARK_HIP_FINE_ORDER=c1,FINE_WINDOWS=8 c1m_x8_rows_attn:BARK_HIP_CROSSCHECK=512,FINE_WINDOWS=8 tol:BARK_HIP_FAST_GEMM=1 tol_x8:BARK_HIP_FAST_GEMM=1,FINE_WINDOWS=8 > gpurun_out/${TAG}_fine_ab.txt 2>&1; cat gpurun_out/${TAG}_fine_ab.txt ;;
        batch_tp) timeout 600 python tools/batch_throughput.py 256 small > gpurun_out/${TAG}_batch_tp.txt 2>&1; cat gpurun_out/${TAG}_batch_tp.txt ;;
        bench)    timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 1500 gpurun_out/${TAG}_bench.json ;;
        soak)     python tools/mfma_f16_order.py soak gpurun_out/${TAG}_mfma_f16_soak.txt ;;
        *)        if [ -f "tools/$step.py" ]; then timeout 900 python "tools/$step.py" > gpurun_out/${TAG}_$step.txt 2>&1; tail -30 gpurun_out/${TAG}_$step.txt; else echo "unknown step $step"; fi ;;
    esac
done
