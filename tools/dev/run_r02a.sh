set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02a
timeout 900 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_edge.py tests/test_gpu_parity.py tests/test_gpu_lbvh.py tests/test_gpu_refit.py -x -q -m gpu > gpurun_out/r02a/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02a/pytest.log
tail -15 gpurun_out/r02a/pytest.log
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02a/pb -o pb -- python3 tools/dev/per_bounce.py > gpurun_out/r02a/per_bounce.log 2>&1
python3 tools/dev/per_bounce_join.py gpurun_out/r02a/per_bounce.log $(find gpurun_out/r02a/pb -name "*kernel_trace.csv" | head -1) | tee gpurun_out/r02a/per_bounce.txt
