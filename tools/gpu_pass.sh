cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/gpu_tests_parity.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_parity.log | cut -c1-300
[ $rc -ne 0 ] && { tail -60 gpurun_out/gpu_tests_parity.log | cut -c1-300; exit 1; }
timeout -k 10 900 python -m pytest tests/test_plugin.py -m gpu -x -q -k "backend_ops or moe" > gpurun_out/gpu_tests_plugin.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_plugin.log | cut -c1-300
[ $rc -ne 0 ] && { tail -60 gpurun_out/gpu_tests_plugin.log | cut -c1-300; exit 1; }
exit 0
