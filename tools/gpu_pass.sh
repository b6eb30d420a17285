cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_plan.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > gpurun_out/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/gpu_tests.log | cut -c1-300
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
