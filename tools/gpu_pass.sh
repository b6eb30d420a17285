cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/gpu_tests.log | cut -c1-300
