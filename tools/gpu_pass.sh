cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 380 python -m pytest tests/test_plugin.py -m gpu -q -x -p no:cacheprovider -k "test_reference_test_backend_ops" > gpurun_out/gpu_tests4.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/gpu_tests4.log | cut -c1-300
