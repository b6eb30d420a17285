cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1150 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_full.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_full.log | cut -c1-300
[ $rc -ne 0 ] && { tail -40 gpurun_out/gpu_tests_full.log | cut -c1-300; exit $rc; }
exit 0
