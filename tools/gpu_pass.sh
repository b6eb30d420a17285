cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "q2_K or q3_K or golden" > gpurun_out/gpu_tests_parity.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_parity.log | cut -c1-300
[ $rc -ne 0 ] && { tail -60 gpurun_out/gpu_tests_parity.log | cut -c1-300; exit 1; }
timeout -k 10 300 python tools/typebench.py gpurun_out/typebench.md > /dev/null 2>&1; cat gpurun_out/typebench.md | tail -21
exit 0
