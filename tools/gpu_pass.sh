cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_plan.py -q -x > gpurun_out/plan_tests.log 2>&1; echo "plan tests rc=$?"; tail -8 gpurun_out/plan_tests.log | cut -c1-300
