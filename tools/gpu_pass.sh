cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_plan.py -q > gpurun_out/plan_tests.log 2>&1; rc=$?; echo "regs plan tests rc=$rc"; tail -1 gpurun_out/plan_tests.log
MI355Q_PLAN_ENGINE=ring timeout -k 10 600 python -m pytest tests/test_gpu_plan.py -q > gpurun_out/plan_tests_ring.log 2>&1; rc=$?; echo "ring plan tests rc=$rc"; tail -1 gpurun_out/plan_tests_ring.log
timeout -k 10 200 python tools/loaderonly.py 2>&1 | tail -1
MI355Q_PLAN_ENGINE=ring timeout -k 10 200 python tools/loaderonly.py 2>&1 | tail -1
