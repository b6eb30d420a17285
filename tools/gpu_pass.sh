cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_plan.py -q -x > gpurun_out/plan_tests.log 2>&1; echo "plan tests rc=$?"; tail -3 gpurun_out/plan_tests.log | cut -c1-300
for i in 1 2 3; do
  echo -n "skip: "; timeout -k 10 200 python tools/loaderonly.py 2>&1 | tail -1
  echo -n "all : "; MI355Q_PLAN_WO_SKIP=0 timeout -k 10 200 python tools/loaderonly.py 2>&1 | tail -1
done
