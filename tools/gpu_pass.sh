cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_plan.py tests/test_gpu_parity.py -q -x > gpurun_out/plan_tests.log 2>&1; echo "plan+parity tests rc=$?"; tail -5 gpurun_out/plan_tests.log | cut -c1-400
timeout -k 10 200 python tools/loaderonly.py 2>&1 | tail -1
MI355Q_PLAN_GROUPS=0 timeout -k 10 200 python tools/loaderonly.py 2>&1 | tail -1
timeout -k 10 900 python -m pytest tests/test_plugin.py -q > gpurun_out/plugin_tests.log 2>&1; echo "plugin tests rc=$?"; tail -3 gpurun_out/plugin_tests.log | cut -c1-300
