cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_plugin.py -q -k "teacher" > gpurun_out/plugin_tests.log 2>&1; echo "plugin tests rc=$?"; tail -3 gpurun_out/plugin_tests.log | cut -c1-300
grep -E "teacher-forced" gpurun_out/plugin_tests.log | head -12 | cut -c1-300
