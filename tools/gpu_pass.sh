cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_plugin.py -q -k "mask_copy or teacher or fixture" > gpurun_out/plugin_tests.log 2>&1; echo "plugin tests rc=$?"; tail -6 gpurun_out/plugin_tests.log | cut -c1-300
