cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_plugin.py tests/test_gpu_glue.py -m gpu -q -x -p no:cacheprovider -k "not test_reference_test_backend_ops and not whole_model_logits_against_cpu_fixture" > gpurun_out/gpu_tests2.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/gpu_tests2.log | cut -c1-300
