cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python tools/ppbench.py > gpurun_out/ppbench.log 2>&1; tail -8 gpurun_out/ppbench.log
export GGML_BACKEND_PATH=$GRAFT_REPO_ROOT/llama.cpp.dsp_amd/lib/libggml-mi355.so
timeout -k 10 300 oracle/_ref/avx2/model_parity --preset 8b --layers 32 --vocab 128256 --tokens 1 --no-cpu --pp 512 2>&1 | grep "prefill through"
timeout -k 10 900 python -m pytest tests/test_plugin.py -m gpu -x -q -k "backend_ops" > gpurun_out/gpu_tests_plugin.log 2>&1; tail -2 gpurun_out/gpu_tests_plugin.log | cut -c1-200
exit 0
