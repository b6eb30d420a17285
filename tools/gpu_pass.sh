cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_plugin.py -m gpu -x -q -k "recipes" > gpurun_out/gpu_tests_plugin.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_plugin.log | cut -c1-300
[ $rc -ne 0 ] && { tail -70 gpurun_out/gpu_tests_plugin.log | cut -c1-300; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plan.py -m gpu -x -q -k "plan" 2>&1 | tail -2
export GGML_BACKEND_PATH=$GRAFT_REPO_ROOT/llama.cpp.dsp_amd/lib/libggml-mi355.so
for wt in iq4_xs q8_0 q3_k; do
timeout -k 10 300 oracle/_ref/avx2/model_parity --preset 8b --layers 32 --vocab 128256 --tokens 1 --no-cpu --bench 64 --pp 512 --wtype $wt 2>&1 | grep "decode through\|prefill through" | cut -c1-60,190-330
done
exit 0
