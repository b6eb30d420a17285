cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_glue.py -m gpu -x -q > gpurun_out/gpu_tests_glue.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_glue.log | cut -c1-300
[ $rc -ne 0 ] && { tail -60 gpurun_out/gpu_tests_glue.log | cut -c1-300; exit 1; }
export GGML_BACKEND_PATH=$GRAFT_REPO_ROOT/llama.cpp.dsp_amd/lib/libggml-mi355.so
MP=oracle/_ref/avx2/model_parity
for i in 1 2; do timeout -k 10 300 $MP --preset 8b --layers 32 --vocab 128256 --tokens 1 --no-cpu --pp 512 2>&1 | grep "prefill through"; done
rm -rf gpurun_out/prof_pp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_pp -- $MP --preset 8b --layers 4 --vocab 32000 --tokens 1 --no-cpu --pp 512 > gpurun_out/prof_pp.log 2>&1; echo "rocprof exit $?"
python3 tools/trace_top.py gpurun_out/prof_pp 20
timeout -k 10 900 python -m pytest tests/test_plugin.py -m gpu -x -q -k "backend_ops or layer or whole_model_decode" > gpurun_out/gpu_tests_plugin.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_plugin.log | cut -c1-300
[ $rc -ne 0 ] && { tail -60 gpurun_out/gpu_tests_plugin.log | cut -c1-300; exit 1; }
exit 0
