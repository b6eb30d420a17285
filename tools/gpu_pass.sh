# GPU pass (run through gpurun): plugin plan tests, whole-model timing through the plugin
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_plugin.py -m gpu -x -q -k "whole_model or suffix or decode_loop or fused or kv_cache or resident or flash" > gpurun_out/gpu_tests_plugin.log 2>&1; rc=$?
tail -40 gpurun_out/gpu_tests_plugin.log
[ $rc -ne 0 ] && exit 1
export GGML_BACKEND_PATH=$GRAFT_REPO_ROOT/llama.cpp.dsp_amd/lib/libggml-mi355.so
MP=oracle/_ref/avx2/model_parity
echo "== whole model 8b, 32 layers, through the plugin (plan)"
MI355_GRAPH_STATS=1 timeout -k 10 600 $MP --preset 8b --layers 32 --vocab 128256 --tokens 4 --bench 128 > gpurun_out/model8b_plan.log 2>&1; echo "rc $?"; tail -6 gpurun_out/model8b_plan.log
echo "== whole model 8b, node by node (MI355_NO_PLAN=1)"
MI355_NO_PLAN=1 MI355_GRAPH_STATS=1 timeout -k 10 600 $MP --preset 8b --layers 32 --vocab 128256 --tokens 2 --bench 64 --no-cpu > gpurun_out/model8b_noplan.log 2>&1; echo "rc $?"; tail -4 gpurun_out/model8b_noplan.log
