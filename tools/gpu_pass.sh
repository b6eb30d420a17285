cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_full.log 2>&1; rc=$?
tail -4 gpurun_out/gpu_tests_full.log | cut -c1-300
[ $rc -ne 0 ] && { tail -40 gpurun_out/gpu_tests_full.log | cut -c1-300; exit 1; }
export GGML_BACKEND_PATH=$GRAFT_REPO_ROOT/llama.cpp.dsp_amd/lib/libggml-mi355.so
MP=oracle/_ref/avx2/model_parity
echo "== 2 'devices' (one GPU exposed twice), layer split through ggml_backend_sched"
MI355_PLAN_DEBUG=1 MI355_DUP_DEVICES=2 MI355_GRAPH_STATS=1 timeout -k 10 300 $MP --preset small --layers 4 --vocab 8192 --tokens 6 --devs MI355_0,MI355_1 > gpurun_out/model_2dev.log 2>&1; echo "rc $?"; grep -v "^load_backend\|plan attn" gpurun_out/model_2dev.log | tail -14 | cut -c1-300
