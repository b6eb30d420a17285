cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export GGML_BACKEND_PATH=$GRAFT_REPO_ROOT/llama.cpp.dsp_amd/lib/libggml-mi355.so
MP=oracle/_ref/avx2/model_parity
MI355_PLAN_DEBUG=1 timeout -k 10 300 $MP --preset 8b --layers 2 --vocab 32000 --tokens 2 --no-cpu > gpurun_out/plan_dbg.log 2>&1; grep -c "" gpurun_out/plan_dbg.log; grep "MI355 plan\|  stage\|decode plans" gpurun_out/plan_dbg.log | cut -c1-260 | head -40
exit 0
