cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export GGML_BACKEND_PATH=$GRAFT_REPO_ROOT/llama.cpp.dsp_amd/lib/libggml-mi355.so
for wt in q4_k_m iq4_xs q5_k_m q8_0 q3_k; do
echo "== $wt"
MI355_GRAPH_STATS=1 timeout -k 10 300 oracle/_ref/avx2/model_parity --preset 8b --layers 32 --vocab 128256 --tokens 1 --no-cpu --bench 64 --pp 512 --wtype $wt 2>&1 | grep -o "us/token = [0-9.]* tok/s\|tokens in [0-9.]* us = [0-9.]* tok/s\|decode plans: [0-9]* graph_compute calls"
done
exit 0
