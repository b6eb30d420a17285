cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export GGML_BACKEND_PATH=$GRAFT_REPO_ROOT/llama.cpp.dsp_amd/lib/libggml-mi355.so
MP=oracle/_ref/avx2/model_parity
timeout -k 10 300 $MP --preset 8b --layers 32 --vocab 128256 --tokens 2 --no-cpu --bench 128 2>&1 | grep -v load_backend | tail -4
MI355_NO_PLAN=1 timeout -k 10 300 $MP --preset 8b --layers 32 --vocab 128256 --tokens 2 --no-cpu --bench 128 2>&1 | grep -v load_backend | tail -3
exit 0
