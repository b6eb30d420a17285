cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export GGML_BACKEND_PATH=$PWD/llama.cpp.dsp_amd/lib/libggml-mi355.so MI355_GRAPH_STATS=1
MP_DEBUG=1 timeout -k 10 300 oracle/_ref/avx2/model_parity --preset small --layers 4 --vocab 8192 --tokens 4 --teacher 4 --fa > gpurun_out/teacher_fa.log 2>&1; echo rc=$?
grep -E "dbg|teacher-forced" gpurun_out/teacher_fa.log | cut -c1-220 | head -60
