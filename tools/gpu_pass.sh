cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
R=oracle/_ref/avx2
export LD_LIBRARY_PATH=$PWD/$R:$LD_LIBRARY_PATH
$R/gguf_synth --preset 8b --ftype q4_k_m --out /tmp/m.gguf > /dev/null 2>&1
GGML_BACKEND_PATH=$PWD/llama.cpp.dsp_amd/lib/libggml-mi355.so MI355_TIMING=1 MI355_GRAPH_STATS=1 timeout -k 10 300 $R/llama-bench -m /tmp/m.gguf -p 0 -n 128 -r 3 -ngl 99 -t 16 > gpurun_out/lb_timing.log 2>&1; echo rc=$?
grep -v "^ggml_\|^llama_\|load_backend" gpurun_out/lb_timing.log | tail -25 | cut -c1-250
