cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_full.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_full.log | cut -c1-300
[ $rc -ne 0 ] && { tail -40 gpurun_out/gpu_tests_full.log | cut -c1-300; exit $rc; }
R=oracle/_ref/avx2
export LD_LIBRARY_PATH=$PWD/$R:$LD_LIBRARY_PATH
$R/gguf_synth --preset 8b --ftype q4_k_m --out /tmp/m.gguf > /dev/null 2>&1
for args in "-fa 1" "-fa 1 -ctk q8_0 -ctv q8_0"; do
  echo "== llama-bench $args"
  GGML_BACKEND_PATH=$PWD/llama.cpp.dsp_amd/lib/libggml-mi355.so MI355_GRAPH_STATS=1 timeout -k 10 300 $R/llama-bench -m /tmp/m.gguf -p 512 -n 128 -r 2 -ngl 99 -t 16 $args 2>&1 | grep -E "^\| llama|decode plans|eager" | cut -c1-200
done
