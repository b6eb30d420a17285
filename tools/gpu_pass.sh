cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export GGML_BACKEND_PATH=$PWD/llama.cpp.dsp_amd/lib/libggml-mi355.so LD_LIBRARY_PATH=$PWD/oracle/_ref/avx2:$LD_LIBRARY_PATH
timeout -k 10 400 oracle/_ref/avx2/test-backend-ops test -b MI355_0 -o FLASH_ATTN_EXT > gpurun_out/tbo_fa.log 2>&1; echo rc=$?
grep -c "type_KV=q8_0.*OK" gpurun_out/tbo_fa.log; grep -c "type_KV=q4_0.*OK" gpurun_out/tbo_fa.log; grep "FAIL" gpurun_out/tbo_fa.log | sed 's/\x1b\[[0-9;]*m//g' | cut -c1-220 | head -5; tail -2 gpurun_out/tbo_fa.log | cut -c1-100
R=oracle/_ref/avx2
$R/gguf_synth --preset 8b --ftype q4_k_m --out /tmp/m.gguf > /dev/null 2>&1
for args in "-fa 1 -ctk q8_0 -ctv q8_0" "-fa 1 -ctk q4_0 -ctv q4_0"; do
  echo "== llama-bench $args"
  MI355_GRAPH_STATS=1 timeout -k 10 300 $R/llama-bench -m /tmp/m.gguf -p 512 -n 128 -r 2 -ngl 99 -t 16 $args 2>&1 | grep -E "^\| llama" | cut -c1-200
done
unset GGML_BACKEND_PATH
timeout -k 10 300 python -m pytest tests/test_gpu_glue.py -q > gpurun_out/glue_tests.log 2>&1; echo "glue rc=$?"; tail -2 gpurun_out/glue_tests.log
