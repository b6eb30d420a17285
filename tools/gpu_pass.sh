cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
REF=oracle/_ref/avx2
G=/tmp/m.gguf
$REF/gguf_synth --preset 8b --ftype q4_k_m --out $G > gpurun_out/synth.log 2>&1 || { tail -3 gpurun_out/synth.log; exit 1; }
( export GGML_BACKEND_PATH=$PWD/llama.cpp.dsp_amd/lib/libggml-mi355.so MI355_GRAPH_STATS=1 MI355_TIMING=1
MI355Q_PLAN_VERBOSE=1 timeout -k 10 300 $REF/llama-bench -m $G -p 0 -n 512 -r 1 -ngl 99 -t 16 -o json > gpurun_out/lb_tg512.json 2> gpurun_out/lb_tg512.err; echo "tg512 rc=$?"
grep -h "avg_ts\|n_gen\"" gpurun_out/lb_tg512.json; grep -h "MI355 decode\|MI355 graph_compute host" gpurun_out/lb_tg512.err | cut -c1-400; grep -h "host time: desc\|MI355 plan compile" gpurun_out/lb_tg512.err | sed -n '1,3p;40,45p' | cut -c1-300
timeout -k 10 300 $REF/llama-bench -m $G -p 0 -n 128 -r 3 -ngl 99 -t 16 -o json > gpurun_out/lb_tg128.json 2> gpurun_out/lb_tg128.err; echo "tg128 rc=$?"
grep -h "avg_ts\|n_gen\"" gpurun_out/lb_tg128.json; grep -h "MI355 graph_compute host" gpurun_out/lb_tg128.err | cut -c1-400 )
timeout -k 10 500 python -m pytest tests/test_plugin.py -m gpu -q -x -k "whole_model_decode_equal or other_weight_recipes or suffix_nodes or libllama_flash or teacher or result_norm or timeout or flash_attention or two_devices" > gpurun_out/plan_tests.log 2>&1; echo "plugin plan tests rc=$?"; tail -4 gpurun_out/plan_tests.log | cut -c1-300
