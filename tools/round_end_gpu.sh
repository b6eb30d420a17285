# Round-end GPU pass (run through gpurun): parity tests, smoke, bench, rocprofv3 kernel-trace and FETCH_SIZE passes, summaries.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_full.log 2>&1 || { tail -30 gpurun_out/gpu_tests_full.log; exit 1; }; fi
[ -f gpurun_out/gpu_tests_full.log ] && tail -3 gpurun_out/gpu_tests_full.log || true
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 400 python bench.py > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.err
cut -c1-300 gpurun_out/bench_r1.json
rm -rf gpurun_out/prof_kt gpurun_out/prof_pmc
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/prof_kt.log 2>&1 || echo "rocprofv3 kernel-trace pass exited with $? (its CSVs are written before the profiler's exit-time crash with cooperative launches)"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_pmc -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/prof_pmc.log 2>&1 || echo "rocprofv3 pmc pass exited with $?"
cp gpurun_out/bench_r1.json profiles/round1_bench.json
python tools/summarize_profile.py round1 gpurun_out/prof_kt gpurun_out/prof_pmc
# one resident decoder layer through the plugin: timing with / without launch graphs and fusions, and its kernel timeline
export GGML_BACKEND_PATH=$GRAFT_REPO_ROOT/llama.cpp.dsp_amd/lib/libggml-mi355.so
LP=oracle/_ref/avx2/layer_parity
if [ -x $LP ]; then
  for mode in default MI355_NO_GRAPHS MI355_NO_FUSION; do
    echo "== layer_parity 8b ($mode)"
    if [ $mode = default ]; then MI355_GRAPH_STATS=1 timeout -k 10 200 $LP 1 MI355_0 8b 8 300 2>&1 | grep -v "^step\|load_backend"
    else env $mode=1 MI355_GRAPH_STATS=1 timeout -k 10 200 $LP 1 MI355_0 8b 8 300 2>&1 | grep -v "^step\|load_backend"; fi
  done > gpurun_out/layer_modes.log 2>&1
  cat gpurun_out/layer_modes.log
  rm -rf gpurun_out/prof_layer
  MI355_NO_GRAPHS=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_layer -- $LP 1 MI355_0 8b 2 50 > gpurun_out/prof_layer.log 2>&1 || echo "rocprofv3 layer pass exited with $?"
  python tools/summarize_layer_trace.py round1 gpurun_out/prof_layer || true
fi
timeout -k 10 300 python tools/ppbench.py > gpurun_out/ppbench.log 2>&1 || true
tail -8 gpurun_out/ppbench.log
cp profiles/round1_* gpurun_out/ 
