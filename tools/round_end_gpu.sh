# Round-end GPU pass (run through gpurun): parity tests, smoke, bench, rocprofv3 kernel-trace and FETCH_SIZE passes, summaries under profiles/.
# Every step must succeed: a failing step stops the pass with its exit code (no "|| echo").
TAG=${TAG:-round2}
cd $GRAFT_REPO_ROOT || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
step() { echo "== $*"; }
if [ -z "$SKIP_TESTS" ]; then
  step "pytest -m gpu"
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_full.log 2>&1; rc=$?
  tail -3 gpurun_out/gpu_tests_full.log | cut -c1-300
  [ $rc -ne 0 ] && { tail -60 gpurun_out/gpu_tests_full.log | cut -c1-300; exit $rc; }
fi
step "smoke"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; rc=$?; tail -2 gpurun_out/smoke.log; [ $rc -ne 0 ] && exit $rc
step "bench.py (default)"
timeout -k 10 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; rc=$?
[ $rc -ne 0 ] && { tail -20 gpurun_out/bench_$TAG.err; exit $rc; }
cut -c1-400 gpurun_out/bench_$TAG.json
rm -rf gpurun_out/prof_kt gpurun_out/prof_pmc
step "rocprofv3 --kernel-trace --stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/prof_kt.log 2>&1; rc=$?
echo "rocprofv3 kernel-trace pass: exit $rc"; [ $rc -ne 0 ] && { tail -20 gpurun_out/prof_kt.log; exit $rc; }
step "rocprofv3 --pmc FETCH_SIZE (own pass)"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_pmc -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pp > gpurun_out/prof_pmc.log 2>&1; rc=$?
echo "rocprofv3 pmc pass: exit $rc"; [ $rc -ne 0 ] && { tail -20 gpurun_out/prof_pmc.log; exit $rc; }
python tools/summarize_profile.py $TAG gpurun_out/prof_kt gpurun_out/prof_pmc || exit 1
step "bench.py again: the roofline now quotes the traffic measured on these sources"
timeout -k 10 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; rc=$?
[ $rc -ne 0 ] && { tail -20 gpurun_out/bench_$TAG.err; exit $rc; }
cp gpurun_out/bench_$TAG.json profiles/${TAG}_bench.json
python3 - <<EOF
import json; j = json.load(open("gpurun_out/bench_$TAG.json"))
print("value", j["value"], j["unit"], "| roofline", {k: j["roofline"][k] for k in ("achieved", "frac", "traffic", "avg_launch_us")}, "| pp512", j.get("pp512", {}).get("value"),
      "| cpu", (j.get("cpu_baseline") or {}).get("value"), "| graph_compute", {k: (j.get("graph_compute") or {}).get(k) for k in ("value", "us_per_token")})
EOF
step "other configurations"
timeout -k 10 300 python bench.py --ftype Q8_0 --no-cpu-baseline > profiles/${TAG}_bench_llama8b_q8_0.json 2> gpurun_out/bench_q8.err; rc=$?; [ $rc -ne 0 ] && { tail gpurun_out/bench_q8.err; exit $rc; }
timeout -k 10 500 python bench.py --model mixtral-8x7b --no-cpu-baseline --steps 32 --warmup 4 > profiles/${TAG}_bench_mixtral8x7b.json 2> gpurun_out/bench_mx.err; rc=$?; [ $rc -ne 0 ] && { tail gpurun_out/bench_mx.err; exit $rc; }
timeout -k 10 500 python bench.py --model llama3-70b --no-cpu-baseline --steps 32 --warmup 4 > profiles/${TAG}_bench_llama70b_1gpu.json 2> gpurun_out/bench_70.err; rc=$?; [ $rc -ne 0 ] && { tail gpurun_out/bench_70.err; exit $rc; }
python3 - <<EOF
import json
for f in ("llama8b_q8_0", "mixtral8x7b", "llama70b_1gpu"):
    j = json.load(open("profiles/${TAG}_bench_%s.json" % f)); print(f, j["value"], j["unit"], "roofline", j["roofline"]["achieved"], "pp512", j.get("pp512", {}).get("value"), j.get("pp512", {}).get("TFLOPs"))
EOF
step "whole model through the plugin (oracle/_ref host side): plan, no plan, two devices through the scheduler"
export GGML_BACKEND_PATH=$GRAFT_REPO_ROOT/llama.cpp.dsp_amd/lib/libggml-mi355.so
MP=oracle/_ref/avx2/model_parity
MPS=oracle/_ref/scalar/model_parity
if [ -x $MP ] && [ -x $MPS ]; then
  # the reference against itself first (its scalar and AVX2 builds, CPU only): the spread two correct evaluations of this model have
  timeout -k 10 600 $MPS --preset 8b --layers 32 --vocab 128256 --tokens 4 --dump /tmp/m8b_scalar.bin > gpurun_out/m8b_dump_scalar.log 2>&1 || exit 1
  timeout -k 10 300 $MP  --preset 8b --layers 32 --vocab 128256 --tokens 4 --dump /tmp/m8b_avx2.bin   > gpurun_out/m8b_dump_avx2.log 2>&1 || exit 1
  {
    echo "# $TAG: Llama-3-8B-shaped model (32 layers, 128256-row output, random Q4_K_M weights) through ggml_backend_graph_compute of libggml-mi355.so"
    echo; echo "Harness: oracle/model_parity (reference libggml host side). Logits of the plugin against the reference CPU backend (AVX2 build), each step"
    echo "bounded by max(1e-3 / NMSE 1e-5, 3 x the spread between the reference's own scalar and AVX2 builds); then 128 timed decode steps."
    echo; echo '```'
    MI355_GRAPH_STATS=1 timeout -k 10 300 $MP --preset 8b --layers 32 --vocab 128256 --tokens 4 --check /tmp/m8b_avx2.bin --noise /tmp/m8b_scalar.bin --bench 128 2>&1 | grep -v "load_backend" ; echo "exit ${PIPESTATUS[0]}"
    echo "-- the same with the CPU backend run live beside it (CPU timing)"
    MI355_GRAPH_STATS=1 timeout -k 10 300 $MP --preset 8b --layers 32 --vocab 128256 --tokens 4 --check /tmp/m8b_avx2.bin --noise /tmp/m8b_scalar.bin --time-cpu --bench 64 2>&1 | grep -v "load_backend"; echo "exit ${PIPESTATUS[0]}"
    echo "-- MI355_NO_PLAN=1 (one launch per node / fused group, launch graphs)"
    MI355_NO_PLAN=1 MI355_GRAPH_STATS=1 timeout -k 10 300 $MP --preset 8b --layers 32 --vocab 128256 --tokens 4 --check /tmp/m8b_avx2.bin --noise /tmp/m8b_scalar.bin --bench 128 2>&1 | grep -v "load_backend"; echo "exit ${PIPESTATUS[0]}"
    echo "-- two devices (the card presented twice, MI355_DUP_DEVICES=2), layer split by ggml_backend_sched, boundary through cpy_tensor_async"
    MI355_DUP_DEVICES=2 MI355_GRAPH_STATS=1 timeout -k 10 300 $MP --preset 8b --layers 32 --vocab 128256 --tokens 4 --check /tmp/m8b_avx2.bin --noise /tmp/m8b_scalar.bin --bench 128 --devs MI355_0,MI355_1 --sched 2>&1 | grep -v "load_backend"; echo "exit ${PIPESTATUS[0]}"
    echo '```'
  } > profiles/${TAG}_plugin_whole_model.md
  cut -c1-300 profiles/${TAG}_plugin_whole_model.md
  grep -q "exit [1-9]" profiles/${TAG}_plugin_whole_model.md && { echo "a whole-model run failed"; exit 1; }
fi
step "plan timeline (diagnostic build), 2 layers and 32 layers"
if [ -f llama.cpp.dsp_amd/lib/libmi355q_dbg.so ]; then
  {
    echo "# $TAG: in-kernel timeline of the decode plan (libmi355q_dbg.so, tools/planstamps.py; Llama-3-8B Q4_K_M, position 100, n_kv 128)"
    echo; echo "Stages per layer: q|k (norm), v, ATTN, wo, gate|up (norm, SiLU x up), down.  Stamps: see tools/planstamps.py."
    echo; echo '```'
    MI355Q_LIB=$PWD/llama.cpp.dsp_amd/lib/libmi355q_dbg.so timeout -k 10 200 python tools/planstamps.py --layers 2 2>&1 | grep -v "Warn\|amdgpu.ids\|line.append" | cut -c1-330
    echo '```'
    echo; echo "Whole token (32 layers + output), per-launch time with HIP events (product library):"
    echo; echo '```'
    timeout -k 10 200 python tools/loaderonly.py 2>&1 | tail -1
    timeout -k 10 200 python tools/loaderonly.py --pos 250 --n-ctx 256 2>&1 | tail -1
    MI355Q_PLAN_KV_LDS=0 timeout -k 10 200 python tools/loaderonly.py 2>&1 | tail -1 | sed 's/^/round-2 attention stage (MI355Q_PLAN_KV_LDS=0): /'
    echo '```'
  } > profiles/${TAG}_plan_timeline.md
  tail -8 profiles/${TAG}_plan_timeline.md | cut -c1-200
fi
step "prefill shapes"
timeout -k 10 300 python tools/ppbench.py > gpurun_out/ppbench.log 2>&1; rc=$?; tail -9 gpurun_out/ppbench.log; [ $rc -ne 0 ] && exit $rc
cp gpurun_out/ppbench.log profiles/${TAG}_ppbench.txt
cp profiles/${TAG}_* gpurun_out/
exit 0
