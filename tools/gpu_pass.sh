cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_plan.py -q > gpurun_out/plan_tests.log 2>&1; rc=$?; echo "plan tests rc=$rc"
grep -E 'sync words|passed|failed|FAILED' gpurun_out/plan_tests.log | cut -c1-900 | head -30
MI355Q_LIB=$PWD/llama.cpp.dsp_amd/lib/libmi355q_dbg.so timeout -k 10 300 python tools/planstamps.py --layers 2 > gpurun_out/stamps_v4.txt 2>&1; echo "stamps rc=$?"
grep '^stage' gpurun_out/stamps_v4.txt | head -24 | tail -6 | cut -c1-420
grep -A14 "step-loop" gpurun_out/stamps_v4.txt
timeout -k 10 500 python bench.py --steps 20 --warmup 5 --no-pp --no-plugin --no-cpu-baseline > gpurun_out/bench_v4.json 2> gpurun_out/bench_v4.err; echo "bench rc=$?"
tail -3 gpurun_out/bench_v4.err | cut -c1-300
cut -c1-330 gpurun_out/bench_v4.json
