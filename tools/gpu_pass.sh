cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mul_mat_id or mixtral or mfma_tier or moe or q8_0" > gpurun_out/gpu_tests_moe.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_moe.log | cut -c1-300
[ $rc -ne 0 ] && { tail -60 gpurun_out/gpu_tests_moe.log | cut -c1-300; exit 1; }
timeout -k 10 300 python -m pytest tests/test_plugin.py -m gpu -x -q -k "moe or MUL_MAT_ID or backend_ops" 2>&1 | tail -2
rm -rf gpurun_out/prof_mx
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_mx -- python3 bench.py --model mixtral-8x7b --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/prof_mx.log 2>&1; echo "rocprof exit $?"
python3 -c "
import json
for ln in open('gpurun_out/prof_mx.log'):
    if ln.startswith('{\"metric\"'):
        j=json.loads(ln); print(j['value'], j['pp512']['value'], j['pp512']['TFLOPs'])"
python3 tools/trace_top.py gpurun_out/prof_mx 14
exit 0
