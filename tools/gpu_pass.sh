cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_plan.py -m gpu -x -q > gpurun_out/gpu_tests_plan.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_plan.log | cut -c1-300
[ $rc -ne 0 ] && { tail -60 gpurun_out/gpu_tests_plan.log | cut -c1-300; exit 1; }
for rep in 1 2; do for which in prev new; do
if [ $which = prev ]; then export MI355Q_LIB=$GRAFT_REPO_ROOT/tools/micro/libmi355q_prev.so; else unset MI355Q_LIB; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-pp > gpurun_out/bench_$which.json 2> gpurun_out/bench_$which.err; python3 -c "
import json; j=json.load(open('gpurun_out/bench_$which.json')); print('$which', j['value'], j['ms_per_step'], j['roofline']['avg_launch_us'])"
done; done
unset MI355Q_LIB
timeout -k 10 900 python -m pytest tests/test_plugin.py -m gpu -x -q -k "whole_model or decode_layer" > gpurun_out/gpu_tests_plugin.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_plugin.log | cut -c1-300
[ $rc -ne 0 ] && { tail -60 gpurun_out/gpu_tests_plugin.log | cut -c1-300; exit 1; }
exit 0
