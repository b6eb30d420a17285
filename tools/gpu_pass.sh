cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 200 python tools/fa_op_debug.py flash_attn_batch3_d64_kv40 2>&1 | grep -v amdgpu.ids | head -1
timeout -k 10 200 python tools/fa_op_debug.py flash_attn_decode_gqa_d128_kv256 2>&1 | grep -v amdgpu.ids | head -1
timeout -k 10 200 python tools/fa_exact_check.py 256 201 2>&1 | tail -2
timeout -k 10 200 python tools/fa_exact_check.py 64 40 2>&1 | tail -2
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_full.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_full.log | cut -c1-300
[ $rc -ne 0 ] && tail -40 gpurun_out/gpu_tests_full.log | cut -c1-300
exit $rc
