cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mfma_tier or full_size or prefill or multi" > gpurun_out/gpu_tests_mmq.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_mmq.log | cut -c1-300
[ $rc -ne 0 ] && { tail -60 gpurun_out/gpu_tests_mmq.log | cut -c1-300; exit 1; }
for wide in 0 -1 1; do
echo "== MI355Q_MMQ_BF16_WIDE=$wide"
if [ $wide = -1 ]; then unset MI355Q_MMQ_BF16_WIDE; else export MI355Q_MMQ_BF16_WIDE=$wide; fi
timeout -k 10 120 python tools/pp_one.py q6_k 4096 14336 512 2>/dev/null
timeout -k 10 120 python tools/pp_one.py q6_k 128256 4096 512 2>/dev/null
timeout -k 10 120 python tools/pp_one.py q5_k 14336 4096 512 2>/dev/null
timeout -k 10 120 python tools/pp_one.py q4_0 14336 4096 512 2>/dev/null
timeout -k 10 120 python tools/pp_one.py q6_k 4096 14336 2048 2>/dev/null
done
exit 0
