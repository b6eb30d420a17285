cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in 14 18 28; do echo "cfg $cfg"; MI355Q_Q80_CFG=$cfg timeout -k 10 120 python tools/pp_one.py q8_0 14336 4096 512 2>/dev/null; MI355Q_Q80_CFG=$cfg timeout -k 10 120 python tools/pp_one.py q8_0 4096 4096 512 2>/dev/null; MI355Q_Q80_CFG=$cfg timeout -k 10 120 python tools/pp_one.py q8_0 4096 14336 512 2>/dev/null; done
for cfg in 14 18 28; do MI355Q_Q80_CFG=$cfg timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "q8_0" 2>&1 | tail -2; done
exit 0
