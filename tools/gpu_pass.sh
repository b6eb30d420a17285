cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/gpu_tests_parity.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_parity.log | cut -c1-300
[ $rc -ne 0 ] && { tail -60 gpurun_out/gpu_tests_parity.log | cut -c1-300; exit 1; }
timeout -k 10 900 python -m pytest tests/test_plugin.py -m gpu -x -q -k "backend_ops or moe" > gpurun_out/gpu_tests_plugin.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_plugin.log | cut -c1-300
[ $rc -ne 0 ] && { tail -60 gpurun_out/gpu_tests_plugin.log | cut -c1-300; exit 1; }
python3 - <<'PY'
import sys
sys.path[:0] = ['.', 'llama.cpp.dsp_amd', 'tests']
import numpy as np, torch
import ggml_mi355 as g, oracle
from qdata import quantized_weights
rng = np.random.default_rng(1)
for t in (oracle.IQ4_XS, oracle.IQ4_NL):
    M, K, N = 14336, 4096, 512
    w = quantized_weights(t, 256, K, rng); w = np.tile(w, (M // 256, 1))
    W = g.QWeight.from_host(t, w, M, K)
    x = torch.randn((N, K), device='cuda'); y = torch.empty((N, M), device='cuda')
    for flags, name in ((0, 'bf16 MFMA tier'), (4, 'GEMV columns')):
        g.mul_mat(W, x, out=y, flags=flags); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.mul_mat(W, x, out=y, flags=flags)
        e1.record(); torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) * 1e-3 / 5
        print(f"{oracle.TYPE_NAMES[t]:8s} 14336x4096 N=512 {name:14s}: {dt*1e6:10.1f} us  {2*M*N*K/dt/1e12:7.2f} TFLOP/s", flush=True)
PY
exit 0
