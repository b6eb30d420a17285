cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/gpu_tests_parity.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_parity.log | cut -c1-300
[ $rc -ne 0 ] && { tail -60 gpurun_out/gpu_tests_parity.log | cut -c1-300; exit 1; }
timeout -k 10 900 python -m pytest tests/test_plugin.py -m gpu -x -q -k "backend_ops" > gpurun_out/gpu_tests_plugin.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests_plugin.log | cut -c1-300
[ $rc -ne 0 ] && { tail -60 gpurun_out/gpu_tests_plugin.log | cut -c1-300; exit 1; }
python3 - <<'PY'
import sys, time
sys.path[:0] = ['.', 'llama.cpp.dsp_amd']
import numpy as np, torch
import ggml_mi355 as g, oracle
sys.path.insert(0, 'tests')
from qdata import quantized_weights
rng = np.random.default_rng(1)
for t in (oracle.Q3_K, oracle.Q2_K, oracle.IQ3_S, oracle.IQ2_XS, oracle.Q5_0, oracle.IQ1_S):
    M, K, N = 4096, 4096, 512
    w = quantized_weights(t, 256, K, rng); w = np.tile(w, (M // 256, 1))
    W = g.QWeight.from_host(t, w, M, K)
    x = torch.randn((N, K), device='cuda'); y = torch.empty((N, M), device='cuda')
    for flags, name in ((0, 'batched'), (4, 'per-column')):
        g.mul_mat(W, x, out=y, flags=flags); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5 if flags == 0 else 1
        e0.record()
        for _ in range(reps): g.mul_mat(W, x, out=y, flags=flags)
        e1.record(); torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) * 1e-3 / reps
        print(f"{oracle.TYPE_NAMES[t]:8s} 4096x4096 N=512 {name:10s}: {dt*1e6:10.1f} us  {2*M*N*K/dt/1e12:7.2f} TFLOP/s", flush=True)
PY
exit 0
