#!/usr/bin/env python3
"""Time one prefill shape (dev tool): python tools/pp_shape.py [type m k n reps]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "llama.cpp.dsp_amd")]
import torch
import ggml_mi355 as g
from ggml_mi355.workloads import MatSpec
from bench import device_random_weight
t = {"q4_K": g.Q4_K, "q6_K": g.Q6_K, "q8_0": g.Q8_0}[sys.argv[1]] if len(sys.argv) > 1 else g.Q4_K
m, k, n, reps = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (14336, 4096, 512, 20)
dev = torch.device("cuda", 0)
w = device_random_weight(torch, g, MatSpec("w", t, m, k, 0), dev)
x = torch.randn((n, k), dtype=torch.float32, device=dev)
y = torch.empty((n, m), dtype=torch.float32, device=dev)
for _ in range(3): g.mul_mat(w, x, out=y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): g.mul_mat(w, x, out=y)
e1.record(); torch.cuda.synchronize()
dt = e0.elapsed_time(e1) * 1e-3 / reps
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'q4_K'} {m}x{k} N={n}: {dt*1e6:9.1f} us  {2*m*n*k/dt/1e12:7.1f} TFLOP/s", flush=True)
import os
if os.environ.get("MI355Q_I8_STAMPS"):
    print("stamps (us, wave 0 of workgroup 0): wait+barrier %.1f  dequant %.1f  tiles %.1f  epilogue %.1f" % tuple(y[0, :4].tolist()))
