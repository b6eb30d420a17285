#!/usr/bin/env python3
"""Time one prefill shape (dev tool; also the loop for rocprofv3 counter passes):  python tools/pp_one.py <type> <M> <K> [N=512] [reps=20]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "llama.cpp.dsp_amd")]
import torch
import ggml_mi355 as g
from ggml_mi355.workloads import MatSpec
from bench import device_random_weight
dev = torch.device("cuda", 0)
t = {n.lower(): v for n, v in vars(g).items() if n[:1] in "QI" and isinstance(v, int)}[sys.argv[1].lower()]
m, k = int(sys.argv[2]), int(sys.argv[3]); N = int(sys.argv[4]) if len(sys.argv) > 4 else 512; reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
w = device_random_weight(torch, g, MatSpec("w", t, m, k, 0), dev)
x = torch.randn((N, k), dtype=torch.float32, device=dev); y = torch.empty((N, m), dtype=torch.float32, device=dev)
g.mul_mat(w, x, out=y); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): g.mul_mat(w, x, out=y)
e1.record(); torch.cuda.synchronize()
dt = e0.elapsed_time(e1) * 1e-3 / reps
print(f"{sys.argv[1]} {m}x{k} N={N}: {dt*1e6:9.1f} us  {2*m*N*k/dt/1e12:7.1f} TFLOP/s", flush=True)
