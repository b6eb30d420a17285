#!/usr/bin/env python3
"""Prefill (pp512) benchmark of the MFMA tier (dev tool + numbers for DESIGN.md): per-shape TFLOP/s at N=512 and a
whole Llama-3-8B Q4_K_M pp512 pass over all 225 quantized matmuls (15.0 GFLOP/token)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "llama.cpp.dsp_amd")]
import torch
import ggml_mi355 as g
from ggml_mi355 import workloads as wl
from ggml_mi355.workloads import MatSpec
from bench import device_random_weight

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512

def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps

for name, (t, m, k) in {"q4_K 4096x4096": (g.Q4_K, 4096, 4096), "q4_K 14336x4096": (g.Q4_K, 14336, 4096),
                         "q4_K 4096x14336": (g.Q4_K, 4096, 14336), "q6_K 4096x14336": (g.Q6_K, 4096, 14336),
                         "q8_0 14336x4096": (g.Q8_0, 14336, 4096), "q6_K 128256x4096": (g.Q6_K, 128256, 4096)}.items():
    w = device_random_weight(torch, g, MatSpec("w", t, m, k, 0), dev)
    x = torch.randn((N, k), dtype=torch.float32, device=dev)
    y = torch.empty((N, m), dtype=torch.float32, device=dev)
    dt = timeit(lambda: g.mul_mat(w, x, out=y), 10)
    print(f"{name:18s} N={N}: {dt*1e6:9.1f} us  {2*m*N*k/dt/1e12:7.1f} TFLOP/s", flush=True)
    del w

specs = wl.llama_matmuls(wl.LLAMA3_8B, "Q4_K_M")
ws = [device_random_weight(torch, g, s, dev) for s in specs]
xs = {k: torch.randn((N, k), dtype=torch.float32, device=dev) for k in (4096, 14336)}
ys = {m: torch.empty((N, m), dtype=torch.float32, device=dev) for m in {s.M for s in specs}}
def full():
    for w in ws: g.mul_mat(w, xs[w.K], out=ys[w.M])
dt = timeit(full, 3)
flop = sum(2 * s.M * s.K for s in specs) * N
print(f"Llama-3-8B Q4_K_M pp{N} (225 quantized matmuls): {dt*1e3:.2f} ms  {N/dt:.0f} tok/s  {flop/dt/1e12:.1f} TFLOP/s")
