#!/usr/bin/env python3
"""Throughput of the standalone activation quantizer (dev tool): is q8k_wave itself slow?"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "llama.cpp.dsp_amd")]
import torch
import ggml_mi355 as g
for act in (g.Q8_K, g.Q8_0):
    for n, k in ((8192, 4096), (64, 4096), (1, 4096), (1, 14336)):
        x = torch.randn((n, k), dtype=torch.float32, device="cuda")
        for _ in range(3): g.quantize_act(act, x)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        out = g.quantize_act(act, x)
        with torch.cuda.graph(gr):
            for _ in range(20): g.lib().mi355q_quantize_act(act, x.data_ptr(), x.stride(0) * 4, out.data_ptr(), n, k, 0, int(torch.cuda.current_stream().cuda_stream))
        gr.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): gr.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 400
        spans = n * k / 256
        print(f"{g.TYPE_NAMES[act]} n={n} k={k}: {dt*1e6:8.2f} us/launch  {n*k*4/dt/1e9:8.1f} GB/s in   {dt*1e9/max(1,spans/ (256*16)):8.1f} ns per span-slot(256CUx16waves)")
