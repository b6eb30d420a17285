#!/usr/bin/env python3
"""Sweep M for fixed (type, K): time = a + b*M separates fixed launch cost from streaming rate (dev tool)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "llama.cpp.dsp_amd")]
import torch
import ggml_mi355 as g
from ggml_mi355.workloads import MatSpec
from bench import device_random_weight

def run(t, k, ms, ncols=1):
    dev = torch.device("cuda", 0)
    for m in ms:
        nbytes = g.row_size(t, k) * m
        copies = max(2, int(600e6 // nbytes) + 1)
        sets = [device_random_weight(torch, g, MatSpec("w", t, m, k, 0), dev) for _ in range(copies)]
        x = torch.randn((ncols, k), dtype=torch.float32, device=dev)
        y = torch.empty((ncols, m), dtype=torch.float32, device=dev)
        def go():
            for w in sets:
                g.mul_mat(w, x, out=y)
        go(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            go()
        gr.replay(); torch.cuda.synchronize()
        reps = max(3, int(3000 // copies))
        t0 = time.perf_counter()
        for _ in range(reps):
            gr.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (reps * copies)
        print(f"{g.TYPE_NAMES[t]:6s} K={k:6d} M={m:7d} N={ncols} {nbytes/1e6:8.1f} MB {dt*1e6:8.2f} us {nbytes/dt/1e9:8.1f} GB/s", flush=True)
        del sets

if __name__ == "__main__":
    ncols = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    for t in (g.Q4_K, g.Q6_K, g.Q8_0):
        run(t, 4096, [512, 2048, 4096, 8192, 16384, 32768, 65536, 131072], ncols)
    for t in (g.Q4_K, g.Q6_K):
        run(t, 14336, [1024, 4096, 16384, 32768], ncols)
