#!/usr/bin/env python3
"""Per-shape micro-benchmark of the GEMV tier (development tool, GPU box only).

Each case is launched over enough distinct weight copies to defeat the 256 MiB Infinity Cache, the
launch sequence is captured in a hipGraph and replayed; reported time per launch includes the ~1.5 us
kernel boundary.  Usage: python tools/mbench.py [case-substring]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "llama.cpp.dsp_amd")]
import torch
import ggml_mi355 as g
from ggml_mi355.workloads import MatSpec
from bench import device_random_weight

CASES = {
    "wq_q4k_4096x4096":        [(g.Q4_K, 4096, 4096)],
    "qkv_q4k":                 [(g.Q4_K, 4096, 4096), (g.Q4_K, 1024, 4096), (g.Q4_K, 1024, 4096)],
    "qkv_mixed":               [(g.Q4_K, 4096, 4096), (g.Q4_K, 1024, 4096), (g.Q6_K, 1024, 4096)],
    "gateup_q4k":              [(g.Q4_K, 14336, 4096), (g.Q4_K, 14336, 4096)],
    "down_q4k_4096x14336":     [(g.Q4_K, 4096, 14336)],
    "down_q6k_4096x14336":     [(g.Q6_K, 4096, 14336)],
    "output_q6k_128256x4096":  [(g.Q6_K, 128256, 4096)],
    "q8_0_14336x4096":         [(g.Q8_0, 14336, 4096)],
}

def main():
    filt = sys.argv[1] if len(sys.argv) > 1 else ""
    ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = torch.device("cuda", 0)
    for name, mats in CASES.items():
        if filt not in name:
            continue
        nbytes = sum(g.row_size(t, k) * m for t, m, k in mats)
        copies = max(2, int(600e6 // nbytes) + 1)
        sets = [[device_random_weight(torch, g, MatSpec(name, t, m, k, 0), dev) for t, m, k in mats] for _ in range(copies)]
        x = torch.randn((ncols, mats[0][2]), dtype=torch.float32, device=dev)
        ys = [torch.empty((ncols, m), dtype=torch.float32, device=dev) for _, m, _ in mats]
        def run():
            for ws in sets:
                g.mul_mat_multi(ws, x, outs=ys)
        run(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            run()
        gr.replay(); torch.cuda.synchronize()
        reps = max(3, int(2000 // copies))
        t0 = time.perf_counter()
        for _ in range(reps):
            gr.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (reps * copies)
        print(f"{name:28s} N={ncols} {nbytes/1e6:8.1f} MB  {dt*1e6:8.2f} us/launch  {nbytes/dt/1e9:8.1f} GB/s  ({copies} copies)", flush=True)
        del sets

if __name__ == "__main__":
    main()
