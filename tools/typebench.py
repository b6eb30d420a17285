#!/usr/bin/env python3
"""GEMV (N = 1) streaming rate per weight type on a 14336 x 4096 matrix (dev tool, GPU box): which tier serves the type and at how many GB/s.
Usage: python tools/typebench.py [out.md]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "llama.cpp.dsp_amd"), str(ROOT / "tests")]
import numpy as np, torch
import ggml_mi355 as g
from qdata import random_blocks
M, K = 14336, 4096
rng = np.random.default_rng(0)
rows = []
for t in g.WEIGHT_TYPES:
    w = g.QWeight.from_host(t, random_blocks(t, M, K, rng), M, K)
    x = torch.randn((1, K), dtype=torch.float32, device="cuda"); y = torch.empty((1, M), dtype=torch.float32, device="cuda")
    for _ in range(3): g.mul_mat(w, x, out=y)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10): g.mul_mat(w, x, out=y)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    rows.append((g.TYPE_NAMES[t], "planar streaming (k_gemv_fast)" if g.is_planar(t, K) else "canonical rows (k_gemv_generic)", w.nbytes / 1e6, us, w.nbytes / us / 1e3))
out = ["| type | tier at K = 4096 | MB | us / launch | GB/s |", "|---|---|---|---|---|"] + [f"| {n} | {tier} | {mb:.1f} | {us:.1f} | {gb:.0f} |" for n, tier, mb, us, gb in rows]
text = f"GEMV N = 1, {M} x {K}, one launch per matmul (hipGraph of 10), HIP events:\n\n" + "\n".join(out) + "\n"
print(text)
if len(sys.argv) > 1: Path(sys.argv[1]).write_text(text)
