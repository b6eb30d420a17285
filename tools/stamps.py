#!/usr/bin/env python3
"""In-kernel timeline of the GEMV kernel (diagnostic build libmi355q_dbg.so; dev tool, GPU box only).
stamps: 0 kernel entry, 1 before ring prime, 2 after prime issued, 3 activations in LDS (after barrier),
4 first item consumed, 5 wave done.   Usage: MI355Q_LIB=.../libmi355q_dbg.so python tools/stamps.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "llama.cpp.dsp_amd")]
import numpy as np, torch
import ggml_mi355 as g
from ggml_mi355.workloads import MatSpec
from bench import device_random_weight

dev = torch.device("cuda", 0)
L = g.lib()
buf = torch.zeros(1024 * 16 * 8, dtype=torch.int64, device=dev)
import ctypes
L.mi355q_debug_set_stamps.argtypes = [ctypes.c_void_p]      # a bare Python int would be truncated to 32 bits
assert L.mi355q_debug_set_stamps(buf.data_ptr()) == 0
for name, (t, m, k) in {"wq q4k 4096x4096": (g.Q4_K, 4096, 4096), "gate q4k 14336x4096": (g.Q4_K, 14336, 4096),
                         "down q4k 4096x14336": (g.Q4_K, 4096, 14336), "down q6k 4096x14336": (g.Q6_K, 4096, 14336)}.items():
    ws = [device_random_weight(torch, g, MatSpec("w", t, m, k, 0), dev) for _ in range(12)]
    x = torch.randn((1, k), dtype=torch.float32, device=dev)
    y = torch.empty((1, m), dtype=torch.float32, device=dev)
    for w in ws[:-1]:
        g.mul_mat(w, x, out=y)
    buf.zero_(); torch.cuda.synchronize()
    g.mul_mat(ws[-1], x, out=y); torch.cuda.synchronize()
    s = buf.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    rel = (s[:, :6] - t0) * 10.0 / 1000.0        # us (100 MHz clock)
    print(f"{name}: {len(s)} waves")
    for i, lab in enumerate(["entry", "pre-prime", "primed", "acts in LDS", "1st item done", "wave done"]):
        col = rel[:, i][s[:, i] > 0]
        if col.size == 0:
            continue
        print(f"   {lab:14s} min {col.min():6.2f}  med {np.median(col):6.2f}  p90 {np.percentile(col, 90):6.2f}  max {col.max():6.2f} us")
