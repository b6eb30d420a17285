#!/usr/bin/env python3
"""One prefill shape in a loop (for rocprofv3 counter passes).  python tools/pp_one.py [type m k n reps]"""
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
for _ in range(reps):
    g.mul_mat(w, x, out=y)
torch.cuda.synchronize()
