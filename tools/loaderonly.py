#!/usr/bin/env python3
"""Dev tool (GPU box): time the decode plan's launch with MI355Q_PLAN_LOADER_ONLY=1 (the loader wave alone streams every stage's weights through
the LDS ring, the consumers leave at once) and without.  Usage: python tools/loaderonly.py [--layers 32] [--pos 100] [--n-ctx 128]"""
import argparse, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "llama.cpp.dsp_amd")]
import torch
import ggml_mi355 as g
from ggml_mi355 import workloads as wl
import bench as B

ap = argparse.ArgumentParser(); ap.add_argument("--layers", type=int, default=32); ap.add_argument("--pos", type=int, default=100); ap.add_argument("--n-ctx", type=int, default=128)
a = ap.parse_args()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
cfg = dict(wl.LLAMA3_8B)
specs = [s for s in wl.llama_matmuls(cfg, "Q4_K_M") if s.layer < a.layers]
stage = B.Stage(torch, g, specs, True, dev)
act = torch.randn((1, cfg["n_embd"]), dtype=torch.float32, device=dev)
plan = stage.make_decode_plan(cfg, act, a.n_ctx, False)
stage.set_token(a.pos)
for _ in range(3):
    plan.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    plan.run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"{'loader only' if os.environ.get('MI355Q_PLAN_LOADER_ONLY') else 'full plan'}: {ms * 1e3:.1f} us per launch, {plan.weight_bytes / ms * 1e-6:.0f} GB/s, status {plan.status()}")
