#!/usr/bin/env python3
"""Dev tool (GPU box): mi355q_op_flash_attn_ext against a golden fixture, element by element."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "llama.cpp.dsp_amd"), str(ROOT / "tests")]
import numpy as np, torch
import ggml_mi355 as G
from oracle import glue
name = sys.argv[1] if len(sys.argv) > 1 else "flash_attn_batch3_d64_kv40"
g = np.load(ROOT / "tests" / "golden" / f"{name}.npz", allow_pickle=False)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
def run(q, k, v, mask):
    return G.op_flash_attn_ext(dev(q), dev(k), dev(v), dev(mask), float(g["scale"]), float(g["max_bias"]), float(g["softcap"])).cpu().numpy()
def report(tag, y, want):
    nb = y.view(np.uint32) != want.view(np.uint32)
    d = np.abs(y.astype(np.float64) - want)
    print(tag, "differ", int(nb.sum()), "of", nb.size, "max", d.max() / np.abs(want).max(), "rows (t,h) with differences:", sorted({(int(a[1]), int(a[2])) for a in np.argwhere(nb)})[:12])
q, k, v, mask, want = g["q"], g["k"], g["v"], g["mask"], g["y"]
report("fixture      ", run(q, k, v, mask), want)
k0 = np.nan_to_num(k.astype(np.float32)).astype(np.float16); v0 = np.nan_to_num(v.astype(np.float32)).astype(np.float16)
w0 = glue.flash_attn_ext(q, k0, v0, mask, float(g["scale"]), float(g["max_bias"]), float(g["softcap"]))
print("oracle with NaN rows zeroed equals fixture:", np.array_equal(w0.view(np.uint32), want.view(np.uint32)))
report("NaNs zeroed  ", run(q, k0, v0, mask), want)
for t in range(q.shape[2]):
    m1 = np.zeros_like(mask); m1[0] = mask[t]
    report(f"row {t} alone  ", run(q[:, :, t:t + 1], k0, v0, m1), want[:, t:t + 1])
