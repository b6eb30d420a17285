#!/usr/bin/env python3
"""Dev tool (GPU box): the plan's attention stage with a flash-layout cache against oracle.glue.flash_attn_ext (bit-exact with the reference CPU
backend): how many output elements differ, and by how much.  Usage: python tools/fa_exact_check.py [n_kv] [pos]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "llama.cpp.dsp_amd"), str(ROOT / "tests")]
import numpy as np, torch
import ggml_mi355 as G
import oracle
from oracle import glue
from qdata import quantized_weights

n_kv = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pos = int(sys.argv[2]) if len(sys.argv) > 2 else 201
n_head, n_head_kv, hd = 32, 8, 128
rng = np.random.default_rng(77)
n_q, n_k, n_ctx = n_head * hd, n_head_kv * hd, n_kv
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
q = dev(rng.standard_normal((1, n_q)).astype(np.float32)); k = dev(rng.standard_normal((1, n_k)).astype(np.float32)); v = dev(rng.standard_normal((1, n_k)).astype(np.float32))
kc_h = rng.standard_normal((n_ctx, n_k)).astype(np.float16); vc_h = rng.standard_normal((n_ctx, n_k)).astype(np.float16)
mask_h = np.full(n_kv, -np.inf, np.float16); mask_h[:pos + 1] = 0.0
scale = 1.0 / np.sqrt(hd)
posd = dev(np.array([pos], np.int32))
w0 = G.QWeight.from_host(oracle.Q4_K, quantized_weights(oracle.Q4_K, 32, 256, rng), 32, 256)
x0 = torch.zeros((1, 256), dtype=torch.float32, device="cuda"); y0 = torch.zeros((1, 32), dtype=torch.float32, device="cuda")
kc, vc = dev(kc_h), dev(vc_h)
out = torch.zeros((1, n_q), dtype=torch.float32, device="cuda")
dst = torch.tensor([kc.data_ptr() + pos * n_k * 2, vc.data_ptr() + pos * n_k * 2], dtype=torch.int64, device="cuda")
attn = dict(q=q, k=k, v=v, pos=posd, rope=dict(n_dims=hd, mode=0, n_ctx_orig=8192, freq_base=500000.0), k_cache=kc, v_cache=vc,
            k_nb_pos=n_k * 2, k_nb_head=hd * 2, v_nb_pos=n_k * 2, v_nb_dim=2, v_nb_head=hd * 2, k_dst=dst[0:1], v_dst=dst[1:2], v_dst_nb=2,
            mask=dev(mask_h), n_head=n_head, n_head_kv=n_head_kv, head_dim=hd, n_kv=n_kv, scale=scale, out=out)
plan = G.Plan([([w0], x0, [y0], False), dict(attn=attn)])
plan.run(); torch.cuda.synchronize()
assert plan.status() == 0
got = out.reshape(n_head, hd).cpu().numpy()
kc_a, vc_a = kc.cpu().numpy(), vc.cpu().numpy()
# q as the device roped it: rope the f32 q with the oracle, then the oracle rounds it to f16 itself
q_r = glue.rope(q.cpu().numpy().reshape(1, 1, n_head, hd), np.array([pos], np.int32), hd, 0, freq_base=500000.0, n_ctx_orig=8192).reshape(n_head, hd)
K = kc_a[:n_kv].reshape(1, n_kv, n_head_kv, hd).transpose(0, 2, 1, 3); V = vc_a[:n_kv].reshape(1, n_kv, n_head_kv, hd).transpose(0, 2, 1, 3)
m2 = np.zeros((64, n_kv), np.float16); m2[0] = mask_h
ref = glue.flash_attn_ext(q_r.reshape(1, n_head, 1, hd), K, V, m2, scale).reshape(n_head, hd)
d = np.abs(got.astype(np.float64) - ref); top = np.abs(ref).max()
nb = (got.view(np.uint32) != ref.view(np.uint32))
print(f"n_kv {n_kv} pos {pos}: {nb.sum()} of {nb.size} elements differ in bits; max |d|/max|ref| {d.max() / top:.3e}; > 1e-6: {(d > 1e-6 * top).sum()}; > 1e-4: {(d > 1e-4 * top).sum()}")
heads = np.where(nb.any(axis=1))[0]
print("heads with differences:", heads[:16], "...")
