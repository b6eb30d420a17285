#!/usr/bin/env python3
"""Generate tests/golden/flash_attn_*.npz from the REAL reference: GGML_OP_FLASH_ATTN_EXT with an F16 K / V cache computed by the reference CPU
backend (ggml-cpu/ops.cpp:6686-6905) through oracle/_ref/avx2 (refshim ref_flash_attn_ext).  (tests/test_oracle_glue.py checks the restatement against both builds; the scalar one differs in the last place.)  Run where /root/reference exists:  python tests/golden/make_flash_attn_golden.py
Inputs are seeded gaussians; arrays only (loadable with allow_pickle=False)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402

OUT = Path(__file__).resolve().parent
# tag: (n_head, n_head_kv, n_queries, head_dim, n_kv, first masked position of query 0 (causal from there), max_bias, logit_softcap)
CASES = {
    "decode_gqa_d128_kv256": (8, 2, 1, 128, 256, 201, 0.0, 0.0),      # Llama-3 head geometry, one token at position 200 of a 256-padded window
    "decode_d64_kv96": (4, 4, 1, 64, 96, 38, 0.0, 0.0),
    "batch3_d64_kv40": (4, 2, 3, 64, 40, 35, 0.0, 0.0),
    "alibi_d128_kv64": (8, 8, 2, 128, 64, 60, 8.0, 0.0),
    "softcap_d96_kv64": (4, 1, 1, 96, 64, 64, 0.0, 10.0),
}


def main() -> None:
    ref = oracle.Reference("avx2")
    for tag, (H, Hk, N, DK, n_kv, first_masked, max_bias, softcap) in CASES.items():
        rng = np.random.default_rng(sum(map(ord, tag)))
        q = rng.standard_normal((1, H, N, DK)).astype(np.float32)
        k = rng.standard_normal((1, Hk, n_kv, DK)).astype(np.float16)
        v = rng.standard_normal((1, Hk, n_kv, DK)).astype(np.float16)
        k[0, :, first_masked + N:] = np.float16(np.nan); v[0, :, first_masked + N:] = np.float16(np.nan)      # never-written cache rows must not leak through the mask
        mask = np.zeros((64, n_kv), np.float16)                       # rows padded to GGML_KQ_MASK_PAD
        for t in range(N):
            mask[t, first_masked + t:] = -np.inf
        if max_bias > 0:
            mask[:N] += (rng.standard_normal((N, n_kv)) * 0.1).astype(np.float16)
        scale = np.float32(1.0 / np.sqrt(DK))
        y = ref.flash_attn_ext(q, k, v, mask, float(scale), max_bias, softcap)
        assert np.isfinite(y).all()
        np.savez_compressed(OUT / f"flash_attn_{tag}.npz", q=q, k=k, v=v, mask=mask, scale=scale, max_bias=np.float32(max_bias),
                            softcap=np.float32(softcap), y=y)
    # the same op on a Q8_0 K / V cache (-ctk q8_0 -ctv q8_0): the cache rows as the reference quantizer stores them, and the CPU backend's output
    for tag, (H, Hk, N, DK, n_kv, first_masked) in {"q8_0_decode_gqa_d128_kv256": (8, 2, 1, 128, 256, 201), "q8_0_batch3_d64_kv96": (4, 4, 3, 64, 96, 60)}.items():
        rng = np.random.default_rng(sum(map(ord, tag)))
        q = rng.standard_normal((1, H, N, DK)).astype(np.float32)
        k = rng.standard_normal((1, Hk, n_kv, DK)).astype(np.float32); v = rng.standard_normal((1, Hk, n_kv, DK)).astype(np.float32)
        mask = np.zeros((64, n_kv), np.float16)
        for t in range(N):
            mask[t, first_masked + t:] = -np.inf
        scale = np.float32(1.0 / np.sqrt(DK))
        y, kb, vb = ref.flash_attn_ext_q8_0(q, k, v, mask, float(scale))
        assert np.isfinite(y).all()
        np.savez_compressed(OUT / f"flash_attn_{tag}.npz", q=q, k_blocks=kb, v_blocks=vb, mask=mask, scale=scale, max_bias=np.float32(0), softcap=np.float32(0), y=y)
    print("wrote", len(CASES) + 2, "flash_attn fixtures to", OUT)


if __name__ == "__main__":
    main()
