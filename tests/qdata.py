"""Seeded synthetic quantized data for the tests (works with or without oracle/_ref).

`random_blocks` makes VALID packed rows straight from random bytes: any byte pattern is a legal
block as long as its f16 scale fields are finite, so those fields are overwritten with small finite
halfs.  `quantized_weights` prefers the real reference quantizer (ggml_quantize_chunk through
oracle/_ref) when it is present, as tests/test-backend-ops.cpp:39-128 does."""
from __future__ import annotations

import numpy as np

import oracle

# (block bytes, [byte offsets of f16 scale fields inside a block])
_F16_FIELDS = {
    oracle.Q4_0: (18, [0]), oracle.Q4_1: (20, [0, 2]), oracle.Q5_0: (22, [0]), oracle.Q5_1: (24, [0, 2]),
    oracle.Q8_0: (34, [0]), oracle.Q2_K: (84, [80, 82]), oracle.Q3_K: (110, [108]), oracle.Q4_K: (144, [0, 2]),
    oracle.Q5_K: (176, [0, 2]), oracle.Q6_K: (210, [208]), oracle.IQ4_NL: (18, [0]), oracle.IQ4_XS: (136, [0]),
    oracle.IQ2_XXS: (66, [0]), oracle.IQ2_XS: (74, [0]), oracle.IQ2_S: (82, [0]), oracle.IQ3_XXS: (98, [0]),
    oracle.IQ3_S: (110, [0]), oracle.IQ1_S: (50, [0]),
    oracle.IQ1_M: (56, []),      # its f16 super-scale is scattered over the top nibbles of scales[4] (u16 at 48..55)
}
BLCK = {t: (32 if t in (oracle.Q4_0, oracle.Q4_1, oracle.Q5_0, oracle.Q5_1, oracle.Q8_0, oracle.IQ4_NL) else 256)
        for t in _F16_FIELDS}


def row_size(t: int, k: int) -> int:
    return k // BLCK[t] * _F16_FIELDS[t][0]


def random_blocks(t: int, nrows: int, k: int, rng: np.random.Generator) -> np.ndarray:
    bs, fields = _F16_FIELDS[t]
    nb = k // BLCK[t]
    w = rng.integers(0, 256, (nrows, nb, bs), dtype=np.uint8)
    for off in fields:
        scale = rng.uniform(1e-3, 5e-2, (nrows, nb)).astype(np.float16)
        if rng.random() < 0.5:
            scale = -scale if off == 0 and t in (oracle.Q6_K, oracle.Q3_K) else scale   # signed super-scales occur in practice
        w[:, :, off:off + 2] = scale.view(np.uint8).reshape(nrows, nb, 2)
    if t == oracle.IQ1_M:
        h = rng.uniform(1e-3, 5e-2, (nrows, nb)).astype(np.float16).view(np.uint16)
        for i in range(4):       # nibble i of the half lives in bits 12..15 of scales[i]
            w[:, :, 49 + 2 * i] = (w[:, :, 49 + 2 * i] & 0x0f) | ((((h >> (4 * i)) & 0xf) << 4).astype(np.uint8))
    return w.reshape(nrows, nb * bs)


def quantized_weights(t: int, nrows: int, k: int, rng: np.random.Generator, scale: float = 1.0) -> np.ndarray:
    if oracle.ref_available("scalar") and nrows * k <= (1 << 24):
        ref = _ref()
        return ref.quantize(t, (rng.uniform(-1, 1, (nrows, k)) * scale).astype(np.float32))
    return random_blocks(t, nrows, k, rng)


_REF = None


def _ref():
    global _REF
    if _REF is None:
        _REF = oracle.Reference("scalar")
    return _REF
