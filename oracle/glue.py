"""oracle/glue.py -- numpy restatement of the reference's small graph ops that sit between the quantized matmuls of a decode
graph.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the product never imports this.

Arrays use numpy order = ggml dims reversed ([ne3, ne2, ne1, ne0]).  Each function follows the reference CPU op it cites
(ggml/src/ggml-cpu/ops.cpp unless noted), with the same f32 / f64 operand types.  Pinned by tests/test_oracle_glue.py against
the real reference (oracle/_ref via refshim ref_glue_op) and by tests/golden/glue_*.npz.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def bin_bcast(op: str, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """ADD / SUB / MUL / DIV with src1 repeated over src0 (binary-ops.cpp binary_op: dst[i] = a[i] op b[i % ne_b])."""
    a = np.asarray(a, f32); b = np.asarray(b, f32)
    a4 = a.reshape((1,) * (4 - a.ndim) + a.shape); b4 = b.reshape((1,) * (4 - b.ndim) + b.shape)
    reps = tuple(x // y for x, y in zip(a4.shape, b4.shape))
    bb = np.tile(b4, reps)
    r = {"add": a4 + bb, "sub": a4 - bb, "mul": a4 * bb, "div": a4 / bb}[op]
    return r.astype(f32).reshape(a.shape)


def silu(x: np.ndarray) -> np.ndarray:
    """vec.h ggml_silu_f32: x / (1 + expf(-x))"""
    x = np.asarray(x, f32)
    return (x / (f32(1.0) + np.exp(-x, dtype=f32))).astype(f32)


def rms_norm(x: np.ndarray, eps: float) -> np.ndarray:
    """ops.cpp:3180-3226: sum += (double)(x*x) with the square rounded to f32; mean = (float)(sum/ne00);
    scale = 1/sqrtf(mean + eps); y = x*scale."""
    x = np.asarray(x, f32)
    sq = (x * x).astype(f32).astype(np.float64)
    mean = (sq.sum(axis=-1) / x.shape[-1]).astype(f32)
    scale = (f32(1.0) / np.sqrt((mean + f32(eps)).astype(f32), dtype=f32)).astype(f32)
    return (x * scale[..., None]).astype(f32)


def soft_max(x: np.ndarray, mask: np.ndarray | None, scale: float, max_bias: float = 0.0) -> np.ndarray:
    """ops.cpp:4641-4736: wp = x*scale + slope(head)*mask[row % ne01]; max; expf(wp - max); sum in double; * (float)(1/sum).
    x: [ne3, ne2 (heads), ne1, ne0]; mask: [>= ne1, ne0] (f32; f16 masks are converted by the caller)."""
    x = np.asarray(x, f32)
    x4 = x.reshape((1,) * (4 - x.ndim) + x.shape)
    ne3, ne2, ne1, ne0 = x4.shape
    w = (x4 * f32(scale)).astype(f32)
    if mask is not None:
        m = np.asarray(mask, f32).reshape(-1, ne0)[:ne1]
        n_head_log2 = 1 << int(np.floor(np.log2(ne2)))
        m0 = np.float32(2.0) ** f32(-(max_bias) / n_head_log2); m1 = np.float32(2.0) ** f32(-(max_bias / 2.0) / n_head_log2)
        for h in range(ne2):
            slope = f32(1.0)
            if max_bias > 0.0:
                slope = f32(m0 ** f32(h + 1)) if h < n_head_log2 else f32(m1 ** f32(2 * (h - n_head_log2) + 1))
            w[:, h] = (w[:, h] + (slope * m).astype(f32)[None]).astype(f32)
    mx = w.max(axis=-1, keepdims=True)
    e = np.exp((w - mx).astype(f32), dtype=f32)
    s = e.astype(np.float64).sum(axis=-1, keepdims=True)
    return (e * (1.0 / s).astype(f32)).astype(f32).reshape(x.shape)


def rope(x: np.ndarray, pos: np.ndarray, n_dims: int, mode: int = 0, freq_factors: np.ndarray | None = None, n_ctx_orig: int = 0,
         freq_base: float = 10000.0, freq_scale: float = 1.0, ext_factor: float = 0.0, attn_factor: float = 1.0,
         beta_fast: float = 32.0, beta_slow: float = 1.0) -> np.ndarray:
    """ops.cpp:4990-5270 (rope_yarn, ggml_rope_cache_init, ggml_compute_forward_rope_f32), normal (mode 0) and neox (mode 2).
    x: [ne3, ne2 (positions), ne1 (heads), ne0].  theta is advanced by repeated f32 multiplication as the CPU does."""
    x = np.asarray(x, f32)
    x4 = x.reshape((1,) * (4 - x.ndim) + x.shape)
    out = x4.copy()
    ne0 = x4.shape[-1]
    theta_scale = f32(np.float32(freq_base) ** f32(-2.0 / n_dims))
    def corr_dim(n_rot):
        return f32(n_dims * np.log(f32(n_ctx_orig / (n_rot * 2 * np.pi))) / (2 * np.log(f32(freq_base)))) if n_ctx_orig > 0 else f32(0)
    corr0 = max(0.0, float(np.floor(corr_dim(beta_fast)))); corr1 = min(n_dims - 1.0, float(np.ceil(corr_dim(beta_slow))))
    for i2, p in enumerate(np.asarray(pos, np.int64)):
        theta = f32(p)
        cos = np.zeros(ne0 // 2, f32); sin = np.zeros(ne0 // 2, f32)
        for ic in range(ne0 // 2):
            ff = f32(freq_factors[ic]) if freq_factors is not None and ic < n_dims // 2 else f32(1.0)
            te = f32(theta / ff)
            ti = f32(f32(freq_scale) * te)
            th, ms = ti, f32(attn_factor)
            if ext_factor != 0.0:
                y = (ic - corr0) / max(0.001, corr1 - corr0)
                ramp = f32((1 - min(1.0, max(0.0, y))) * ext_factor)
                th = f32(ti * (1 - ramp) + te * ramp)
                ms = f32(ms * f32(1.0 + 0.1 * np.log(f32(1.0 / freq_scale))))
            cos[ic] = f32(np.cos(th, dtype=f32) * ms); sin[ic] = f32(np.sin(th, dtype=f32) * ms)
            theta = f32(theta * theta_scale)
        h = n_dims // 2
        src = x4[:, i2]
        if mode == 2:
            x0, x1 = src[..., :h], src[..., h:n_dims]
            out[:, i2, :, :h] = x0 * cos[:h] - x1 * sin[:h]
            out[:, i2, :, h:n_dims] = x0 * sin[:h] + x1 * cos[:h]
        else:
            x0, x1 = src[..., 0:n_dims:2], src[..., 1:n_dims:2]
            out[:, i2, :, 0:n_dims:2] = x0 * cos[:h] - x1 * sin[:h]
            out[:, i2, :, 1:n_dims:2] = x0 * sin[:h] + x1 * cos[:h]
    return out.astype(f32).reshape(x.shape)


def mul_mat_f(a: np.ndarray, b: np.ndarray, a_is_f16: bool) -> np.ndarray:
    """MUL_MAT with an f16 / f32 src0 (ggml-cpu.c:1266-1458): dst[.., n, m] = sum_k a[.., m, k] * b[.., n, k], dims 2/3 of a
    broadcast; an f16 src0 makes the CPU round src1 to f16 first (vec_dot_type F16), products accumulated in f32."""
    a = np.asarray(a, f32); b = np.asarray(b, f32)
    if a_is_f16:
        a = a.astype(np.float16).astype(f32); b = b.astype(np.float16).astype(f32)
    a4 = a.reshape((1,) * (4 - a.ndim) + a.shape); b4 = b.reshape((1,) * (4 - b.ndim) + b.shape)
    r3, r2 = b4.shape[0] // a4.shape[0], b4.shape[1] // a4.shape[1]
    out = np.empty(b4.shape[:2] + (b4.shape[2], a4.shape[2]), f32)
    for i3 in range(b4.shape[0]):
        for i2 in range(b4.shape[1]):
            out[i3, i2] = (b4[i3, i2].astype(np.float64) @ a4[i3 // r3, i2 // r2].astype(np.float64).T).astype(f32)
    return out.reshape(b.shape[:-2] + (b.shape[-2], a.shape[-2]))
