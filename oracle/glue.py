"""oracle/glue.py -- numpy restatement of the reference's small graph ops that sit between the quantized matmuls of a decode
graph.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the product never imports this.

Arrays use numpy order = ggml dims reversed ([ne3, ne2, ne1, ne0]).  Each function follows the reference CPU op it cites
(ggml/src/ggml-cpu/ops.cpp unless noted), with the same f32 / f64 operand types.  Pinned by tests/test_oracle_glue.py against
the real reference (oracle/_ref via refshim ref_glue_op) and by tests/golden/glue_*.npz.
"""
from __future__ import annotations

import ctypes
import ctypes.util

import numpy as np

f32 = np.float32

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.expf.restype = ctypes.c_float; _libm.expf.argtypes = [ctypes.c_float]
_libm.tanhf.restype = ctypes.c_float; _libm.tanhf.argtypes = [ctypes.c_float]


def _expf(x) -> np.float32:
    """the C library's expf, which is what the reference's scalar code calls (numpy's exp is a different implementation, 1 ulp apart now and then)"""
    return f32(_libm.expf(float(x)))


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


def vec_dot_f16_simd(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """ggml_vec_dot_f16 in its AVX2 / F16C / FMA form (vec.cpp:128-168; simd-mappings.h: GGML_F16_STEP 32, GGML_F16_EPR 8, 4 accumulators,
    GGML_F32x8_REDUCE): rows x [R, n] (f16 values) against y [n] (f16 values) -> [R] f32.  Element e feeds accumulator (j, l) = ((e % 32) / 8, e % 8)
    by an f32 fma, blocks of 32 in order; then per l (a0 + a2) + (a1 + a3) = s[l]; t[i] = s[i] + s[i + 4]; result (t0 + t1) + (t2 + t3); the tail
    beyond the last whole block is summed in f64 from f32-rounded products.  (The fma is taken as round(f64 sum of the exact product and the
    accumulator): the product of two f16 values is exact in f64; the one remaining double rounding needs a 2^-29 coincidence.)"""
    x = np.asarray(x, np.float16).astype(np.float64); y = np.asarray(y, np.float16).astype(np.float64)
    R, n = x.shape
    npb = n & ~31
    acc = np.zeros((R, 4, 8), f32)
    for b0 in range(0, npb, 32):
        prod = (x[:, b0:b0 + 32] * y[b0:b0 + 32]).reshape(R, 4, 8)
        acc = (prod + acc.astype(np.float64)).astype(f32)
    s_ = ((acc[:, 0] + acc[:, 2]).astype(f32) + (acc[:, 1] + acc[:, 3]).astype(f32)).astype(f32)      # [R, 8]
    t = (s_[:, :4] + s_[:, 4:]).astype(f32)
    res = ((t[:, 0] + t[:, 1]).astype(f32) + (t[:, 2] + t[:, 3]).astype(f32)).astype(f32)
    if npb < n:
        tail = (x[:, npb:] * y[npb:]).astype(f32).astype(np.float64)
        sumf = res.astype(np.float64)
        for i in range(n - npb):
            sumf = sumf + tail[:, i]
        res = sumf.astype(f32)
    return res


def flash_attn_ext(q: np.ndarray, k: np.ndarray, v: np.ndarray, mask: np.ndarray | None, scale: float, max_bias: float = 0.0,
                   logit_softcap: float = 0.0) -> np.ndarray:
    """FLASH_ATTN_EXT with an F16 K / V cache (ops.cpp:6686-6905 ggml_compute_forward_flash_attn_ext_f16).
    q: [B, H, N, DK] f32; k: [B, Hk, n_kv, DK] f16; v: [B, Hv, n_kv, DV] f16; mask: [>= N, n_kv] f16 or None.  Returns [B, N, H, DV] f32.
    Per query row the CPU walks the positions IN ORDER with a running maximum M and sum S (online softmax) and keeps V.P in an F16 accumulator:
      q -> f16;  s = dot(k_j, q) * scale [softcap] + slope * mask_j;  positions with mask == -inf are skipped;
      s > M: ms = expf(Mold - s), VKQ16 = f16(f32(VKQ16) * ms) (ggml_vec_scale_f16), vs = 1;  else vs = expf(s - M);
      VKQ16 = f16(fma(f32(v_j), vs, f32(VKQ16))) (ggml_vec_mad_f16, the F16C / FMA form);  S = S * ms + vs;
    result = f32(VKQ16) * (1 / S).  The dot product is vec_dot_f16_simd above (the AVX2 build's order), expf / tanhf are the C library's.
    Bit-exact against the reference's AVX2 build, within one f16 flip per few hundred elements of its scalar build (tests/test_oracle_glue.py)."""
    q = np.asarray(q, f32); k = np.asarray(k, np.float16); v = np.asarray(v, np.float16)
    B, H, N, DK = q.shape
    Hk, n_kv, Hv, DV = k.shape[1], k.shape[2], v.shape[1], v.shape[3]
    n_head_log2 = 1 << int(np.floor(np.log2(H)))
    m0 = f32(2.0) ** f32(-(max_bias) / n_head_log2); m1 = f32(2.0) ** f32(-(max_bias / 2.0) / n_head_log2)
    sc = f32(scale)
    if logit_softcap != 0.0:
        sc = f32(sc / f32(logit_softcap))
    out = np.zeros((B, N, H, DV), f32)
    q16 = q.astype(np.float16).astype(np.float64)
    for b in range(B):
        for h in range(H):
            slope = f32(1.0)
            if max_bias > 0.0:
                slope = f32(m0 ** f32(h + 1)) if h < n_head_log2 else f32(m1 ** f32(2 * (h - n_head_log2) + 1))
            kh = k[b, h // (H // Hk)].astype(np.float64); vh = v[b, h // (H // Hv)].astype(np.float64)
            dots = np.stack([vec_dot_f16_simd(kh, q16[b, h, n]) for n in range(N)])     # [N, n_kv]
            for n in range(N):
                S = f32(0.0); M = f32(-np.inf)
                acc = np.zeros(DV, np.float16)
                for j in range(n_kv):
                    mv = f32(slope * f32(mask[n, j])) if mask is not None else f32(0.0)
                    if mv == -np.inf:
                        continue
                    s = f32(dots[n, j] * sc)
                    if logit_softcap != 0.0:
                        s = f32(f32(logit_softcap) * f32(_libm.tanhf(float(s))))
                    s = f32(s + mv)
                    ms = f32(1.0); vs = f32(1.0)
                    if s > M:
                        Mold = M; M = s
                        ms = _expf(f32(Mold - M))
                        acc = (acc.astype(f32) * ms).astype(np.float16)
                    else:
                        vs = _expf(f32(s - M))
                    acc = (vh[j] * np.float64(vs) + acc.astype(np.float64)).astype(f32).astype(np.float16)     # fma in f32, then f16
                    S = f32(f32(S * ms) + vs)
                out[b, n, h] = acc.astype(f32) * f32(f32(1.0) / S)
    return out


def quantize_row_q8_0_simd(x: np.ndarray):
    """quantize_row_q8_0 in its AVX2 form (ggml-cpu-quants.c:738-…, rounding :842-845): per 32 elements d = amax / 127, id = 1 / d (0 if d == 0),
    q = round-half-to-even(x * id), the stored scale is f16(d).  x: [..., n] f32 -> (d f16 [..., n/32], q int8 [..., n/32, 32])."""
    x = np.asarray(x, f32)
    xb = x.reshape(x.shape[:-1] + (x.shape[-1] // 32, 32))
    amax = np.abs(xb).max(axis=-1)
    d = (amax / f32(127.0)).astype(f32)
    with np.errstate(divide="ignore"):
        idv = np.where(d != 0, (f32(1.0) / d).astype(f32), f32(0.0)).astype(f32)
    qv = np.rint((xb * idv[..., None]).astype(f32)).astype(np.int8)
    return d.astype(np.float16), qv


def vec_dot_q8_0_q8_0_simd(kd: np.ndarray, kq: np.ndarray, qd: np.ndarray, qq: np.ndarray) -> np.ndarray:
    """ggml_vec_dot_q8_0_q8_0 in its AVX2 form (ggml-cpu-quants.c:3597-…): rows kd [R, nb] f16 / kq [R, nb, 32] int8 against one quantized vector
    qd [nb] f16 / qq [nb, 32] int8.  Per block d = f32(kd) * f32(qd); eight f32 lanes, lane l += fma(d, float(sum of products 4l..4l+3));
    result hsum_float_8: t[i] = a[i] + a[i+4]; (t0 + t2) + (t1 + t3)."""
    R, nb = kd.shape
    acc = np.zeros((R, 8), f32)
    for b in range(nb):
        dd = (kd[:, b].astype(f32) * f32(qd[b])).astype(f32)
        s4 = (kq[:, b].astype(np.int32) * qq[b].astype(np.int32)).reshape(R, 8, 4).sum(axis=-1).astype(f32)
        acc = (dd[:, None].astype(np.float64) * s4.astype(np.float64) + acc.astype(np.float64)).astype(f32)
    t = (acc[:, :4] + acc[:, 4:]).astype(f32)
    return ((t[:, 0] + t[:, 2]).astype(f32) + (t[:, 1] + t[:, 3]).astype(f32)).astype(f32)


def flash_attn_ext_q8_0(q: np.ndarray, k_blocks: np.ndarray, v_blocks: np.ndarray, mask: np.ndarray | None, scale: float, max_bias: float = 0.0,
                        logit_softcap: float = 0.0, kv: str = "q8_0") -> np.ndarray:
    """FLASH_ATTN_EXT on a Q8_0 K / V cache (ops.cpp:6686-6905 with k->type = v->type = Q8_0).  q: [B, H, N, DK] f32; k_blocks / v_blocks: the cache rows
    as stored, uint8 [B, Hk, n_kv, DK/32*34] (block_q8_0: f16 d, 32 int8).  The CPU quantizes q to Q8_0 (the K type's vec_dot_type; SIMD quantizer),
    takes the scores with ggml_vec_dot_q8_0_q8_0, walks the positions in order with a running maximum, and keeps V.P in an F32 accumulator: V is
    dequantized (q * d), the accumulator rescaled by an f32 multiply when the maximum grows and updated with an f32 fma per position.
    kv = "q4_0": the same on a Q4_0 cache (block_q4_0: f16 d, 16 bytes of nibbles, low nibbles = elements 0..15, value = nibble - 8; the K dot is
    ggml_vec_dot_q4_0_q8_0, whose AVX2 form has the lane structure of the Q8_0 one after bytes_from_nibbles_32)."""
    q = np.asarray(q, f32)
    B, H, N, DK = q.shape
    Hk, n_kv = k_blocks.shape[1], k_blocks.shape[2]
    bb = 34 if kv == "q8_0" else 18
    def unpack(blk):
        nbk = blk.shape[-1] // bb
        b = np.ascontiguousarray(blk).reshape(blk.shape[:-1] + (nbk, bb))
        d = np.ascontiguousarray(b[..., :2]).view(np.float16)[..., 0]
        if kv == "q8_0":
            return d, np.ascontiguousarray(b[..., 2:]).view(np.int8)
        qs = b[..., 2:].astype(np.int16)
        return d, np.concatenate([(qs & 15) - 8, (qs >> 4) - 8], axis=-1).astype(np.int8)       # elements 0..15 | 16..31
    kd, kq = unpack(k_blocks); vd, vq = unpack(v_blocks)
    DV = vq.shape[-2] * 32
    n_head_log2 = 1 << int(np.floor(np.log2(H)))
    m0 = f32(2.0) ** f32(-(max_bias) / n_head_log2); m1 = f32(2.0) ** f32(-(max_bias / 2.0) / n_head_log2)
    sc = f32(scale)
    if logit_softcap != 0.0:
        sc = f32(sc / f32(logit_softcap))
    out = np.zeros((B, N, H, DV), f32)
    for b in range(B):
        for h in range(H):
            slope = f32(1.0)
            if max_bias > 0.0:
                slope = f32(m0 ** f32(h + 1)) if h < n_head_log2 else f32(m1 ** f32(2 * (h - n_head_log2) + 1))
            hk = h // (H // Hk)
            vdeq = (vq[b, hk].astype(f32) * vd[b, hk].astype(f32)[..., None]).astype(f32).reshape(n_kv, DV)      # dequantize_row_q8_0
            for n in range(N):
                qdn, qqn = quantize_row_q8_0_simd(q[b, h, n])
                dots = vec_dot_q8_0_q8_0_simd(kd[b, hk], kq[b, hk], qdn, qqn)
                S = f32(0.0); M = f32(-np.inf)
                acc = np.zeros(DV, f32)
                for j in range(n_kv):
                    mv = f32(slope * f32(mask[n, j])) if mask is not None else f32(0.0)
                    if mv == -np.inf:
                        continue
                    s = f32(dots[j] * sc)
                    if logit_softcap != 0.0:
                        s = f32(f32(logit_softcap) * f32(_libm.tanhf(float(s))))
                    s = f32(s + mv)
                    ms = f32(1.0); vs = f32(1.0)
                    if s > M:
                        Mold = M; M = s
                        ms = _expf(f32(Mold - M))
                        acc = (acc * ms).astype(f32)                                                        # ggml_vec_scale_f32
                    else:
                        vs = _expf(f32(s - M))
                    acc = (vdeq[j].astype(np.float64) * np.float64(vs) + acc.astype(np.float64)).astype(f32)    # ggml_vec_mad_f32 (fma)
                    S = f32(f32(S * ms) + vs)
                out[b, n, h] = acc * f32(f32(1.0) / S)
    return out
