"""oracle -- TEST INFRASTRUCTURE ONLY.

ctypes loaders for
  * ``liboracle.so``  : our plain-C restatement of the reference CPU arithmetic (oracle/*.c)
  * ``_ref/<variant>/librefshim.so`` : the REAL reference (ggml CPU backend compiled from
    /root/reference by oracle/Makefile), when it has been built.

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of bench.py may import
this package.  The product package (llama.cpp.dsp_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent

# ggml type ids (ggml/include/ggml.h: enum ggml_type)
F32, F16 = 0, 1
Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q8_1 = 2, 3, 6, 7, 8, 9
Q2_K, Q3_K, Q4_K, Q5_K, Q6_K, Q8_K = 10, 11, 12, 13, 14, 15
IQ2_XXS, IQ2_XS, IQ3_XXS, IQ1_S, IQ4_NL, IQ3_S, IQ2_S, IQ4_XS, IQ1_M = 16, 17, 18, 19, 20, 21, 22, 23, 29

TYPE_NAMES = {
    Q4_0: "q4_0", Q4_1: "q4_1", Q5_0: "q5_0", Q5_1: "q5_1", Q8_0: "q8_0", Q8_1: "q8_1",
    Q2_K: "q2_K", Q3_K: "q3_K", Q4_K: "q4_K", Q5_K: "q5_K", Q6_K: "q6_K", Q8_K: "q8_K",
    IQ4_NL: "iq4_nl", IQ4_XS: "iq4_xs",
    IQ2_XXS: "iq2_xxs", IQ2_XS: "iq2_xs", IQ2_S: "iq2_s", IQ3_XXS: "iq3_xxs", IQ3_S: "iq3_s", IQ1_S: "iq1_s", IQ1_M: "iq1_m",
}
ROUND_AWAY, ROUND_EVEN = 0, 1

_i64, _i32, _vp, _fp = C.c_int64, C.c_int, C.c_void_p, C.POINTER(C.c_float)


def build(force: bool = False) -> None:
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    if force or not (HERE / "liboracle.so").exists():
        subprocess.check_call(["make", "-s", "-C", str(HERE), "oracle"])
    if Path("/root/reference/ggml/src/ggml.c").exists():
        subprocess.check_call(["make", "-s", "-j8", "-C", str(HERE), "ref"])


def _ptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_vp)


class Oracle:
    """The C restatement (liboracle.so)."""

    def __init__(self) -> None:
        so = HERE / "liboracle.so"
        if not so.exists():
            build()
        L = self.lib = C.CDLL(str(so))
        L.orc_supported.argtypes = [_i32]
        L.orc_blck_size.restype = _i64; L.orc_blck_size.argtypes = [_i32]
        L.orc_type_size.restype = _i64; L.orc_type_size.argtypes = [_i32]
        L.orc_row_size.restype = _i64; L.orc_row_size.argtypes = [_i32, _i64]
        L.orc_vec_dot_type.argtypes = [_i32]
        L.orc_f16_to_f32.restype = C.c_float; L.orc_f16_to_f32.argtypes = [C.c_uint16]
        L.orc_f32_to_f16.restype = C.c_uint16; L.orc_f32_to_f16.argtypes = [C.c_float]
        L.orc_dequantize_row.argtypes = [_i32, _vp, _vp, _i64]
        L.orc_quantize_row_act.argtypes = [_i32, _vp, _vp, _i64, _i32]
        L.orc_vec_dot.argtypes = [_i32, _i64, _vp, _vp, _vp]
        L.orc_mul_mat.argtypes = [_i32, _vp, _vp, _vp] + [_i64] * 7 + [_i32]
        L.orc_mul_mat_id.argtypes = [_i32, _vp, _vp, _vp, _vp] + [_i64] * 6 + [_i32]

    def supported(self, t: int) -> bool: return bool(self.lib.orc_supported(t))
    def blck_size(self, t: int) -> int: return self.lib.orc_blck_size(t)
    def type_size(self, t: int) -> int: return self.lib.orc_type_size(t)
    def row_size(self, t: int, k: int) -> int: return self.lib.orc_row_size(t, k)
    def vec_dot_type(self, t: int) -> int: return self.lib.orc_vec_dot_type(t)

    def dequantize(self, t: int, blocks: np.ndarray, k: int) -> np.ndarray:
        blocks = np.ascontiguousarray(blocks).view(np.uint8).reshape(-1)
        nrows = blocks.size // self.row_size(t, k)
        out = np.empty((nrows, k), np.float32)
        rc = self.lib.orc_dequantize_row(t, _ptr(blocks), _ptr(out), nrows * k)
        assert rc == 0, rc
        return out

    def quantize_act(self, act_t: int, x: np.ndarray, round_mode: int = ROUND_AWAY) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        k = x.shape[-1]
        rows = x.reshape(-1, k)
        out = np.zeros((rows.shape[0], self.row_size(act_t, k)), np.uint8)
        for r in range(rows.shape[0]):
            rc = self.lib.orc_quantize_row_act(act_t, _ptr(rows[r]), _ptr(out[r]), k, round_mode)
            assert rc == 0
        return out

    def vec_dot(self, t: int, k: int, w_row: np.ndarray, act_row: np.ndarray) -> float:
        out = np.zeros(1, np.float32)
        rc = self.lib.orc_vec_dot(t, k, _ptr(out), _ptr(np.ascontiguousarray(w_row)), _ptr(np.ascontiguousarray(act_row)))
        assert rc == 0, rc
        return float(out[0])

    def mul_mat(self, t: int, w: np.ndarray, x: np.ndarray, M: int, N: int, K: int,
                ne02: int = 1, ne03: int = 1, ne12: int = 1, ne13: int = 1,
                round_mode: int = ROUND_AWAY) -> np.ndarray:
        w = np.ascontiguousarray(w).view(np.uint8).reshape(-1)
        x = np.ascontiguousarray(x, np.float32)
        assert w.size == self.row_size(t, K) * M * ne02 * ne03, (w.size, self.row_size(t, K) * M * ne02 * ne03)
        assert x.size == K * N * ne12 * ne13
        dst = np.empty((ne13, ne12, N, M), np.float32)
        rc = self.lib.orc_mul_mat(t, _ptr(w), _ptr(x), _ptr(dst), M, N, K, ne02, ne03, ne12, ne13, round_mode)
        assert rc == 0, rc
        return dst if (ne12 * ne13 > 1) else dst.reshape(N, M)

    def mul_mat_id(self, t: int, as_: np.ndarray, b: np.ndarray, ids: np.ndarray, M: int, K: int,
                   n_expert: int, round_mode: int = ROUND_AWAY) -> np.ndarray:
        as_ = np.ascontiguousarray(as_).view(np.uint8).reshape(-1)
        b = np.ascontiguousarray(b, np.float32)          # [n_tok, b_ne1, K]
        ids = np.ascontiguousarray(ids, np.int32)        # [n_tok, n_used]
        n_tok, n_used = ids.shape
        b_ne1 = b.shape[1]
        dst = np.empty((n_tok, n_used, M), np.float32)
        rc = self.lib.orc_mul_mat_id(t, _ptr(as_), _ptr(b), _ptr(ids), _ptr(dst), M, K, n_expert, n_used, n_tok, b_ne1, round_mode)
        assert rc == 0, rc
        return dst


def ref_available(variant: str = "scalar") -> bool:
    return (HERE / "_ref" / variant / "librefshim.so").exists()


class Reference:
    """The real reference CPU backend (oracle/_ref/<variant>/librefshim.so)."""

    def __init__(self, variant: str = "scalar") -> None:
        so = HERE / "_ref" / variant / "librefshim.so"
        if not so.exists():
            raise FileNotFoundError(f"{so} not built (run `make -C oracle ref` where /root/reference exists)")
        self.variant = variant
        L = self.lib = C.CDLL(str(so))
        for f in ("ref_blck_size", "ref_type_size"):
            getattr(L, f).restype = _i64; getattr(L, f).argtypes = [_i32]
        L.ref_row_size.restype = _i64; L.ref_row_size.argtypes = [_i32, _i64]
        L.ref_type_name.restype = C.c_char_p; L.ref_type_name.argtypes = [_i32]
        L.ref_vec_dot_type.argtypes = [_i32]
        L.ref_requires_imatrix.argtypes = [_i32]
        L.ref_quantize_chunk.restype = _i64; L.ref_quantize_chunk.argtypes = [_i32, _vp, _vp, _i64, _i64]
        L.ref_dequantize_row.restype = None; L.ref_dequantize_row.argtypes = [_i32, _vp, _vp, _i64]
        L.ref_from_float_ref.restype = None; L.ref_from_float_ref.argtypes = [_i32, _vp, _vp, _i64]
        L.ref_from_float_cpu.restype = None; L.ref_from_float_cpu.argtypes = [_i32, _vp, _vp, _i64]
        L.ref_vec_dot.restype = None; L.ref_vec_dot.argtypes = [_i32, _i64, _vp, _vp, _vp]
        L.ref_mul_mat.argtypes = [_i32, _vp, _vp, _vp, _i64, _i64, _i64, _i32]
        L.ref_mul_mat_id.argtypes = [_i32, _vp, _vp, _vp, _vp] + [_i64] * 6 + [_i32]
        L.ref_bench_chain.restype = C.c_double
        L.ref_bench_chain.argtypes = [_i32, _vp, _vp, _vp, _i64, _i32, _i32, _i32]
        if hasattr(L, "ref_glue_op"):
            L.ref_glue_op.argtypes = [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32]
        if hasattr(L, "ref_flash_attn_ext"):
            L.ref_flash_attn_ext.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_float, C.c_float, C.c_float, _vp, _i32]
        if hasattr(L, "ref_flash_attn_ext_t"):
            L.ref_flash_attn_ext_t.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_float, C.c_float, C.c_float, _i32, _vp, _vp, _vp, _i32]

    def blck_size(self, t): return self.lib.ref_blck_size(t)
    def type_size(self, t): return self.lib.ref_type_size(t)
    def row_size(self, t, k): return self.lib.ref_row_size(t, k)
    def type_name(self, t): return self.lib.ref_type_name(t).decode()
    def vec_dot_type(self, t): return self.lib.ref_vec_dot_type(t)

    def quantize(self, t: int, x: np.ndarray) -> np.ndarray:
        """f32 [nrows, k] -> packed rows uint8 [nrows, row_size] via ggml_quantize_chunk."""
        x = np.ascontiguousarray(x, np.float32)
        nrows, k = x.reshape(-1, x.shape[-1]).shape
        out = np.zeros((nrows, self.row_size(t, k)), np.uint8)
        n = self.lib.ref_quantize_chunk(t, _ptr(x), _ptr(out), nrows, k)
        assert n == out.size, (n, out.size)
        return out

    def dequantize(self, t: int, blocks: np.ndarray, k: int) -> np.ndarray:
        blocks = np.ascontiguousarray(blocks).view(np.uint8).reshape(-1)
        nrows = blocks.size // self.row_size(t, k)
        out = np.empty((nrows, k), np.float32)
        self.lib.ref_dequantize_row(t, _ptr(blocks), _ptr(out), nrows * k)
        return out

    def quantize_act(self, act_t: int, x: np.ndarray, cpu_path: bool = False) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        k = x.shape[-1]
        rows = x.reshape(-1, k)
        out = np.zeros((rows.shape[0], self.row_size(act_t, k)), np.uint8)
        fn = self.lib.ref_from_float_cpu if cpu_path else self.lib.ref_from_float_ref
        for r in range(rows.shape[0]):
            fn(act_t, _ptr(rows[r]), _ptr(out[r]), k)
        return out

    def vec_dot(self, t: int, k: int, w_row: np.ndarray, act_row: np.ndarray) -> float:
        out = np.zeros(1, np.float32)
        self.lib.ref_vec_dot(t, k, _ptr(out), _ptr(np.ascontiguousarray(w_row)), _ptr(np.ascontiguousarray(act_row)))
        return float(out[0])

    def mul_mat(self, t: int, w: np.ndarray, x: np.ndarray, M: int, N: int, K: int, n_threads: int = 1) -> np.ndarray:
        w = np.ascontiguousarray(w).view(np.uint8).reshape(-1)
        x = np.ascontiguousarray(x, np.float32)
        dst = np.empty((N, M), np.float32)
        rc = self.lib.ref_mul_mat(t, _ptr(w), _ptr(x), _ptr(dst), M, N, K, n_threads)
        assert rc == 0, rc
        return dst

    def mul_mat_id(self, t: int, as_: np.ndarray, b: np.ndarray, ids: np.ndarray, M: int, K: int,
                   n_expert: int, n_threads: int = 1) -> np.ndarray:
        as_ = np.ascontiguousarray(as_).view(np.uint8).reshape(-1)
        b = np.ascontiguousarray(b, np.float32)
        ids = np.ascontiguousarray(ids, np.int32)
        n_tok, n_used = ids.shape
        dst = np.empty((n_tok, n_used, M), np.float32)
        rc = self.lib.ref_mul_mat_id(t, _ptr(as_), _ptr(b), _ptr(ids), _ptr(dst), M, K, n_expert, n_used, n_tok, b.shape[1], n_threads)
        assert rc == 0, rc
        return dst

    def glue_op(self, op: int, a, b=None, pos=None, fparams=(), iparams=(), out_shape=None):
        """One residency op on the reference CPU backend (refshim ref_glue_op); arrays are numpy with ggml dims REVERSED
        (numpy shape [ne3, ne2, ne1, ne0], leading dims optional)."""
        def ne(x):
            sh = list(x.shape)[::-1] + [1] * (4 - x.ndim)
            return np.asarray(sh, np.int64)
        a = np.ascontiguousarray(a, np.float32)
        nea = ne(a)
        bb = np.ascontiguousarray(b, np.float32) if b is not None else None
        neb = ne(bb) if bb is not None else np.zeros(4, np.int64)
        pp = np.ascontiguousarray(pos, np.int32) if pos is not None else None
        fp = np.asarray(list(fparams) + [0.0] * 8, np.float32); ip = np.asarray(list(iparams) + [0] * 8, np.int32)
        out = np.empty(out_shape if out_shape is not None else a.shape, np.float32)
        rc = self.lib.ref_glue_op(op, _ptr(a), _ptr(nea), _ptr(bb) if bb is not None else None, _ptr(neb), _ptr(pp) if pp is not None else None,
                                  _ptr(fp), _ptr(ip), _ptr(out), 1)
        if rc != 0:
            raise RuntimeError(f"ref_glue_op({op}) failed: {rc}")
        return out

    def flash_attn_ext(self, q, k, v, mask, scale: float, max_bias: float = 0.0, softcap: float = 0.0, n_threads: int = 1):
        """FLASH_ATTN_EXT on the reference CPU backend (refshim ref_flash_attn_ext).  numpy shapes as oracle.glue.flash_attn_ext:
        q [B, H, N, DK] f32; k [B, Hk, n_kv, DK], v [B, Hv, n_kv, DV] (f16 values); mask [n_pad, n_kv] (f16 values) or None -> [B, N, H, DV] f32."""
        q = np.ascontiguousarray(q, np.float32); k = np.ascontiguousarray(k, np.float32); v = np.ascontiguousarray(v, np.float32)
        ne = lambda x: np.asarray(list(x.shape)[::-1], np.int64)
        m = np.ascontiguousarray(mask, np.float32) if mask is not None else None
        out = np.empty((q.shape[0], q.shape[2], q.shape[1], v.shape[3]), np.float32)
        rc = self.lib.ref_flash_attn_ext(_ptr(q), _ptr(ne(q)), _ptr(k), _ptr(ne(k)), _ptr(v), _ptr(ne(v)), _ptr(m) if m is not None else None,
                                         m.shape[0] if m is not None else 0, scale, max_bias, softcap, _ptr(out), n_threads)
        if rc != 0:
            raise RuntimeError(f"ref_flash_attn_ext failed: {rc}")
        return out

    def flash_attn_ext_q8_0(self, q, k, v, mask, scale: float, max_bias: float = 0.0, softcap: float = 0.0, n_threads: int = 1, kv_type: int = 8):
        """FLASH_ATTN_EXT on a Q8_0 K / V cache on the reference CPU backend: k / v (f32) are quantized by the reference quantizer
        (ggml_quantize_chunk).  Returns (out [B, N, H, DV] f32, k blocks uint8 [B, Hk, n_kv, DK/32*34], v blocks likewise)."""
        q = np.ascontiguousarray(q, np.float32); k = np.ascontiguousarray(k, np.float32); v = np.ascontiguousarray(v, np.float32)
        ne = lambda x: np.asarray(list(x.shape)[::-1], np.int64)
        m = np.ascontiguousarray(mask, np.float32) if mask is not None else None
        out = np.empty((q.shape[0], q.shape[2], q.shape[1], v.shape[3]), np.float32)
        bb = 34 if kv_type == Q8_0 else 18                               # (Q8_0 or Q4_0 blocks)
        kb = np.empty(k.shape[:-1] + (k.shape[-1] // 32 * bb,), np.uint8); vb = np.empty(v.shape[:-1] + (v.shape[-1] // 32 * bb,), np.uint8)
        rc = self.lib.ref_flash_attn_ext_t(_ptr(q), _ptr(ne(q)), _ptr(k), _ptr(ne(k)), _ptr(v), _ptr(ne(v)), _ptr(m) if m is not None else None,
                                           m.shape[0] if m is not None else 0, scale, max_bias, softcap, kv_type, _ptr(out), _ptr(kb), _ptr(vb), n_threads)
        if rc != 0:
            raise RuntimeError(f"ref_flash_attn_ext_t failed: {rc}")
        return out, kb, vb

    def bench_chain(self, types, Ms, Ks, N: int, n_threads: int, warmup: int, iters: int) -> float:
        ty = np.asarray(types, np.int32); ms = np.asarray(Ms, np.int64); ks = np.asarray(Ks, np.int64)
        return float(self.lib.ref_bench_chain(len(ty), _ptr(ty), _ptr(ms), _ptr(ks), N, n_threads, warmup, iters))


def best_ref_variant() -> str | None:
    """The fastest reference build this host can execute (for the CPU baseline)."""
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        flags = ""
    if ref_available("avx2") and all(f in flags for f in (" avx2", " fma", " f16c", " bmi2")):
        return "avx2"
    if ref_available("scalar"):
        return "scalar"
    return None
