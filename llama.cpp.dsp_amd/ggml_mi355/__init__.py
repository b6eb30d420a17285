"""ggml_mi355 -- Python host side of the MI355X quantized-matmul path.

A thin ctypes binding of the C-ABI in include/mi355q.h (libmi355q.so, hand-written HIP for gfx950)
plus a small mirror of the reference's operator interface for this path:

    ggml_mul_mat(a, b)       -> mul_mat(w, x)          ggml/src/ggml.c:2730-2745
    ggml_mul_mat_id(as,b,ids)-> mul_mat_id(w, x, ids)  ggml/src/ggml.c:2771-2796
    ggml_backend_tensor_set  -> QWeight.from_host      ggml/src/ggml-backend.cpp (tensor_set -> buffer.set_tensor)
    ggml_backend_tensor_get  -> QWeight.to_host

Shapes follow ggml: a weight of ggml shape [ne00=K, ne01=M(, ne02=n_expert)] is `M` packed rows of
`K` elements; activations x are f32 [N, K] row-major (ggml [K, N]); the result is f32 [N, M].

torch is used ONLY for device memory and streams.  There is no CPU fallback: importing works anywhere,
but every compute call raises if libmi355q.so or a gfx950 device is missing.
"""
from __future__ import annotations

import ctypes as C
import sys
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
import os as _os
LIB_PATH = Path(_os.environ["MI355Q_LIB"]) if _os.environ.get("MI355Q_LIB") else _HERE.parent / "lib" / "libmi355q.so"   # MI355Q_LIB: diagnostic builds only

# ggml type ids (ggml/include/ggml.h)
F32 = 0
Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q8_1 = 2, 3, 6, 7, 8, 9
Q2_K, Q3_K, Q4_K, Q5_K, Q6_K, Q8_K = 10, 11, 12, 13, 14, 15
IQ2_XXS, IQ2_XS, IQ3_XXS, IQ1_S, IQ4_NL, IQ3_S, IQ2_S, IQ4_XS, IQ1_M = 16, 17, 18, 19, 20, 21, 22, 23, 29
TYPE_NAMES = {Q4_0: "q4_0", Q4_1: "q4_1", Q5_0: "q5_0", Q5_1: "q5_1", Q8_0: "q8_0", Q8_1: "q8_1",
              Q2_K: "q2_K", Q3_K: "q3_K", Q4_K: "q4_K", Q5_K: "q5_K", Q6_K: "q6_K", Q8_K: "q8_K",
              IQ4_NL: "iq4_nl", IQ4_XS: "iq4_xs", IQ2_XXS: "iq2_xxs", IQ2_XS: "iq2_xs", IQ2_S: "iq2_s", IQ3_XXS: "iq3_xxs",
              IQ3_S: "iq3_s", IQ1_S: "iq1_s", IQ1_M: "iq1_m"}
WEIGHT_TYPES = [Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q2_K, Q3_K, Q4_K, Q5_K, Q6_K, IQ4_NL, IQ4_XS,
                IQ2_XXS, IQ2_XS, IQ2_S, IQ3_XXS, IQ3_S, IQ1_S, IQ1_M]

FLAG_ROUND_AWAY, FLAG_ROUND_EVEN = 0, 1

# every symbol include/mi355q.h declares (tests check that the library exports all of them)
ABI_SYMBOLS = [
    "mi355q_api_version", "mi355q_device_count", "mi355q_set_device", "mi355q_device_info", "mi355q_last_error",
    "mi355q_type_supported", "mi355q_blck_size", "mi355q_type_size", "mi355q_row_size", "mi355q_act_type",
    "mi355q_weights_are_planar",
    "mi355q_malloc", "mi355q_free", "mi355q_memset", "mi355q_memcpy_h2d", "mi355q_memcpy_d2h", "mi355q_memcpy_d2d",
    "mi355q_host_malloc", "mi355q_host_free", "mi355q_memcpy_peer", "mi355q_event_create", "mi355q_event_destroy", "mi355q_event_record",
    "mi355q_event_wait", "mi355q_event_synchronize",
    "mi355q_stream_create", "mi355q_stream_destroy", "mi355q_stream_synchronize", "mi355q_device_synchronize",
    "mi355q_weights_upload", "mi355q_weights_download", "mi355q_weights_pack_d2d", "mi355q_weights_unpack_d2d",
    "mi355q_quantize_act",
    "mi355q_mul_mat_workspace", "mi355q_mul_mat", "mi355q_mul_mat_multi",
    "mi355q_mul_mat_id_workspace", "mi355q_mul_mat_id",
    "mi355q_plan_create", "mi355q_plan_run", "mi355q_plan_status", "mi355q_plan_status_async", "mi355q_plan_debug_set_runs", "mi355q_plan_debug_words", "mi355q_plan_weight_bytes", "mi355q_plan_launch_stages",
    "mi355q_plan_destroy",
    "mi355q_op_bin_bcast", "mi355q_op_unary", "mi355q_op_rms_norm", "mi355q_op_cpy", "mi355q_op_soft_max",
    "mi355q_op_rope", "mi355q_op_mul_mat_f", "mi355q_op_get_rows", "mi355q_op_scale", "mi355q_op_cpy_indirect", "mi355q_op_argsort", "mi355q_op_sum_rows",
    "mi355q_graph_capture_begin", "mi355q_graph_capture_end", "mi355q_graph_launch", "mi355q_graph_destroy",
    "mi355q_op_add_rms_norm_mul", "mi355q_op_unary_mul", "mi355q_op_flash_attn_ext", "mi355q_op_flash_attn_ext_workspace",
]


class Mi355qError(RuntimeError):
    pass


class _Mat(C.Structure):
    _fields_ = [("type", C.c_int), ("w", C.c_void_p), ("w_stride", C.c_int64), ("y", C.c_void_p),
                ("y_stride", C.c_int64), ("m", C.c_int64)]


class _RopeParams(C.Structure):
    _fields_ = [("n_dims", C.c_int), ("mode", C.c_int), ("n_ctx_orig", C.c_int), ("freq_base", C.c_float), ("freq_scale", C.c_float),
                ("ext_factor", C.c_float), ("attn_factor", C.c_float), ("beta_fast", C.c_float), ("beta_slow", C.c_float)]


class _Attn(C.Structure):          # mi355q_attn
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("pos", C.c_void_p), ("rope", _RopeParams),
                ("freq_factors", C.c_void_p), ("k_cache", C.c_void_p), ("v_cache", C.c_void_p),
                ("k_nb_pos", C.c_int64), ("k_nb_head", C.c_int64), ("v_nb_pos", C.c_int64), ("v_nb_dim", C.c_int64), ("v_nb_head", C.c_int64),
                ("k_dst", C.c_void_p), ("v_dst", C.c_void_p), ("v_dst_nb", C.c_int64), ("mask", C.c_void_p), ("mask_f16", C.c_int),
                ("n_head", C.c_int), ("n_head_kv", C.c_int), ("head_dim", C.c_int), ("n_kv", C.c_int), ("scale", C.c_float), ("out", C.c_void_p),
                ("n_kv_dev", C.c_void_p), ("q_id", C.c_int64), ("k_id", C.c_int64), ("v_id", C.c_int64), ("out_id", C.c_int64)]


class _Stage(C.Structure):         # mi355q_stage
    _fields_ = [("mats", _Mat * 4), ("n_mats", C.c_int), ("flags", C.c_int), ("x", C.c_void_p), ("k", C.c_int64),
                ("kind", C.c_int), ("x_kind", C.c_int), ("x_unary", C.c_int), ("eps", C.c_float),
                ("x1", C.c_void_p), ("norm_w", C.c_void_p), ("sum_out", C.c_void_p), ("attn", C.POINTER(_Attn)),
                ("y_id", C.c_int64 * 4), ("sum_id", C.c_int64), ("x_id", C.c_int64), ("x1_id", C.c_int64), ("y_kind", C.c_int), ("y_unary", C.c_int), ("x_out", C.c_void_p)]


STAGE_DEPENDS, STAGE_NO_PLAIN = 0x1, 0x2
STAGE_GEMV, STAGE_ATTN = 0, 1
X_PLAIN, X_NORM, X_UNARY_MUL = 0, 1, 2
Y_ROWS, Y_UNARY_MUL = 0, 1


class _Tensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("type", C.c_int), ("ne", C.c_int64 * 4), ("nb", C.c_int64 * 4)]


OP_ADD, OP_SUB, OP_MUL, OP_DIV = 1, 2, 3, 4
UNARY_SILU, UNARY_RELU, UNARY_SIGMOID, UNARY_TANH, UNARY_NEG, UNARY_ABS = 1, 2, 3, 4, 5, 6

_lib = None


def lib() -> C.CDLL:
    """Load libmi355q.so (built by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise Mi355qError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for this path)")
    L = C.CDLL(str(LIB_PATH))
    i64, i32, vp, sz = C.c_int64, C.c_int, C.c_void_p, C.c_size_t
    L.mi355q_last_error.restype = C.c_char_p
    for f in ("mi355q_blck_size", "mi355q_type_size"):
        getattr(L, f).restype = i64; getattr(L, f).argtypes = [i32]
    L.mi355q_row_size.restype = i64; L.mi355q_row_size.argtypes = [i32, i64]
    L.mi355q_act_type.argtypes = [i32]
    L.mi355q_type_supported.argtypes = [i32]
    L.mi355q_weights_are_planar.argtypes = [i32, i64]
    L.mi355q_device_info.argtypes = [i32, C.c_char_p, sz, C.POINTER(sz), C.POINTER(sz), C.POINTER(i32)]
    L.mi355q_weights_upload.argtypes = [i32, vp, vp, i64, i64, vp]
    L.mi355q_weights_download.argtypes = [i32, vp, vp, i64, i64, vp]
    L.mi355q_weights_pack_d2d.argtypes = [i32, vp, vp, i64, i64, vp]
    L.mi355q_weights_unpack_d2d.argtypes = [i32, vp, vp, i64, i64, vp]
    L.mi355q_quantize_act.argtypes = [i32, vp, i64, vp, i64, i64, i32, vp]
    L.mi355q_mul_mat_workspace.restype = sz; L.mi355q_mul_mat_workspace.argtypes = [i32, i64, i64, i64]
    L.mi355q_mul_mat.argtypes = [i32, vp, i64, vp, i64, vp, i64, i64, i64, i64, vp, sz, i32, vp]
    L.mi355q_mul_mat_multi.argtypes = [C.POINTER(_Mat), i32, vp, i64, i64, i64, vp, sz, i32, vp]
    L.mi355q_plan_create.argtypes = [C.POINTER(vp), C.POINTER(_Stage), i32, i32]
    L.mi355q_plan_run.argtypes = [vp, vp]
    L.mi355q_plan_status.argtypes = [vp]
    L.mi355q_plan_debug_set_runs.argtypes = [vp, C.c_ulonglong]
    L.mi355q_plan_debug_words.argtypes = [vp, C.POINTER(C.c_uint)]
    L.mi355q_plan_destroy.argtypes = [vp]
    L.mi355q_plan_weight_bytes.restype = i64; L.mi355q_plan_weight_bytes.argtypes = [vp]
    L.mi355q_plan_launch_stages.argtypes = [vp]
    TP = C.POINTER(_Tensor)
    L.mi355q_op_bin_bcast.argtypes = [i32, TP, TP, TP, vp]
    L.mi355q_op_unary.argtypes = [i32, TP, TP, vp]
    L.mi355q_op_rms_norm.argtypes = [TP, TP, C.c_float, vp]
    L.mi355q_op_cpy.argtypes = [TP, TP, vp]
    L.mi355q_op_add_rms_norm_mul.argtypes = [TP, TP, TP, vp, TP, C.c_float, vp]
    L.mi355q_op_unary_mul.argtypes = [i32, TP, TP, TP, vp]
    L.mi355q_op_flash_attn_ext.argtypes = [TP, TP, TP, TP, TP, C.c_float, C.c_float, C.c_float, vp, sz, vp]
    L.mi355q_op_flash_attn_ext_workspace.restype = sz; L.mi355q_op_flash_attn_ext_workspace.argtypes = [i64, i64, i64, i64, i64]
    L.mi355q_op_soft_max.argtypes = [TP, TP, TP, C.c_float, C.c_float, vp]
    L.mi355q_op_rope.argtypes = [TP, vp, vp, TP, C.POINTER(_RopeParams), vp]
    L.mi355q_op_mul_mat_f.argtypes = [TP, TP, TP, vp]
    L.mi355q_op_get_rows.argtypes = [TP, TP, TP, vp]
    L.mi355q_op_scale.argtypes = [TP, TP, C.c_float, vp]
    L.mi355q_op_argsort.argtypes = [TP, TP, i32, vp]
    L.mi355q_op_sum_rows.argtypes = [TP, TP, vp]
    L.mi355q_mul_mat_id_workspace.restype = sz; L.mi355q_mul_mat_id_workspace.argtypes = [i32, i64, i64, i64, i64, i64, i64]
    L.mi355q_mul_mat_id.argtypes = [i32, vp, i64, i64, i64, vp, i64, i64, i64, vp, i64, vp, i64, i64, i64, i64, vp, sz, i32, vp]
    _lib = L
    return L


def shutdown() -> None:
    """Explicit, ordered teardown of the binding: drop the cached workspaces, finish the device's work and UNLOAD libmi355q.so.
    Unloading runs the library's fat-binary unregistration (the finalizer hipcc puts into every code-object-carrying shared object) now,
    while the HIP runtime is fully alive, instead of from an exit handler of the dying process -- where, under a profiler that has already
    finalized, it was seen to touch freed runtime state.  Every Plan / QWeight must have been released by the caller; lib() reloads on demand."""
    global _lib
    _ws_cache.clear()
    if _lib is None:
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass
    import _ctypes
    h = _lib._handle
    _lib = None
    _ctypes.dlclose(h)


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise Mi355qError(f"{what} failed ({rc}): {lib().mi355q_last_error().decode(errors='replace')}")


def row_size(t: int, k: int) -> int:
    return int(lib().mi355q_row_size(t, k))


def act_type(t: int) -> int:
    return int(lib().mi355q_act_type(t))


def is_planar(t: int, k: int) -> bool:
    return bool(lib().mi355q_weights_are_planar(t, k))


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise Mi355qError("no GPU visible: the MI355X path has no CPU fallback")
    if lib().mi355q_device_count() < 1:
        raise Mi355qError("no gfx950 device: libmi355q.so is built for MI355X only")
    return torch


def _stream(torch) -> int:
    return int(torch.cuda.current_stream().cuda_stream)


class QWeight:
    """A quantized weight resident in HBM in device layout.  ggml shape [K, M] or [K, M, n_expert]."""

    def __init__(self, t: int, data, M: int, K: int, n_expert: int = 1):
        self.type, self.data, self.M, self.K, self.n_expert = t, data, M, K, n_expert
        self.row_bytes = row_size(t, K)

    @property
    def nbytes(self) -> int:
        return self.row_bytes * self.M * self.n_expert

    @staticmethod
    def from_host(t: int, rows: np.ndarray, M: int, K: int, n_expert: int = 1, device: str = "cuda") -> "QWeight":
        """Upload canonical ggml rows (uint8 [M*n_expert, row_size]) -- set_tensor."""
        torch = _torch()
        rows = np.ascontiguousarray(rows).view(np.uint8).reshape(-1)
        rb = row_size(t, K)
        if rb <= 0 or rows.size != rb * M * n_expert:
            raise Mi355qError(f"from_host: {rows.size} bytes given, expected {rb}*{M}*{n_expert}")
        buf = torch.empty(rows.size + 64, dtype=torch.uint8, device=device)   # torch allocations are >= 256-B aligned
        _check(lib().mi355q_weights_upload(t, buf.data_ptr(), rows.ctypes.data, M * n_expert, K, _stream(torch)), "weights_upload")
        return QWeight(t, buf, M, K, n_expert)

    def to_host(self) -> np.ndarray:
        """Canonical ggml rows back on the host -- get_tensor."""
        torch = _torch()
        out = np.empty((self.M * self.n_expert, self.row_bytes), np.uint8)
        _check(lib().mi355q_weights_download(self.type, out.ctypes.data, self.data.data_ptr(), self.M * self.n_expert, self.K, _stream(torch)), "weights_download")
        return out


_ws_cache: dict = {}


def _workspace(torch, nbytes: int, device):
    if nbytes == 0:
        return None, 0
    key = (str(device), int(torch.cuda.current_stream().cuda_stream))
    cur = _ws_cache.get(key)
    if cur is None or cur.numel() < nbytes:
        cur = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = cur
    return cur, cur.numel()


def quantize_act(act_t: int, x, flags: int = 0):
    """f32 [N, K] device tensor -> canonical activation blocks, uint8 [N, row_size(act_t, K)] on the device."""
    torch = _torch()
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    n, k = x.shape
    out = torch.empty((n, row_size(act_t, k)), dtype=torch.uint8, device=x.device)
    _check(lib().mi355q_quantize_act(act_t, x.data_ptr(), x.stride(0) * 4, out.data_ptr(), n, k, flags, _stream(torch)), "quantize_act")
    return out


def mul_mat(w: QWeight, x, out=None, flags: int = 0):
    """y[N, M] = x[N, K] . W[M, K]^T  (GGML_OP_MUL_MAT, quantized src0, f32 src1)."""
    torch = _torch()
    assert w.n_expert == 1
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == w.K
    n = x.shape[0]
    y = out if out is not None else torch.empty((n, w.M), dtype=torch.float32, device=x.device)
    ws, wsb = _workspace(torch, int(lib().mi355q_mul_mat_workspace(w.type, w.M, n, w.K)), x.device)
    _check(lib().mi355q_mul_mat(w.type, w.data.data_ptr(), w.row_bytes, x.data_ptr(), x.stride(0) * 4,
                                y.data_ptr(), y.stride(0) * 4, w.M, n, w.K,
                                ws.data_ptr() if ws is not None else None, wsb, flags, _stream(torch)), "mul_mat")
    return y


def mul_mat_multi(ws_: list, x, outs=None, flags: int = 0):
    """Several weights against the same activations in one launch (wq/wk/wv, ffn_gate/ffn_up)."""
    torch = _torch()
    n, k = x.shape
    assert x.dtype == torch.float32 and x.stride(1) == 1 and all(w.K == k and w.n_expert == 1 for w in ws_)
    ys = outs if outs is not None else [torch.empty((n, w.M), dtype=torch.float32, device=x.device) for w in ws_]
    arr = (_Mat * len(ws_))()
    wsb = 0
    for i, (w, y) in enumerate(zip(ws_, ys)):
        arr[i] = _Mat(w.type, w.data.data_ptr(), w.row_bytes, y.data_ptr(), y.stride(0) * 4, w.M)
        wsb = max(wsb, int(lib().mi355q_mul_mat_workspace(w.type, w.M, n, k)))
    ws, wsb = _workspace(torch, wsb, x.device)
    _check(lib().mi355q_mul_mat_multi(arr, len(ws_), x.data_ptr(), x.stride(0) * 4, n, k,
                                      ws.data_ptr() if ws is not None else None, wsb, flags, _stream(torch)), "mul_mat_multi")
    return ys


class Plan:
    """A token's dependent chain as ONE persistent launch (mi355q_plan_*).

    stages: a list whose entries are
      * (weights: list[QWeight] (<= 4, same K), x: f32 [1, K] or [K], ys: list of f32 [1, M_i] / [M_i], depends)   -- a plain GEMV stage
        (`depends` is accepted for API version 1 and ignored: an x that is an earlier stage's output is found by its address), or
      * dict(ws=, ys=, x=, x_kind=X_PLAIN|X_NORM|X_UNARY_MUL, x1=None, norm_w=None, eps=0.0, sum_out=None, unary=UNARY_SILU, no_plain=False,
             y_kind=Y_ROWS|Y_UNARY_MUL (two weights: ys[0] = y_unary(W0 x) * (W1 x), ys[1] unused), y_unary=UNARY_SILU), or
      * dict(attn=dict(q=, k=, v=, pos=, rope=dict(n_dims=, mode=0, n_ctx_orig=0, freq_base=10000.0, freq_scale=1.0, ext_factor=0.0,
                       attn_factor=1.0, beta_fast=32.0, beta_slow=1.0), freq_factors=None, k_cache=, v_cache=, k_nb_pos=, k_nb_head=, v_nb_pos=,
                       v_nb_dim=, v_nb_head=, k_dst=, v_dst= (int64 device tensors holding the destination ADDRESSES), v_dst_nb=, mask=None,
                       n_head=, n_head_kv=, head_dim=, n_kv=, scale=, out=), no_plain=False)                 -- see include/mi355q.h mi355q_attn.
    The tensors are referenced by address: keep them alive (the Plan holds references) and do not move them."""

    def __init__(self, stages, flags: int = 0):
        torch = _torch()
        arr = (_Stage * len(stages))()
        self._keep = []
        ptr = lambda t: t.data_ptr() if t is not None else None
        for i, sd in enumerate(stages):
            st = _Stage()
            if isinstance(sd, dict) and "attn" in sd:
                a = sd["attn"]
                at = _Attn()
                for name in ("q", "k", "v", "pos", "freq_factors", "k_cache", "v_cache", "k_dst", "v_dst", "mask", "out", "n_kv_dev"):
                    setattr(at, name, ptr(a.get(name)))
                rp = dict(n_dims=a["head_dim"], mode=0, n_ctx_orig=0, freq_base=10000.0, freq_scale=1.0, ext_factor=0.0, attn_factor=1.0, beta_fast=32.0, beta_slow=1.0)
                rp.update(a.get("rope", {}))
                at.rope = _RopeParams(rp["n_dims"], rp["mode"], rp["n_ctx_orig"], rp["freq_base"], rp["freq_scale"], rp["ext_factor"], rp["attn_factor"], rp["beta_fast"], rp["beta_slow"])
                for name in ("k_nb_pos", "k_nb_head", "v_nb_pos", "v_nb_dim", "v_nb_head", "v_dst_nb", "n_head", "n_head_kv", "head_dim", "n_kv"):
                    setattr(at, name, int(a[name]))
                at.mask_f16 = 1 if (a.get("mask") is not None and a["mask"].dtype == torch.float16) else 0
                at.scale = float(a["scale"])
                st.kind = STAGE_ATTN; st.attn = C.pointer(at); st.flags = STAGE_NO_PLAIN if sd.get("no_plain") else 0
                self._keep.append((a, at))
            else:
                if isinstance(sd, dict):
                    ws_, x, ys = sd["ws"], sd["x"], sd["ys"]
                    st.x_kind = sd.get("x_kind", X_PLAIN); st.x_unary = sd.get("unary", UNARY_SILU); st.eps = float(sd.get("eps", 0.0))
                    st.x1 = ptr(sd.get("x1")); st.norm_w = ptr(sd.get("norm_w")); st.sum_out = ptr(sd.get("sum_out")); st.x_out = ptr(sd.get("x_out"))
                    st.flags = STAGE_NO_PLAIN if sd.get("no_plain") else 0
                    st.y_kind = sd.get("y_kind", Y_ROWS); st.y_unary = sd.get("y_unary", UNARY_SILU)
                    for t in (sd.get("x1"), sd.get("norm_w"), sd.get("sum_out"), sd.get("x_out")):
                        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == x.shape[-1])
                else:
                    ws_, x, ys, depends = sd
                    st.flags = STAGE_DEPENDS if depends else 0
                k = x.shape[-1]
                assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() == k
                assert 1 <= len(ws_) <= 4 and len(ws_) == len(ys)
                for j, (w, y) in enumerate(zip(ws_, ys)):
                    assert w.K == k and w.n_expert == 1 and y.dtype == torch.float32 and y.is_contiguous() and y.numel() == w.M
                    st.mats[j] = _Mat(w.type, w.data.data_ptr(), w.row_bytes, y.data_ptr(), w.M * 4, w.M)
                st.kind = STAGE_GEMV; st.n_mats = len(ws_); st.x = x.data_ptr(); st.k = k
                self._keep.append(sd)
            arr[i] = st
        h = C.c_void_p()
        _check(lib().mi355q_plan_create(C.byref(h), arr, len(stages), flags), "plan_create")
        self._h = h

    @property
    def weight_bytes(self) -> int:
        return int(lib().mi355q_plan_weight_bytes(self._h))

    @property
    def launch_stages(self) -> int:
        return int(lib().mi355q_plan_launch_stages(self._h))

    def run(self):
        _check(lib().mi355q_plan_run(self._h, _stream(_torch())), "plan_run")

    def status(self) -> int:
        st = int(lib().mi355q_plan_status(self._h))
        if st != 0:
            w = (C.c_uint * 32)()
            lib().mi355q_plan_debug_words(self._h, w)
            print("mi355q plan aborted: sync words", [hex(x) for x in w], file=sys.stderr)
        return st

    def debug_set_runs(self, runs: int):
        """Test hook: set the run counter the granule tags' epoch derives from (near 2**32 / (stages + 1) the next run resets the granules)."""
        _check(lib().mi355q_plan_debug_set_runs(self._h, C.c_ulonglong(runs)), "plan_debug_set_runs")

    def close(self):
        if getattr(self, "_h", None):
            lib().mi355q_plan_destroy(self._h)
            self._h = None

    def __del__(self, _sys=sys):
        if _sys is None or _sys.is_finalizing():     # the HIP runtime may already be gone: leave the device memory to process teardown
            return
        try:
            self.close()
        except Exception:
            pass


def _td(t) -> _Tensor:
    """torch tensor (f32/f16, <= 4 dims, torch dims are ggml dims reversed) -> mi355q_tensor with ggml-ordered ne/nb."""
    torch = _torch()
    assert t.dtype in (torch.float32, torch.float16) and 1 <= t.dim() <= 4
    shape = list(t.shape)[::-1] + [1] * (4 - t.dim())
    es = t.element_size()
    strides = [s * es for s in list(t.stride())[::-1]]
    nb = strides + [0] * (4 - t.dim())
    for i in range(t.dim(), 4):                       # ggml convention for the padded dims: nb[i] = nb[i-1] * ne[i-1]
        nb[i] = nb[i - 1] * shape[i - 1]
    d = _Tensor(); d.data = t.data_ptr(); d.type = 0 if t.dtype == torch.float32 else 1
    for i in range(4):
        d.ne[i] = shape[i]; d.nb[i] = nb[i]
    return d


def _td_q8_0(t, kv: str = "q8_0") -> _Tensor:
    """uint8 torch tensor holding contiguous block_q8_0 (34-byte) or block_q4_0 (18-byte) rows, shape [..., n / 32 * bb] (<= 4 dims) -> mi355q_tensor of
    that type with ne[0] = n."""
    torch = _torch()
    bb, code = (34, 8) if kv == "q8_0" else (18, 2)
    assert t.dtype == torch.uint8 and t.is_contiguous() and 1 <= t.dim() <= 4 and t.shape[-1] % bb == 0
    shape = list(t.shape)[::-1] + [1] * (4 - t.dim())
    d = _Tensor(); d.data = t.data_ptr(); d.type = code
    d.ne[0] = shape[0] // bb * 32; d.nb[0] = bb
    d.nb[1] = shape[0]
    for i in range(1, 4):
        d.ne[i] = shape[i]
        if i > 1:
            d.nb[i] = d.nb[i - 1] * shape[i - 1]
    return d


def op_bin_bcast(op: int, a, b, out=None):
    """GGML_OP_ADD/SUB/MUL/DIV: out = a (op) b with ggml broadcasting of b over a (shapes given torch-style)."""
    torch = _torch()
    out = out if out is not None else torch.empty_like(a, memory_format=torch.contiguous_format)
    _check(lib().mi355q_op_bin_bcast(op, C.byref(_td(a)), C.byref(_td(b)), C.byref(_td(out)), _stream(torch)), "op_bin_bcast")
    return out


def op_unary(uop: int, a, out=None):
    torch = _torch()
    out = out if out is not None else torch.empty_like(a, memory_format=torch.contiguous_format)
    _check(lib().mi355q_op_unary(uop, C.byref(_td(a)), C.byref(_td(out)), _stream(torch)), "op_unary")
    return out


def op_rms_norm(a, eps: float, out=None):
    torch = _torch()
    out = out if out is not None else torch.empty_like(a, memory_format=torch.contiguous_format)
    _check(lib().mi355q_op_rms_norm(C.byref(_td(a)), C.byref(_td(out)), eps, _stream(torch)), "op_rms_norm")
    return out


def op_add_rms_norm_mul(a, eps: float, b=None, weight=None, want_sum: bool = False):
    """Fused [a + b ->] rms_norm [-> * weight]; returns out or (out, sum).  Bit-identical to the separate ops (include/mi355q.h)."""
    torch = _torch()
    out = torch.empty_like(a, memory_format=torch.contiguous_format)
    s = torch.empty_like(a, memory_format=torch.contiguous_format) if (want_sum and b is not None) else None
    if weight is not None and (weight.dtype != torch.float32 or not weight.is_contiguous() or weight.numel() != a.shape[-1]):
        raise ValueError("weight must be a contiguous f32 vector of the row length")
    _check(lib().mi355q_op_add_rms_norm_mul(C.byref(_td(a)), C.byref(_td(b)) if b is not None else None, C.byref(_td(s)) if s is not None else None,
                                            weight.data_ptr() if weight is not None else None, C.byref(_td(out)), eps,
                                            _stream(torch)), "op_add_rms_norm_mul")
    return (out, s) if s is not None else out


def op_unary_mul(uop: int, a, b):
    torch = _torch()
    out = torch.empty_like(a, memory_format=torch.contiguous_format)
    _check(lib().mi355q_op_unary_mul(uop, C.byref(_td(a)), C.byref(_td(b)), C.byref(_td(out)), _stream(torch)), "op_unary_mul")
    return out


def op_flash_attn_ext(q, k, v, mask, scale: float, max_bias: float = 0.0, logit_softcap: float = 0.0, split: bool = True, kv: str = "q8_0"):
    """GGML_OP_FLASH_ATTN_EXT: q f32 [B, H, N, DK] (torch order), k f16 [Bk, Hk, n_kv, DK], v f16 [Bv, Hv, n_kv, DV], mask f16 [>= N, >= n_kv]
    or None -> f32 [B, N, H, DV]."""
    torch = _torch()
    out = torch.empty((q.shape[0], q.shape[2], q.shape[1], v.shape[3]), dtype=torch.float32, device=q.device)
    ws, wsb = (_workspace(torch, int(lib().mi355q_op_flash_attn_ext_workspace(v.shape[3], q.shape[2], q.shape[1], q.shape[0], k.shape[2])), q.device)
               if split else (None, 0))
    tdkv = (lambda t: _td_q8_0(t, kv)) if k.dtype == torch.uint8 else _td      # (uint8: block_q8_0 / block_q4_0 rows, a quantized cache)
    if k.dtype == torch.uint8:
        out = torch.empty((q.shape[0], q.shape[2], q.shape[1], v.shape[3] // (34 if kv == "q8_0" else 18) * 32), dtype=torch.float32, device=q.device)
    _check(lib().mi355q_op_flash_attn_ext(C.byref(_td(q)), C.byref(tdkv(k)), C.byref(tdkv(v)), C.byref(_td(mask)) if mask is not None else None,
                                          C.byref(_td(out)), scale, max_bias, logit_softcap, ws.data_ptr() if ws is not None else None, wsb,
                                          _stream(torch)), "op_flash_attn_ext")
    return out


def op_cpy(a, out, kv: str = "q8_0"):
    """GGML_OP_CPY; a uint8 `out` of shape [..., n / 32 * 34] (kv "q8_0") or [..., n / 32 * 18] ("q4_0") is a quantized destination (KV cache rows)."""
    td_out = _td_q8_0(out, kv) if out.dtype == _torch().uint8 else _td(out)
    _check(lib().mi355q_op_cpy(C.byref(_td(a)), C.byref(td_out), _stream(_torch())), "op_cpy")
    return out


def op_soft_max(a, mask=None, scale: float = 1.0, max_bias: float = 0.0, out=None):
    torch = _torch()
    out = out if out is not None else torch.empty_like(a, memory_format=torch.contiguous_format)
    _check(lib().mi355q_op_soft_max(C.byref(_td(a)), C.byref(_td(mask)) if mask is not None else None, C.byref(_td(out)), scale, max_bias,
                                    _stream(torch)), "op_soft_max")
    return out


def op_rope(a, pos, n_dims: int, mode: int = 0, freq_factors=None, n_ctx_orig: int = 0, freq_base: float = 10000.0, freq_scale: float = 1.0,
            ext_factor: float = 0.0, attn_factor: float = 1.0, beta_fast: float = 32.0, beta_slow: float = 1.0, out=None):
    """GGML_OP_ROPE on a [ne3, ne2 (positions), ne1 (heads), ne0] tensor (torch order); pos: int32 [ne2]."""
    torch = _torch()
    out = out if out is not None else torch.empty_like(a, memory_format=torch.contiguous_format)
    p = _RopeParams(n_dims, mode, n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow)
    _check(lib().mi355q_op_rope(C.byref(_td(a)), pos.data_ptr(), freq_factors.data_ptr() if freq_factors is not None else None,
                                C.byref(_td(out)), C.byref(p), _stream(torch)), "op_rope")
    return out


def op_get_rows(a, ids, out=None):
    """GGML_OP_GET_ROWS: a [.., n_rows, ne0] (f32/f16), ids int32 [.., n_ids] -> f32 [.., n_ids, ne0]."""
    torch = _torch()
    assert ids.dtype == torch.int32
    if out is None:
        out = torch.empty(tuple(ids.shape) + (a.shape[-1],), dtype=torch.float32, device=a.device)
    t = _Tensor(); t.data = ids.data_ptr(); t.type = 0
    shape = list(ids.shape)[::-1] + [1] * (4 - ids.dim()); st = [s_ * 4 for s_ in list(ids.stride())[::-1]]
    for i in range(4):
        t.ne[i] = shape[i]; t.nb[i] = st[i] if i < ids.dim() else (t.nb[i - 1] * t.ne[i - 1])
    _check(lib().mi355q_op_get_rows(C.byref(_td(a)), C.byref(t), C.byref(_td(out)), _stream(torch)), "op_get_rows")
    return out


def op_argsort(a, descending: bool = False):
    """GGML_OP_ARGSORT per row of the last dimension: int32 indices, rank order (ties keep index order)."""
    torch = _torch()
    out = torch.empty(a.shape, dtype=torch.int32, device=a.device)
    d = _td(out.view(torch.float32)); d.type = 0
    _check(lib().mi355q_op_argsort(C.byref(_td(a)), C.byref(d), 1 if descending else 0, _stream(torch)), "op_argsort")
    return out


def op_sum_rows(a):
    """GGML_OP_SUM_ROWS: sums over the last (ggml: first) dimension, keeping it with size 1."""
    torch = _torch()
    out = torch.empty(tuple(a.shape[:-1]) + (1,), dtype=torch.float32, device=a.device)
    _check(lib().mi355q_op_sum_rows(C.byref(_td(a)), C.byref(_td(out)), _stream(torch)), "op_sum_rows")
    return out


def op_scale(a, scale: float, out=None):
    torch = _torch()
    out = out if out is not None else torch.empty_like(a, memory_format=torch.contiguous_format)
    _check(lib().mi355q_op_scale(C.byref(_td(a)), C.byref(_td(out)), scale, _stream(torch)), "op_scale")
    return out


def op_mul_mat_f(a, b, out=None):
    """GGML_OP_MUL_MAT with an f16/f32 src0: a [.., M, K], b [.., N, K] (torch order) -> [.., N, M] f32."""
    torch = _torch()
    if out is None:
        out = torch.empty(tuple(b.shape[:-2]) + (b.shape[-2], a.shape[-2]), dtype=torch.float32, device=b.device)
    _check(lib().mi355q_op_mul_mat_f(C.byref(_td(a)), C.byref(_td(b)), C.byref(_td(out)), _stream(torch)), "op_mul_mat_f")
    return out


def mul_mat_id(w: QWeight, x, ids, flags: int = 0):
    """y[T, U, M] = W[ids[T, U]] . x[T, U % x_ne1, :]   (GGML_OP_MUL_MAT_ID).
    x: f32 [T, x_ne1, K]; ids: int32 [T, U] on the device (read there: no host round trip)."""
    torch = _torch()
    assert x.dtype == torch.float32 and x.dim() == 3 and x.stride(2) == 1 and x.shape[2] == w.K
    assert ids.dtype == torch.int32 and ids.dim() == 2 and ids.stride(1) == 1 and ids.shape[0] == x.shape[0]
    n_tok, x_ne1, _ = x.shape
    n_used = ids.shape[1]
    y = torch.empty((n_tok, n_used, w.M), dtype=torch.float32, device=x.device)
    ws, wsb = _workspace(torch, int(lib().mi355q_mul_mat_id_workspace(w.type, w.M, w.K, n_used, n_tok, x_ne1, w.n_expert)), x.device)
    _check(lib().mi355q_mul_mat_id(w.type, w.data.data_ptr(), w.row_bytes, w.row_bytes * w.M, w.n_expert,
                                   x.data_ptr(), x_ne1, x.stride(1) * 4, x.stride(0) * 4,
                                   ids.data_ptr(), ids.stride(0) * 4, y.data_ptr(), w.M, w.K, n_used, n_tok,
                                   ws.data_ptr() if ws is not None else None, wsb, flags, _stream(torch)), "mul_mat_id")
    return y
