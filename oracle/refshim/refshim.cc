// oracle/refshim/refshim.cc -- TEST INFRASTRUCTURE (oracle side), not product code.
//
// Flat C entry points onto the REAL reference libraries (libggml-base / libggml-cpu built by
// oracle/Makefile from /root/reference sources into oracle/_ref/<variant>/).  Python tests and the
// bench.py cpu_baseline leg load this through ctypes; nothing in the product path links it.
//
// Every function here just forwards to the public ggml API of the reference:
//   type traits ............ ggml/include/ggml.h:2123-2133, ggml/include/ggml-cpu.h:112-119
//   ggml_quantize_chunk .... ggml/src/ggml.c:6386-6450
//   CPU mul_mat ............ ggml/src/ggml-cpu/ggml-cpu.c:1266-1458 (through ggml_graph_compute_with_ctx)
//   CPU mul_mat_id ......... ggml/src/ggml-cpu/ggml-cpu.c:1540-1718
#include <ggml.h>
#include <ggml-cpu.h>
#include <ggml-backend.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

// ggml fills its f16->f32 lookup table inside the first ggml_init() (ggml/src/ggml.c, ggml_init);
// the scalar (non-F16C) build reads that table from every GGML_FP16_TO_FP32, so force it once.
namespace {
struct ref_static_init {
    ref_static_init() {
        struct ggml_init_params ip = { 1024, nullptr, false };
        struct ggml_context * ctx = ggml_init(ip);
        if (ctx) ggml_free(ctx);
        ggml_cpu_init();
    }
} g_ref_static_init;
}

extern "C" {

int64_t ref_blck_size(int type) { return ggml_blck_size((ggml_type) type); }
int64_t ref_type_size(int type) { return (int64_t) ggml_type_size((ggml_type) type); }
int64_t ref_row_size(int type, int64_t k) { return (int64_t) ggml_row_size((ggml_type) type, k); }
const char * ref_type_name(int type) { return ggml_type_name((ggml_type) type); }
int ref_vec_dot_type(int type) { ggml_cpu_init(); return (int) ggml_get_type_traits_cpu((ggml_type) type)->vec_dot_type; }
int ref_requires_imatrix(int type) { return ggml_quantize_requires_imatrix((ggml_type) type) ? 1 : 0; }

// f32 rows -> packed blocks of `type` (imatrix = all ones when the type demands one, as
// tests/test-backend-ops.cpp:80-89 does).
int64_t ref_quantize_chunk(int type, const float * src, void * dst, int64_t nrows, int64_t n_per_row) {
    ggml_quantize_init((ggml_type) type);
    std::vector<float> im;
    const float * imatrix = nullptr;
    if (ggml_quantize_requires_imatrix((ggml_type) type)) {
        im.assign(n_per_row, 1.0f);
        imatrix = im.data();
    }
    return (int64_t) ggml_quantize_chunk((ggml_type) type, src, dst, 0, nrows, n_per_row, imatrix);
}

// packed blocks -> f32 (the exact decode spec: dequantize_row_*, ggml-quants.c)
void ref_dequantize_row(int type, const void * src, float * dst, int64_t k) {
    ggml_get_type_traits((ggml_type) type)->to_float(src, dst, k);
}

// reference (scalar, deterministic) quantizer quantize_row_*_ref
// (Q8_K has no from_float_ref in the base traits, ggml.c:788-793; its only quantizer is the CPU one,
//  which forwards to quantize_row_q8_K_ref -- ggml-cpu-quants.c:1746)
void ref_from_float_ref(int type, const float * x, void * y, int64_t k) {
    ggml_from_float_t f = ggml_get_type_traits((ggml_type) type)->from_float_ref;
    if (f) f(x, y, k);
    else   ggml_get_type_traits_cpu((ggml_type) type)->from_float(x, y, k);
}

// the CPU backend's own activation quantizer (may be SIMD) for `type` (a vec_dot_type)
void ref_from_float_cpu(int type, const float * x, void * y, int64_t k) {
    ggml_cpu_init();
    ggml_get_type_traits_cpu((ggml_type) type)->from_float(x, y, k);
}

// one row . one quantized activation row, the CPU hot loop (ggml-cpu.c:1176-1264 calls this)
void ref_vec_dot(int type, int64_t n, float * s, const void * vx, const void * vy) {
    ggml_cpu_init();
    ggml_get_type_traits_cpu((ggml_type) type)->vec_dot((int) n, s, 0, vx, 0, vy, 0, 1);
}

// dst[M x N] (f32, row-major N rows of M) = W[M x K](type) . x[N rows of K] (f32)
// through the real CPU backend op.  Returns 0 on success.
int ref_mul_mat(int type, const void * W, const float * x, float * dst,
                int64_t M, int64_t N, int64_t K, int n_threads) {
    const size_t wbytes = ggml_row_size((ggml_type) type, K) * M;
    const size_t mem = wbytes + (size_t) K * N * 4 + (size_t) M * N * 4 + (size_t) 64 * 1024 * 1024
                     + (size_t) K * N * 8; // work buffer head-room (from_float scratch)
    struct ggml_init_params ip = { mem, nullptr, false };
    struct ggml_context * ctx = ggml_init(ip);
    if (!ctx) return 1;
    struct ggml_tensor * a = ggml_new_tensor_2d(ctx, (ggml_type) type, K, M);
    struct ggml_tensor * b = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, K, N);
    memcpy(a->data, W, wbytes);
    memcpy(b->data, x, (size_t) K * N * 4);
    struct ggml_tensor * c = ggml_mul_mat(ctx, a, b);
    struct ggml_cgraph * gf = ggml_new_graph(ctx);
    ggml_build_forward_expand(gf, c);
    enum ggml_status st = ggml_graph_compute_with_ctx(ctx, gf, n_threads);
    if (st == GGML_STATUS_SUCCESS) memcpy(dst, c->data, (size_t) M * N * 4);
    ggml_free(ctx);
    return st == GGML_STATUS_SUCCESS ? 0 : 2;
}

// dst[M, n_used, n_tok] = as[M x K x n_expert][ids] . b[K, nb1 (n_used or 1), n_tok]
// ids i32 [n_used, n_tok].  (ggml.c:2771-2796 / ggml-cpu.c:1540-1718)
int ref_mul_mat_id(int type, const void * as, const float * b, const int32_t * ids, float * dst,
                   int64_t M, int64_t K, int64_t n_expert, int64_t n_used, int64_t n_tok, int64_t b_ne1,
                   int n_threads) {
    const size_t wbytes = ggml_row_size((ggml_type) type, K) * M * n_expert;
    const size_t mem = wbytes + (size_t) K * b_ne1 * n_tok * 12 + (size_t) M * n_used * n_tok * 4
                     + (size_t) 64 * 1024 * 1024;
    struct ggml_init_params ip = { mem, nullptr, false };
    struct ggml_context * ctx = ggml_init(ip);
    if (!ctx) return 1;
    struct ggml_tensor * a  = ggml_new_tensor_3d(ctx, (ggml_type) type, K, M, n_expert);
    struct ggml_tensor * bb = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, K, b_ne1, n_tok);
    struct ggml_tensor * id = ggml_new_tensor_2d(ctx, GGML_TYPE_I32, n_used, n_tok);
    memcpy(a->data, as, wbytes);
    memcpy(bb->data, b, (size_t) K * b_ne1 * n_tok * 4);
    memcpy(id->data, ids, (size_t) n_used * n_tok * 4);
    struct ggml_tensor * c = ggml_mul_mat_id(ctx, a, bb, id);
    struct ggml_cgraph * gf = ggml_new_graph(ctx);
    ggml_build_forward_expand(gf, c);
    enum ggml_status st = ggml_graph_compute_with_ctx(ctx, gf, n_threads);
    if (st == GGML_STATUS_SUCCESS) memcpy(dst, c->data, (size_t) M * n_used * n_tok * 4);
    ggml_free(ctx);
    return st == GGML_STATUS_SUCCESS ? 0 : 2;
}

// ---------------------------------------------------------------------------------------------
// CPU baseline: a chain of independent quantized mul_mats (the per-token weight set of a model),
// built ONCE as one ggml graph on the CPU backend, then computed `iters` times.  Returns seconds
// per graph evaluation (mean), or <0 on error.  Weights are filled with a cheap LCG byte pattern
// whose f16 scale fields are forced finite -- values do not matter for timing.
// ---------------------------------------------------------------------------------------------
static void fill_blocks(void * p, size_t nbytes, uint32_t seed) {
    uint8_t * b = (uint8_t *) p;
    uint32_t s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < nbytes; ++i) { s = s * 1664525u + 1013904223u; b[i] = (uint8_t)(s >> 24) & 0x3F; }
}

double ref_bench_chain(int n_mats, const int * types, const int64_t * Ms, const int64_t * Ks,
                       int64_t N, int n_threads, int warmup, int iters) {
    size_t mem = (size_t) 256 * 1024 * 1024;
    for (int i = 0; i < n_mats; ++i) {
        mem += ggml_row_size((ggml_type) types[i], Ks[i]) * Ms[i] + (size_t) Ks[i] * N * 4
             + (size_t) Ms[i] * N * 4 + 4096;
    }
    struct ggml_init_params ip = { mem, nullptr, false };
    struct ggml_context * ctx = ggml_init(ip);
    if (!ctx) return -1.0;
    struct ggml_cgraph * gf = ggml_new_graph_custom(ctx, 4 * (size_t) n_mats + 64, false);
    for (int i = 0; i < n_mats; ++i) {
        struct ggml_tensor * a = ggml_new_tensor_2d(ctx, (ggml_type) types[i], Ks[i], Ms[i]);
        struct ggml_tensor * b = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, Ks[i], N);
        fill_blocks(a->data, ggml_nbytes(a), (uint32_t) i + 1);
        float * bf = (float *) b->data;
        for (int64_t j = 0; j < Ks[i] * N; ++j) bf[j] = (float)((j * 2654435761u >> 8) & 0xFFFF) / 65536.0f - 0.5f;
        ggml_build_forward_expand(gf, ggml_mul_mat(ctx, a, b));
    }
    struct ggml_cplan plan = ggml_graph_plan(gf, n_threads, nullptr);
    std::vector<uint8_t> work(plan.work_size + 64);
    plan.work_data = work.data();
    for (int w = 0; w < warmup; ++w) {
        if (ggml_graph_compute(gf, &plan) != GGML_STATUS_SUCCESS) { ggml_free(ctx); return -2.0; }
    }
    auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < iters; ++it) {
        if (ggml_graph_compute(gf, &plan) != GGML_STATUS_SUCCESS) { ggml_free(ctx); return -2.0; }
    }
    auto t1 = std::chrono::steady_clock::now();
    ggml_free(ctx);
    return std::chrono::duration<double>(t1 - t0).count() / (iters > 0 ? iters : 1);
}


// ---- residency ops on the reference CPU backend: one op on contiguous f32 tensors -----------------
// op: 1 ADD, 2 SUB, 3 MUL, 4 DIV (b broadcast over a), 10 RMS_NORM (fp[0] = eps), 11 SILU, 12 SOFT_MAX (b = f32 mask or NULL,
// fp[0] = scale, fp[1] = max_bias), 13 ROPE (ip = {n_dims, mode, n_ctx_orig}, fp = {freq_base, freq_scale, ext_factor,
// attn_factor, beta_fast, beta_slow}, pos = i32 [ne_a[2]], b = freq factors [n_dims/2] or NULL), 14 MUL_MAT with an f16 src0
// (a is given as f32 and rounded to f16 here), 15 MUL_MAT f32.  ne_*: ggml order.  Returns 0 on success.
int ref_glue_op(int op, const float * a, const int64_t * ne_a, const float * b, const int64_t * ne_b, const int32_t * pos,
                const float * fp, const int32_t * ip, float * out, int n_threads) {
    const int64_t na = ne_a[0] * ne_a[1] * ne_a[2] * ne_a[3];
    const int64_t nbn = b ? ne_b[0] * ne_b[1] * ne_b[2] * ne_b[3] : 0;
    struct ggml_init_params ipar = { (size_t) (na + nbn) * 16 + (size_t) 256 * 1024 * 1024, nullptr, false };
    struct ggml_context * ctx = ggml_init(ipar);
    if (!ctx) return 1;
    struct ggml_tensor * ta = ggml_new_tensor_4d(ctx, op == 14 ? GGML_TYPE_F16 : GGML_TYPE_F32, ne_a[0], ne_a[1], ne_a[2], ne_a[3]);
    if (op == 14) ggml_fp32_to_fp16_row(a, (ggml_fp16_t *) ta->data, na); else memcpy(ta->data, a, (size_t) na * 4);
    struct ggml_tensor * tb = nullptr;
    if (b) { tb = ggml_new_tensor_4d(ctx, GGML_TYPE_F32, ne_b[0], ne_b[1], ne_b[2], ne_b[3]); memcpy(tb->data, b, (size_t) nbn * 4); }
    struct ggml_tensor * c = nullptr;
    switch (op) {
    case 1: c = ggml_add(ctx, ta, tb); break;
    case 2: c = ggml_sub(ctx, ta, tb); break;
    case 3: c = ggml_mul(ctx, ta, tb); break;
    case 4: c = ggml_div(ctx, ta, tb); break;
    case 10: c = ggml_rms_norm(ctx, ta, fp[0]); break;
    case 11: c = ggml_silu(ctx, ta); break;
    case 12: c = ggml_soft_max_ext(ctx, ta, tb, fp[0], fp[1]); break;
    case 13: {
        struct ggml_tensor * tp = ggml_new_tensor_1d(ctx, GGML_TYPE_I32, ne_a[2]);
        memcpy(tp->data, pos, (size_t) ne_a[2] * 4);
        c = ggml_rope_ext(ctx, ta, tp, tb, ip[0], ip[1], ip[2], fp[0], fp[1], fp[2], fp[3], fp[4], fp[5]);
    } break;
    case 14: case 15: c = ggml_mul_mat(ctx, ta, tb); break;
    default: ggml_free(ctx); return 3;
    }
    struct ggml_cgraph * gf = ggml_new_graph(ctx);
    ggml_build_forward_expand(gf, c);
    enum ggml_status st = ggml_graph_compute_with_ctx(ctx, gf, n_threads);
    if (st == GGML_STATUS_SUCCESS) memcpy(out, c->data, ggml_nbytes(c));
    ggml_free(ctx);
    return st == GGML_STATUS_SUCCESS ? 0 : 2;
}

// FLASH_ATTN_EXT on the reference CPU backend (ggml-cpu/ops.cpp:6686-6905): q f32 [DK, N, H, B]; k / v given as f32 and stored in the cache type kv_type
// (F16: rounded; quantized types: ggml_quantize_chunk, i.e. the reference quantizer) as [DK|DV, n_kv, Hk, B]; mask given as f32 [n_kv, n_pad] and stored
// as F16, or null.  out: f32 [DV, H, N, B].  k_bytes / v_bytes (optional): the cache contents as stored, for the caller to hand to another implementation.
int ref_flash_attn_ext_t(const float * q, const int64_t * ne_q, const float * k, const int64_t * ne_k, const float * v, const int64_t * ne_v,
                         const float * mask, int64_t n_pad, float scale, float max_bias, float softcap, int kv_type, float * out, void * k_bytes, void * v_bytes, int n_threads) {
    const int64_t nq = ne_q[0] * ne_q[1] * ne_q[2] * ne_q[3], nk = ne_k[0] * ne_k[1] * ne_k[2] * ne_k[3], nv = ne_v[0] * ne_v[1] * ne_v[2] * ne_v[3];
    struct ggml_init_params ipar = { (size_t) (nq + nk + nv + ne_k[1] * n_pad) * 8 + (size_t) 256 * 1024 * 1024, nullptr, false };
    struct ggml_context * ctx = ggml_init(ipar);
    if (!ctx) return 1;
    const ggml_type kvt = (ggml_type) kv_type;
    struct ggml_tensor * tq = ggml_new_tensor_4d(ctx, GGML_TYPE_F32, ne_q[0], ne_q[1], ne_q[2], ne_q[3]);
    struct ggml_tensor * tk = ggml_new_tensor_4d(ctx, kvt, ne_k[0], ne_k[1], ne_k[2], ne_k[3]);
    struct ggml_tensor * tv = ggml_new_tensor_4d(ctx, kvt, ne_v[0], ne_v[1], ne_v[2], ne_v[3]);
    memcpy(tq->data, q, (size_t) nq * 4);
    if (kvt == GGML_TYPE_F16) {
        ggml_fp32_to_fp16_row(k, (ggml_fp16_t *) tk->data, nk);
        ggml_fp32_to_fp16_row(v, (ggml_fp16_t *) tv->data, nv);
    } else {
        ggml_quantize_chunk(kvt, k, tk->data, 0, nk / ne_k[0], ne_k[0], nullptr);
        ggml_quantize_chunk(kvt, v, tv->data, 0, nv / ne_v[0], ne_v[0], nullptr);
    }
    if (k_bytes) memcpy(k_bytes, tk->data, ggml_nbytes(tk));
    if (v_bytes) memcpy(v_bytes, tv->data, ggml_nbytes(tv));
    struct ggml_tensor * tm = nullptr;
    if (mask) {
        tm = ggml_new_tensor_2d(ctx, GGML_TYPE_F16, ne_k[1], n_pad);
        ggml_fp32_to_fp16_row(mask, (ggml_fp16_t *) tm->data, ne_k[1] * n_pad);
    }
    struct ggml_tensor * c = ggml_flash_attn_ext(ctx, tq, tk, tv, tm, scale, max_bias, softcap);
    struct ggml_cgraph * gf = ggml_new_graph(ctx);
    ggml_build_forward_expand(gf, c);
    enum ggml_status st = ggml_graph_compute_with_ctx(ctx, gf, n_threads);
    if (st == GGML_STATUS_SUCCESS) memcpy(out, c->data, ggml_nbytes(c));
    ggml_free(ctx);
    return st == GGML_STATUS_SUCCESS ? 0 : 2;
}
int ref_flash_attn_ext(const float * q, const int64_t * ne_q, const float * k, const int64_t * ne_k, const float * v, const int64_t * ne_v,
                       const float * mask, int64_t n_pad, float scale, float max_bias, float softcap, float * out, int n_threads) {
    return ref_flash_attn_ext_t(q, ne_q, k, ne_k, v, ne_v, mask, n_pad, scale, max_bias, softcap, (int) GGML_TYPE_F16, out, nullptr, nullptr, n_threads);
}

} // extern "C"
