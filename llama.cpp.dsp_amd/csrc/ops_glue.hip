// ops_glue.hip -- the small f32/f16 graph ops that sit between the quantized matmuls of a decode graph (SURVEY.md 8f-1:
// the "residency" set).  Without them every layer is ~10 scheduler splits with PCIe round trips; with them a layer's
// activations never leave HBM.  None of these is a roofline path at decode sizes (KBs per op): they are written for
// generality (ggml's strides, broadcasting and views) and exactness against the reference CPU ops, not for speed --
// the fused forms belong in the decode plan.
//
//   ADD / SUB / MUL / DIV   ggml_compute_forward_{add,sub,mul,div}_f32 (binary_op), ggml-cpu/binary-ops.cpp  [broadcast of src1]
//   UNARY(SILU, ...)        ggml-cpu/unary-ops.cpp, vec.h:ggml_silu_f32 = x / (1 + expf(-x))
//   RMS_NORM                ggml-cpu/ops.cpp:3180-3226 (sum of squares accumulated in double, scale = 1/sqrtf(mean + eps))
//   CPY / CONT / DUP        ggml-cpu/ops.cpp ggml_compute_forward_dup (logical element order), f32 <-> f16
//   SOFT_MAX                ggml-cpu/ops.cpp:4641-4736 (scale, mask f32/f16 broadcast over rows, ALiBi slope, max-subtracted exp)
//   ROPE                    ggml-cpu/ops.cpp:4990-5270 (normal / neox, YaRN, frequency factors)
//   GET_ROWS, SCALE         ggml-cpu/ops.cpp:4272-4311, 3840-3878
//   MUL_MAT f16/f32 x f32   ggml-cpu.c:1266-1458 with vec_dot_type F16 / F32 (attention KQ, KQV)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "mi355q_common.h"
#include "act_quant.cuh"

namespace mi355q {

struct TensorD { char * data; int type; int64_t ne[4]; int64_t nb[4]; };     // device copy of mi355q_tensor (type: 0 f32, 1 f16)

static TensorD to_d(const mi355q_tensor * t) {
    TensorD d; d.data = (char *) t->data; d.type = t->type;
    for (int i = 0; i < 4; ++i) { d.ne[i] = t->ne[i]; d.nb[i] = t->nb[i]; }
    return d;
}
static int64_t nelements(const mi355q_tensor * t) { return t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3]; }
static bool same_shape(const mi355q_tensor * a, const mi355q_tensor * b) {
    return a->ne[0] == b->ne[0] && a->ne[1] == b->ne[1] && a->ne[2] == b->ne[2] && a->ne[3] == b->ne[3];
}

__device__ __forceinline__ float ld_elem(const char * p, int type) {
    return type == 0 ? *(const float *) p : __half2float(*(const __half *) p);
}
__device__ __forceinline__ void st_elem(char * p, int type, float v) {
    if (type == 0) *(float *) p = v; else *(__half *) p = __float2half_rn(v);
}

// ---- binary ops with ggml broadcasting: dst[i] = a[i] (op) b[i % ne_b] ------------------------------
enum { BIN_ADD = MI355Q_OP_ADD, BIN_SUB = MI355Q_OP_SUB, BIN_MUL = MI355Q_OP_MUL, BIN_DIV = MI355Q_OP_DIV };

template <int OP>
__global__ void __launch_bounds__(256) k_bin_bcast(const TensorD a, const TensorD b, const TensorD d, int64_t n) {
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t i0 = i % d.ne[0], r = i / d.ne[0], i1 = r % d.ne[1], r2 = r / d.ne[1], i2 = r2 % d.ne[2], i3 = r2 / d.ne[2];
        const float x = ld_elem(a.data + i0 * a.nb[0] + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3], a.type);
        const float y = ld_elem(b.data + (i0 % b.ne[0]) * b.nb[0] + (i1 % b.ne[1]) * b.nb[1] + (i2 % b.ne[2]) * b.nb[2] + (i3 % b.ne[3]) * b.nb[3], b.type);
        float v;
        if constexpr (OP == BIN_ADD) v = __fadd_rn(x, y);
        else if constexpr (OP == BIN_SUB) v = __fsub_rn(x, y);
        else if constexpr (OP == BIN_MUL) v = __fmul_rn(x, y);
        else v = __fdiv_rn(x, y);
        st_elem(d.data + i0 * d.nb[0] + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3], d.type, v);
    }
}

// ---- unary ops -------------------------------------------------------------------------------------
template <int OP>
__global__ void __launch_bounds__(256) k_unary(const TensorD a, const TensorD d, int64_t n) {
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t i0 = i % d.ne[0], r = i / d.ne[0], i1 = r % d.ne[1], r2 = r / d.ne[1], i2 = r2 % d.ne[2], i3 = r2 / d.ne[2];
        const float x = ld_elem(a.data + i0 * a.nb[0] + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3], a.type);
        float v;
        if constexpr (OP == MI355Q_UNARY_SILU)         v = __fdiv_rn(x, 1.0f + expf(-x));
        else if constexpr (OP == MI355Q_UNARY_RELU)    v = x > 0.0f ? x : 0.0f;
        else if constexpr (OP == MI355Q_UNARY_SIGMOID) v = __fdiv_rn(1.0f, 1.0f + expf(-x));
        else if constexpr (OP == MI355Q_UNARY_TANH)    v = tanhf(x);
        else if constexpr (OP == MI355Q_UNARY_NEG)     v = -x;
        else                                           v = fabsf(x);
        st_elem(d.data + i0 * d.nb[0] + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3], d.type, v);
    }
}

// ---- reductions over a wave in double (the CPU accumulates in double too) -----------------------------
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- RMS_NORM, optionally fused with a preceding ADD and a following MUL by a weight vector -------------------
// One 256-thread workgroup per row (a decode step has ONE row of n_embd elements: a single wave walking it exposes a memory round
// trip per 64 elements).  The f32 operations and their order are those of the separate ADD, RMS_NORM and MUL nodes, so the fused
// forms are bit-identical to issuing the nodes one by one; the sum of squares is accumulated in f64 as the CPU's ggml_float does
// (ops.cpp ggml_compute_forward_rms_norm_f32).  b / sum / w may be absent.
__global__ void __launch_bounds__(256) k_add_rms_norm_mul(const TensorD a, const TensorD b, int has_b, const TensorD sum, int has_sum,
                                                          const float * __restrict__ w, const TensorD d, float eps) {
    __shared__ double part[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    const int64_t i1 = row % a.ne[1], r2 = row / a.ne[1], i2 = r2 % a.ne[2], i3 = r2 / a.ne[2];
    const float * x = (const float *) (a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
    const float * xb = has_b ? (const float *) (b.data + i1 * b.nb[1] + i2 * b.nb[2] + i3 * b.nb[3]) : nullptr;
    float * xs = has_sum ? (float *) (sum.data + i1 * sum.nb[1] + i2 * sum.nb[2] + i3 * sum.nb[3]) : nullptr;
    float * y = (float *) (d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
    const int64_t n = a.ne[0];
    double s = 0.0;
#pragma unroll 4
    for (int64_t i = tid; i < n; i += 256) {
        const float v = has_b ? __fadd_rn(x[i], xb[i]) : x[i];
        if (has_sum) xs[i] = v;
        s += (double) __fmul_rn(v, v);                            // (ggml_float)(x*x): the square is rounded to f32 first
    }
    s = wave_sum_f64(s);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    s = (part[0] + part[1]) + (part[2] + part[3]);
    const float mean  = (float) (s / (double) n);
    // 1.0f / sqrtf(mean + eps) with BOTH roundings of the CPU: the compiler folds the f32 form into v_rsq_f32 (1 ulp off in ~20 % of
    // rows) whatever intrinsics spell it, so each step is done in f64 and rounded to f32
    const float root  = (float) sqrt((double) __fadd_rn(mean, eps));
    const float scale = (float) (1.0 / (double) root);
    // second pass: with a stored sum it is re-read from there (each thread reads back its own stores), so a and b are not touched after
    // the barrier -- dst may then alias a or b exactly (same base and strides: row r of dst is row r of the operand, owned by this workgroup)
#pragma unroll 4
    for (int64_t i = tid; i < n; i += 256) {
        const float v = has_sum ? xs[i] : (has_b ? __fadd_rn(x[i], xb[i]) : x[i]);
        const float t = __fmul_rn(v, scale);
        y[i] = w ? __fmul_rn(t, w[i]) : t;
    }
}

// The same kernel for rows of at most 1024*NV4 elements with 16-byte aligned operands: every thread keeps its NV4 float4 pieces of the
// row in registers (one read of the operands, 16-byte accesses).  Same f32 operations; only the order of the f64 partial sums differs.
template <int NV4>
__global__ void __launch_bounds__(256) k_add_rms_norm_mul_vec(const TensorD a, const TensorD b, int has_b, const TensorD sum, int has_sum,
                                                              const float * __restrict__ w, const TensorD d, float eps) {
    __shared__ double part[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    const int64_t i1 = row % a.ne[1], r2 = row / a.ne[1], i2 = r2 % a.ne[2], i3 = r2 / a.ne[2];
    const float4 * x = (const float4 *) (a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
    const float4 * xb = has_b ? (const float4 *) (b.data + i1 * b.nb[1] + i2 * b.nb[2] + i3 * b.nb[3]) : nullptr;
    float4 * xs = has_sum ? (float4 *) (sum.data + i1 * sum.nb[1] + i2 * sum.nb[2] + i3 * sum.nb[3]) : nullptr;
    float4 * y = (float4 *) (d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
    const int64_t n4 = a.ne[0] / 4;
    float4 v[NV4];
#pragma unroll
    for (int u = 0; u < NV4; ++u) {
        const int64_t i = tid + 256 * u;
        v[u] = i < n4 ? x[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (has_b) {
#pragma unroll
        for (int u = 0; u < NV4; ++u) {
            const int64_t i = tid + 256 * u;
            if (i < n4) { const float4 t = xb[i]; v[u].x = __fadd_rn(v[u].x, t.x); v[u].y = __fadd_rn(v[u].y, t.y); v[u].z = __fadd_rn(v[u].z, t.z); v[u].w = __fadd_rn(v[u].w, t.w); }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < NV4; ++u) {
        const int64_t i = tid + 256 * u;
        if (i < n4) {
            if (has_sum) xs[i] = v[u];
            s += (double) __fmul_rn(v[u].x, v[u].x); s += (double) __fmul_rn(v[u].y, v[u].y);
            s += (double) __fmul_rn(v[u].z, v[u].z); s += (double) __fmul_rn(v[u].w, v[u].w);
        }
    }
    s = wave_sum_f64(s);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    s = (part[0] + part[1]) + (part[2] + part[3]);
    const float mean  = (float) (s / (double) a.ne[0]);
    const float root  = (float) sqrt((double) __fadd_rn(mean, eps));      // both roundings of the CPU (see k_add_rms_norm_mul)
    const float scale = (float) (1.0 / (double) root);
#pragma unroll
    for (int u = 0; u < NV4; ++u) {
        const int64_t i = tid + 256 * u;
        if (i < n4) {
            float4 t = make_float4(__fmul_rn(v[u].x, scale), __fmul_rn(v[u].y, scale), __fmul_rn(v[u].z, scale), __fmul_rn(v[u].w, scale));
            if (w) { const float4 ww = ((const float4 *) w)[i]; t.x = __fmul_rn(t.x, ww.x); t.y = __fmul_rn(t.y, ww.y); t.z = __fmul_rn(t.z, ww.z); t.w = __fmul_rn(t.w, ww.w); }
            y[i] = t;
        }
    }
}

// ---- fused UNARY * other (SiLU(gate) * up): contiguous f32 tensors of one shape ---------------------------------
template <int OP>
__global__ void __launch_bounds__(256) k_unary_mul(const float * __restrict__ a, const float * __restrict__ b, float * __restrict__ d, int64_t n) {
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x) {
        const float x = a[i];
        float v;
        if constexpr (OP == MI355Q_UNARY_SILU)         v = __fdiv_rn(x, 1.0f + expf(-x));
        else if constexpr (OP == MI355Q_UNARY_RELU)    v = x > 0.0f ? x : 0.0f;
        else                                           v = __fdiv_rn(1.0f, 1.0f + expf(-x));
        d[i] = __fmul_rn(v, b[i]);
    }
}

// ---- CPY / CONT / DUP: logical element order, any strides, f32 <-> f16 -------------------------------------
__global__ void __launch_bounds__(256) k_cpy(const TensorD a, TensorD d, int64_t n, char * const * dest_table, int dest_index) {
    if (dest_table) d.data = dest_table[dest_index];           // destination resolved on the device (graph replay with a moving KV position)
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t a0 = i % a.ne[0], ar = i / a.ne[0], a1 = ar % a.ne[1], ar2 = ar / a.ne[1], a2 = ar2 % a.ne[2], a3 = ar2 / a.ne[2];
        const int64_t d0 = i % d.ne[0], dr = i / d.ne[0], d1 = dr % d.ne[1], dr2 = dr / d.ne[1], d2 = dr2 % d.ne[2], d3 = dr2 / d.ne[2];
        const char * src = a.data + a0 * a.nb[0] + a1 * a.nb[1] + a2 * a.nb[2] + a3 * a.nb[3];
        char * dst = d.data + d0 * d.nb[0] + d1 * d.nb[1] + d2 * d.nb[2] + d3 * d.nb[3];
        if (a.type == d.type) { if (a.type == 0) *(float *) dst = *(const float *) src; else *(uint16_t *) dst = *(const uint16_t *) src; }
        else st_elem(dst, d.type, ld_elem(src, a.type));
    }
}

// Batch-sized copies (the head merge after attention, the K / V stores of a prompt) take one of two forms of the same element-wise copy:
//  * rows: identical shapes, unit stride along dim 0 on both sides -- 4 elements per thread, 16-byte loads / stores where the addresses allow;
//  * transpose: a 2-D source that is contiguous along its dim 1 into a destination that is contiguous along its dim 0 (the transposed V cache
//    store): 64 x 64 tiles through LDS so that both the reads and the writes are whole lines.
// The values are the same conversions as k_cpy's (f32 -> f16 round to nearest even); only the access pattern differs.
__global__ void __launch_bounds__(256) k_cpy_rows4(const TensorD a, TensorD d, int64_t n4, char * const * dest_table, int dest_index) {
    if (dest_table) d.data = dest_table[dest_index];
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t) gridDim.x * blockDim.x) {
        // logical element e .. e+3: inside one row of the source AND of the destination (both row lengths are multiples of 4; the shapes may differ,
        // e.g. [head_dim, n_head, n_tokens] -> [n_embd, n_tokens])
        const int64_t e = 4 * i;
        const int64_t a0 = e % a.ne[0], ar = e / a.ne[0], a1 = ar % a.ne[1], ar2 = ar / a.ne[1], a2 = ar2 % a.ne[2], a3 = ar2 / a.ne[2];
        const int64_t d0 = e % d.ne[0], dr = e / d.ne[0], d1 = dr % d.ne[1], dr2 = dr / d.ne[1], d2 = dr2 % d.ne[2], d3 = dr2 / d.ne[2];
        const char * src = a.data + a0 * a.nb[0] + a1 * a.nb[1] + a2 * a.nb[2] + a3 * a.nb[3];
        char * dst = d.data + d0 * d.nb[0] + d1 * d.nb[1] + d2 * d.nb[2] + d3 * d.nb[3];
        float v[4];
        if (a.type == 0) {
            if (((uintptr_t) src & 15) == 0) { const float4 t = *(const float4 *) src; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
            else { for (int j = 0; j < 4; ++j) v[j] = ((const float *) src)[j]; }
        } else {
            for (int j = 0; j < 4; ++j) v[j] = __half2float(((const __half *) src)[j]);
        }
        if (d.type == 0) {
            if (((uintptr_t) dst & 15) == 0) *(float4 *) dst = make_float4(v[0], v[1], v[2], v[3]);
            else { for (int j = 0; j < 4; ++j) ((float *) dst)[j] = v[j]; }
        } else {
            __half h[4];
            if (a.type == 1) { for (int j = 0; j < 4; ++j) h[j] = ((const __half *) src)[j]; }
            else             { for (int j = 0; j < 4; ++j) h[j] = __float2half_rn(v[j]); }
            if (((uintptr_t) dst & 7) == 0) *(uint2 *) dst = *(const uint2 *) h;
            else { for (int j = 0; j < 4; ++j) ((__half *) dst)[j] = h[j]; }
        }
    }
}
__global__ void __launch_bounds__(256) k_cpy_transpose(const TensorD a, TensorD d, char * const * dest_table, int dest_index) {
    if (dest_table) d.data = dest_table[dest_index];
    __shared__ float tile[32][33];                              // 32 x 32 tiles: a 512 x 1024 V store is 512 workgroups, not 128
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t b0 = (int64_t) blockIdx.x * 32, b1 = (int64_t) blockIdx.y * 32, i2 = blockIdx.z % a.ne[2], i3 = blockIdx.z / a.ne[2];
    const char * ab = a.data + i2 * a.nb[2] + i3 * a.nb[3];
    char * db = d.data + i2 * d.nb[2] + i3 * d.nb[3];
    for (int j = ty; j < 32; j += 8) {                           // read: lanes run along the source's contiguous dim 1
        const int64_t i0 = b0 + j, i1 = b1 + tx;
        if (i0 < a.ne[0] && i1 < a.ne[1]) tile[j][tx] = ld_elem(ab + i0 * a.nb[0] + i1 * a.nb[1], a.type);
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {                           // write: lanes run along the destination's contiguous dim 0
        const int64_t i0 = b0 + tx, i1 = b1 + j;
        if (i0 < a.ne[0] && i1 < a.ne[1]) st_elem(db + i0 * d.nb[0] + i1 * d.nb[1], d.type, tile[tx][j]);
    }
}
static int grid_for(int64_t n);
static bool cpy_same_shape(const mi355q_tensor * a, const mi355q_tensor * d) { return a->ne[0] == d->ne[0] && a->ne[1] == d->ne[1] && a->ne[2] == d->ne[2] && a->ne[3] == d->ne[3]; }
// picks the form; returns false when the generic element-wise kernel has to run
// ---- CPY f32 -> Q8_0 (the K / V stores of a quantized cache, -ctk q8_0 / -ctv q8_0; the reference: ggml-cuda/cpy.cu:71,416-423, CPU: dup -> from_float) ----
// The CPU quantizes source row by source row into a contiguous destination (ggml_compute_forward_dup_f32, quantized dst): logical block b of 32
// elements -> bytes [34 b, 34 b + 34) of the destination.  Eight lanes own one block (q80_group8: quantize_row_q8_0_ref's arithmetic, round half
// away from zero -- the CPU's SIMD quantizer rounds half to even, the same one-in-a-million tie as in the matmul path's activation quantizer).
__global__ void __launch_bounds__(256) k_cpy_f32_q8_0(const TensorD a, char * dst, int64_t n_blocks, char * const * dest_table, int dest_index) {
    if (dest_table) dst = dest_table[dest_index];
    const int lane8 = threadIdx.x & 7;
    for (int64_t b0 = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 3; b0 < ((n_blocks + 31) & ~(int64_t) 31); b0 += ((int64_t) gridDim.x * blockDim.x) >> 3) {
        const int64_t b = b0 < n_blocks ? b0 : n_blocks - 1;    // (every lane of a wave takes part in the group reductions; surplus groups redo the last block)
        const int64_t e = 32 * b + 4 * lane8;
        const int64_t a0 = e % a.ne[0], ar = e / a.ne[0], a1 = ar % a.ne[1], ar2 = ar / a.ne[1], a2 = ar2 % a.ne[2], a3 = ar2 / a.ne[2];
        const float * src = (const float *) (a.data + a0 * 4 + a1 * a.nb[1] + a2 * a.nb[2] + a3 * a.nb[3]);
        const float4 v = make_float4(src[0], src[1], src[2], src[3]);
        uint32_t q; float d; int sum;
        q80_group8<false>(v, q, d, sum);
        if (b0 < n_blocks) {
            char * blk = dst + 34 * b;                           // 2-byte aligned: the quants are stored as halves of a word
            if (lane8 == 0) *(__half *) blk = __float2half_rn(d);
            *(uint16_t *) (blk + 2 + 4 * lane8) = (uint16_t) (q & 0xFFFFu);
            *(uint16_t *) (blk + 4 + 4 * lane8) = (uint16_t) (q >> 16);
        }
    }
}

// CPY f32 -> Q4_0 (a Q4_0 cache): quantize_row_q4_0_ref (ggml-quants.c:31-66) -- max = the FIRST element of largest magnitude, signed; d = max / -8;
// q = min(15, (int8) (x / d + 8.5)); byte j = q[j] | q[j + 16] << 4.  Eight lanes own a block (lane l: elements 4l .. 4l+3).
__global__ void __launch_bounds__(256) k_cpy_f32_q4_0(const TensorD a, char * dst, int64_t n_blocks, char * const * dest_table, int dest_index) {
    if (dest_table) dst = dest_table[dest_index];
    const int lane = threadIdx.x & 63, lane8 = lane & 7;
    for (int64_t b0 = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 3; b0 < ((n_blocks + 31) & ~(int64_t) 31); b0 += ((int64_t) gridDim.x * blockDim.x) >> 3) {
        const int64_t b = b0 < n_blocks ? b0 : n_blocks - 1;
        const int64_t e = 32 * b + 4 * lane8;
        const int64_t a0 = e % a.ne[0], ar = e / a.ne[0], a1 = ar % a.ne[1], ar2 = ar / a.ne[1], a2 = ar2 % a.ne[2], a3 = ar2 / a.ne[2];
        const float * src = (const float *) (a.data + a0 * 4 + a1 * a.nb[1] + a2 * a.nb[2] + a3 * a.nb[3]);
        const float x[4] = { src[0], src[1], src[2], src[3] };
        float amax = 0.0f, mx = 0.0f;                            // this lane's first element of largest magnitude
#pragma unroll
        for (int i = 0; i < 4; ++i) if (amax < fabsf(x[i])) { amax = fabsf(x[i]); mx = x[i]; }
        const float gmax = oct_max(amax);                         // the block's largest magnitude; its first holder is the lowest lane that has it
        const unsigned long long holders = __ballot(amax == gmax);
        const int first = __ffs((int) ((holders >> (lane & ~7)) & 0xFFull)) - 1;
        const float bmax = __shfl(mx, (lane & ~7) + first, 64);
        const float d = __fdiv_rn(bmax, -8.0f), id = d != 0.0f ? __fdiv_rn(1.0f, d) : 0.0f;
        uint32_t nib = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int qi = min(15, (int) (int8_t) __fadd_rn(__fmul_rn(x[i], id), 8.5f)); nib |= (uint32_t) (qi & 0xFF) << (8 * i); }
        const uint32_t hi = (uint32_t) dpp_i<0x104>((int) nib);  // row_shl:4 -- lanes 0..3 of the group get elements 16..31 from lanes 4..7
        if (b0 < n_blocks && lane8 < 4) {
            char * blk = dst + 18 * b;
            const uint32_t w = (nib & 0x0F0F0F0Fu) | ((hi & 0x0F0F0F0Fu) << 4);
            if (lane8 == 0) *(__half *) blk = __float2half_rn(d);
            *(uint16_t *) (blk + 2 + 4 * lane8) = (uint16_t) (w & 0xFFFFu);
            *(uint16_t *) (blk + 4 + 4 * lane8) = (uint16_t) (w >> 16);
        }
    }
}

static bool cpy_fast(const mi355q_tensor * a, const mi355q_tensor * d, char * const * table, int index, hipStream_t st) {
    const int64_t n = nelements(d);
    if (n < 16384) return false;
    const int64_t ea = a->type == 0 ? 4 : 2, ed = d->type == 0 ? 4 : 2;
    if (a->nb[0] == ea && d->nb[0] == ed && a->ne[0] % 4 == 0 && d->ne[0] % 4 == 0) {
        hipLaunchKernelGGL(k_cpy_rows4, dim3(grid_for(n / 4)), dim3(256), 0, st, to_d(a), to_d(d), n / 4, table, index);
        return true;
    }
    if (cpy_same_shape(a, d) && a->nb[1] == ea && d->nb[0] == ed && a->ne[2] * a->ne[3] <= 65535 && (a->ne[1] + 31) / 32 <= 65535) {
        hipLaunchKernelGGL(k_cpy_transpose, dim3((unsigned) ((a->ne[0] + 31) / 32), (unsigned) ((a->ne[1] + 31) / 32), (unsigned) (a->ne[2] * a->ne[3])), dim3(256), 0, st,
                           to_d(a), to_d(d), table, index);
        return true;
    }
    return false;
}

// ---- SOFT_MAX: one wave per row; mask (f32/f16) broadcast over rows i1 % ne01; ALiBi slope per head -------------
__global__ void __launch_bounds__(256) k_soft_max(const TensorD a, const TensorD m, const TensorD d, float scale, float max_bias,
                                                  float m0, float m1, uint32_t n_head_log2, int64_t nrows, int has_mask) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int64_t nc = a.ne[0];
    const float * sp = (const float *) (a.data + row * a.nb[1]);
    float * dp = (float *) (d.data + row * d.nb[1]);
    const uint32_t h = (uint32_t) ((row / a.ne[1]) % a.ne[2]);
    const float slope = max_bias > 0.0f ? (h < n_head_log2 ? powf(m0, (float) (h + 1)) : powf(m1, (float) (2 * (h - n_head_log2) + 1))) : 1.0f;
    const char * mp = has_mask ? m.data + (row % a.ne[1]) * nc * (m.type == 0 ? 4 : 2) : nullptr;
    auto val = [&](int64_t i) {
        float v = __fmul_rn(sp[i], scale);
        if (mp) v = __fadd_rn(v, __fmul_rn(slope, m.type == 0 ? ((const float *) mp)[i] : __half2float(((const __half *) mp)[i])));
        return v;
    };
    float mx = -INFINITY;
#pragma unroll 8
    for (int64_t i = lane; i < nc; i += 64) mx = fmaxf(mx, val(i));
    mx = wave_max_f32(mx);
    double sum = 0.0;
#pragma unroll 8
    for (int64_t i = lane; i < nc; i += 64) {
        const float e = expf(__fsub_rn(val(i), mx));
        dp[i] = e;
        sum += (double) e;
    }
    sum = wave_sum_f64(sum);
    const float inv = (float) (1.0 / sum);
#pragma unroll 8
    for (int64_t i = lane; i < nc; i += 64) dp[i] = __fmul_rn(dp[i], inv);
}

// The same row softmax with the row held in registers (rows of at most 64*NV elements: every decode-sized attention window): one read
// of the scores and the mask instead of three dependent passes through memory.  Same expressions, same lane-strided order of the f64
// sum as k_soft_max: bit-identical to it.
template <int NV>
__global__ void __launch_bounds__(256) k_soft_max_reg(const TensorD a, const TensorD m, const TensorD d, float scale, float max_bias,
                                                      float m0, float m1, uint32_t n_head_log2, int64_t nrows, int has_mask) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int64_t nc = a.ne[0];
    const float * sp = (const float *) (a.data + row * a.nb[1]);
    float * dp = (float *) (d.data + row * d.nb[1]);
    const uint32_t h = (uint32_t) ((row / a.ne[1]) % a.ne[2]);
    const float slope = max_bias > 0.0f ? (h < n_head_log2 ? powf(m0, (float) (h + 1)) : powf(m1, (float) (2 * (h - n_head_log2) + 1))) : 1.0f;
    const char * mp = has_mask ? m.data + (row % a.ne[1]) * nc * (m.type == 0 ? 4 : 2) : nullptr;
    float v[NV];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int64_t i = lane + 64 * u;
        float x = -INFINITY;
        if (i < nc) {
            x = __fmul_rn(sp[i], scale);
            if (mp) x = __fadd_rn(x, __fmul_rn(slope, m.type == 0 ? ((const float *) mp)[i] : __half2float(((const __half *) mp)[i])));
        }
        v[u] = x; mx = fmaxf(mx, x);
    }
    mx = wave_max_f32(mx);
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        if (lane + 64 * u < nc) { const float e = expf(__fsub_rn(v[u], mx)); v[u] = e; sum += (double) e; }
    }
    sum = wave_sum_f64(sum);
    const float inv = (float) (1.0 / sum);
#pragma unroll
    for (int u = 0; u < NV; ++u) if (lane + 64 * u < nc) dp[lane + 64 * u] = __fmul_rn(v[u], inv);
}

// ---- ROPE (normal and neox modes, YaRN scaling, optional frequency factors): one thread per rotated pair ------------
//   ggml-cpu/ops.cpp:4990-5028 (rope_yarn, ggml_rope_cache_init), :5088-5270 (ggml_compute_forward_rope_f32)
struct RopeP { int n_dims, neox; float freq_scale, ext_factor, attn_factor, theta_scale, corr0, corr1; };

__global__ void __launch_bounds__(256) k_rope(const TensorD a, const int32_t * pos, const float * freq_factors, const TensorD d, const RopeP p, int64_t n_pairs) {
    const int64_t half = a.ne[0] / 2;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t ip = i % half, r = i / half, i1 = r % a.ne[1], r2 = r / a.ne[1], i2 = r2 % a.ne[2], i3 = r2 / a.ne[2];
        const char * src = a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3];
        char * dst = d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3];
        const int64_t i0 = 2 * ip;
        if (i0 >= p.n_dims) {                                   // channels beyond n_dims are copied
            ((float *) dst)[i0] = ((const float *) src)[i0]; ((float *) dst)[i0 + 1] = ((const float *) src)[i0 + 1];
            continue;
        }
        const float ff = freq_factors ? freq_factors[ip] : 1.0f;
        // theta is advanced by repeated f32 multiplication exactly as ggml_rope_cache_init does (pos * powf(scale, ip) differs by up
        // to 1e-6 relative after 64 steps -- 1e-3 rad at position 8191)
        float th = (float) pos[i2];
        for (int64_t j = 0; j < ip; ++j) th = __fmul_rn(th, p.theta_scale);
        const float theta_extrap = __fdiv_rn(th, ff);
        const float theta_interp = p.freq_scale * theta_extrap;
        float theta = theta_interp, mscale = p.attn_factor;
        if (p.ext_factor != 0.0f) {
            const float y = ((float) (i0 / 2) - p.corr0) / fmaxf(0.001f, p.corr1 - p.corr0);
            const float ramp_mix = (1.0f - fminf(1.0f, fmaxf(0.0f, y))) * p.ext_factor;
            theta = theta_interp * (1.0f - ramp_mix) + theta_extrap * ramp_mix;
            mscale *= 1.0f + 0.1f * logf(1.0f / p.freq_scale);
        }
        const float c = cosf(theta) * mscale, sn = sinf(theta) * mscale;
        const int64_t e0 = p.neox ? ip : i0, e1 = p.neox ? ip + p.n_dims / 2 : i0 + 1;
        const float x0 = ((const float *) src)[e0], x1 = ((const float *) src)[e1];
        ((float *) dst)[e0] = x0 * c - x1 * sn;
        ((float *) dst)[e1] = x0 * sn + x1 * c;
    }
}

// Batches: one wave per (token, group of HG heads); lane l owns pairs l, l + 64, ... and computes a pair's angle ONCE for all heads of
// its group (the CPU fills a per-position cache for the same reason, ggml_rope_cache_init) -- the per-pair operations are those of k_rope,
// so the results are bit-identical; what changes is 1 / HG of the sincos + theta-chain work and rows read as whole 8-byte-per-lane lines.
__global__ void __launch_bounds__(256) k_rope_rows(const TensorD a, const int32_t * pos, const float * freq_factors, const TensorD d, const RopeP p, int hg, int64_t n_waves) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    const int64_t groups = (a.ne[1] + hg - 1) / hg, g = w % groups, t = w / groups, i2 = t % a.ne[2], i3 = t / a.ne[2];
    const int64_t h0 = g * hg, h1 = h0 + hg < a.ne[1] ? h0 + hg : a.ne[1];
    const int64_t half = a.ne[0] / 2;
    const float fpos = (float) pos[i2];
    for (int64_t ip = lane; ip < half; ip += 64) {
        const int64_t i0 = 2 * ip;
        float c = 1.0f, sn = 0.0f;
        const bool rot = i0 < p.n_dims;
        if (rot) {
            const float ff = freq_factors ? freq_factors[ip] : 1.0f;
            float th = fpos;
            for (int64_t j = 0; j < ip; ++j) th = __fmul_rn(th, p.theta_scale);
            const float theta_extrap = __fdiv_rn(th, ff);
            const float theta_interp = p.freq_scale * theta_extrap;
            float theta = theta_interp, mscale = p.attn_factor;
            if (p.ext_factor != 0.0f) {
                const float y = ((float) (i0 / 2) - p.corr0) / fmaxf(0.001f, p.corr1 - p.corr0);
                const float ramp_mix = (1.0f - fminf(1.0f, fmaxf(0.0f, y))) * p.ext_factor;
                theta = theta_interp * (1.0f - ramp_mix) + theta_extrap * ramp_mix;
                mscale *= 1.0f + 0.1f * logf(1.0f / p.freq_scale);
            }
            c = cosf(theta) * mscale; sn = sinf(theta) * mscale;
        }
        const int64_t e0 = (p.neox && rot) ? ip : i0, e1 = (p.neox && rot) ? ip + p.n_dims / 2 : i0 + 1;
        for (int64_t i1 = h0; i1 < h1; ++i1) {
            const float * src = (const float *) (a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
            float * dst = (float *) (d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
            const float x0 = src[e0], x1 = src[e1];
            if (rot) { dst[e0] = x0 * c - x1 * sn; dst[e1] = x0 * sn + x1 * c; }
            else     { dst[e0] = x0; dst[e1] = x1; }
        }
    }
}

// ---- MUL_MAT with an f16 / f32 src0 (attention KQ and KQV): one wave per output element ---------------------------
//   dst[m, n, i12, i13] = sum_k a[k, m, i12/r2, i13/r3] * b[k, n, i12, i13]   (ggml-cpu.c:1266-1458; an f16 src0 makes the CPU
//   round src1 to f16 first, vec_dot_type = F16 -- reproduced; the products are accumulated in f32)
constexpr int MMF_ROWS = 8;        // src0 rows per wave: their loads are in flight together (one row per wave is a memory round trip per 4 bytes of output)
__global__ void __launch_bounds__(256) k_mul_mat_f(const TensorD a, const TensorD b, const TensorD d, int64_t n_groups, int64_t groups_per_col) {
    const int lane = threadIdx.x & 63;
    const int64_t o = (int64_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (o >= n_groups) return;
    const int64_t mg = o % groups_per_col, r = o / groups_per_col, n = r % d.ne[1], r2 = r / d.ne[1], i12 = r2 % d.ne[2], i13 = r2 / d.ne[2];
    const int64_t i02 = i12 / (b.ne[2] / a.ne[2]), i03 = i13 / (b.ne[3] / a.ne[3]);
    const int64_t m0 = mg * MMF_ROWS, M = d.ne[0];
    const char * abase = a.data + i02 * a.nb[2] + i03 * a.nb[3];
    const float * bp = (const float *) (b.data + n * b.nb[1] + i12 * b.nb[2] + i13 * b.nb[3]);
    const int64_t K = a.ne[0];
    float s[MMF_ROWS];
#pragma unroll
    for (int q = 0; q < MMF_ROWS; ++q) s[q] = 0.0f;
    // per output the same lane-strided products and additions, in the same order, as one row per wave would make
    if (a.type == 1) {
        for (int64_t k = lane; k < K; k += 64) {
            const float bv = __half2float(__float2half_rn(bp[k]));                       // the CPU converts src1 to src0's vec_dot_type (f16) first
#pragma unroll
            for (int q = 0; q < MMF_ROWS; ++q) s[q] += __half2float(((const __half *) (abase + (m0 + q < M ? m0 + q : m0) * a.nb[1]))[k]) * bv;
        }
    } else {
        for (int64_t k = lane; k < K; k += 64) {
            const float bv = bp[k];
#pragma unroll
            for (int q = 0; q < MMF_ROWS; ++q) s[q] += ((const float *) (abase + (m0 + q < M ? m0 + q : m0) * a.nb[1]))[k] * bv;
        }
    }
#pragma unroll
    for (int q = 0; q < MMF_ROWS; ++q) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s[q] += __shfl_xor(s[q], off, 64);
        if (lane == 0 && m0 + q < M) *(float *) (d.data + (m0 + q) * d.nb[0] + n * d.nb[1] + i12 * d.nb[2] + i13 * d.nb[3]) = s[q];
    }
}

// ---- FLASH_ATTN_EXT, f16 K / V, f32 Q: one workgroup per (query row, head, batch) -----------------------------------
// dst[:, h, t, b] = softmax_j(softcap(scale * k_j . q) + slope_h * mask[t][j]) . v_j      (ggml-cpu/ops.cpp:6690-6905; q is rounded to
// f16 before the dot products as the CPU's vec_dot_type conversion does).  Two passes over a row of scores held in LDS instead of the
// CPU's online update (same mathematics; the CPU accumulates V in f16, this kernel in f32).  Written for correctness and decode sizes:
// every workgroup streams its head's K and V once.
constexpr int FA_THREADS = 1024, FA_WAVES = FA_THREADS / 64;      // 16 waves: a decode step has only n_head workgroups, each must hide its own latency
__global__ void __launch_bounds__(FA_THREADS) k_flash_attn_ext(const TensorD q, const TensorD k, const TensorD v, const TensorD m, int has_mask,
                                                        const TensorD d, float scale, float max_bias, float softcap, float m0, float m1, uint32_t n_head_log2,
                                                        int n_split, float * __restrict__ part_out, int seq) {
    extern __shared__ __attribute__((aligned(16))) float fa_s[];              // [n_kv] scores -> probabilities, then [256] partial sums  (seq: + [n_kv] rescale factors)
    __shared__ float red[2 * FA_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // n_split > 1: the KV range is cut in n_split pieces, one workgroup each (a decode step has only n_head rows: more workgroups, shorter
    // chains); each writes its unnormalised sum, maximum and sum of exponentials, k_flash_attn_combine merges them
    const int64_t t = blockIdx.x / n_split, sp = blockIdx.x - t * n_split, h = blockIdx.y, b = blockIdx.z;
    const int64_t DK = k.ne[0], DV = v.ne[0], n_kv_all = k.ne[1];
    const int64_t chunk = (n_kv_all + n_split - 1) / n_split, j_lo = sp * chunk, n_kv = j_lo < n_kv_all ? (j_lo + chunk <= n_kv_all ? chunk : n_kv_all - j_lo) : 0;
    const int64_t hk = h / (q.ne[2] / k.ne[2]), hv = h / (q.ne[2] / v.ne[2]), bk = b / (q.ne[3] / k.ne[3]), bv = b / (q.ne[3] / v.ne[3]);
    const float * qp = (const float *) (q.data + t * q.nb[1] + h * q.nb[2] + b * q.nb[3]);
    const float slope = max_bias > 0.0f ? ((uint32_t) h < n_head_log2 ? powf(m0, (float) (h + 1)) : powf(m1, (float) (2 * (h - n_head_log2) + 1))) : 1.0f;
    const __half * mp = has_mask ? (const __half *) (m.data + t * m.nb[1]) + j_lo : nullptr;
    float qh[4];                                                             // head sizes up to 256: 4 elements per lane
#pragma unroll
    for (int u = 0; u < 4; ++u) qh[u] = lane + 64 * u < DK ? __half2float(__float2half_rn(qp[lane + 64 * u])) : 0.0f;
    // ---- scores
    constexpr int JB = 8;                                                   // rows per trip: their loads are in flight together
    for (int64_t j0 = wave; j0 < n_kv; j0 += FA_WAVES * JB) {
        float kv[JB][4];
#pragma unroll
        for (int c = 0; c < JB; ++c) {
            const int64_t j = j0 + FA_WAVES * c < n_kv ? j0 + FA_WAVES * c : j0;
            const __half * kp = (const __half *) (k.data + (j_lo + j) * k.nb[1] + hk * k.nb[2] + bk * k.nb[3]);
#pragma unroll
            for (int u = 0; u < 4; ++u) kv[c][u] = lane + 64 * u < DK ? __half2float(kp[lane + 64 * u]) : 0.0f;
        }
#pragma unroll
        for (int c = 0; c < JB; ++c) {
            const int64_t j = j0 + FA_WAVES * c;
            float s = 0.0f;
            if (seq) {
                // ggml_vec_dot_f16's order (vec.cpp:128-168, AVX2 form; see plan.hip plan_attn_lds step 2): element e = lane + 64 u feeds accumulator
                // (jj, l) = ((e % 32) / 8, e % 8) of block e / 32 -- lanes L and L ^ 32 own the even and the odd blocks of one accumulator, whose chain
                // therefore passes between the two halves of the wave; then the tree (a0 + a2) + (a1 + a3), s[l] + s[l + 4], (t0 + t1) + (t2 + t3).
                const int nblk = (int) (DK >> 5), half = lane >> 5;
                float acc = 0.0f;
#pragma unroll
                for (int blk = 0; blk < 8; ++blk) {                       // (head sizes up to 256: 8 blocks; constant register indices)
                    if (blk < nblk) {
                        if (half == (blk & 1)) acc = __builtin_fmaf(kv[c][blk >> 1], qh[blk >> 1], acc);
                        const float other = __shfl_xor(acc, 32, 64);
                        if (half != (blk & 1)) acc = other;
                    }
                }
                acc = __fadd_rn(acc, __shfl_xor(acc, 16, 64));            // (a0 + a2) in the j = 0 lanes, (a1 + a3) in the j = 1 lanes
                acc = __fadd_rn(acc, __shfl_xor(acc, 8, 64));             // s[l]
                acc = __fadd_rn(acc, __shfl_xor(acc, 4, 64));             // t[i] = s[i] + s[i + 4]
                acc = __fadd_rn(acc, __shfl_xor(acc, 1, 64));             // t0 + t1 | t2 + t3
                acc = __fadd_rn(acc, __shfl_xor(acc, 2, 64));
                s = acc;                                                  // (lane 0)
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) s += kv[c][u] * qh[u];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            }
            if (lane == 0 && j < n_kv) {
                s = __fmul_rn(s, scale);
                if (softcap != 0.0f) s = softcap * tanhf(s);
                if (mp) {
                    const float mv = __fmul_rn(slope, __half2float(mp[j]));
                    s = (seq && mv == -INFINITY) ? -INFINITY : __fadd_rn(s, mv);     // (seq: a masked position is SKIPPED by the CPU whatever its cache row holds)
                }
                fa_s[j] = s;
            }
        }
    }
    __syncthreads();
    if (seq) {
        // ---- the CPU's online softmax with its F16 accumulator (ggml-cpu/ops.cpp:6810-6890; plan.hip plan_attn_lds steps 3 and 4; oracle/glue.py
        // flash_attn_ext).  Wave 0 walks the scores 64 at a time: prefix maximum -> per position the accumulator's rescale factor ms and the weight vs.
        float * msv = fa_s + n_kv;
        if (wave == 0) {
            const unsigned long long tab = PLAN_EXP2F_T[lane & 31];
            float carry = -INFINITY;
            for (int64_t j0 = 0; j0 < n_kv; j0 += 64) {
                const int64_t j = j0 + lane;
                const float sj = j < n_kv ? fa_s[j] : -INFINITY;
                float x = sj;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const float o = __shfl_up(x, off, 64); if (lane >= off) x = fmaxf(x, o); }     // inclusive prefix maximum
                float prev = __shfl_up(x, 1, 64);
                prev = fmaxf(lane == 0 ? -INFINITY : prev, carry);
                carry = fmaxf(carry, __shfl(x, 63, 64));
                const bool live = sj != -INFINITY, newmax = live && sj > prev;
                const float e = expf_libm(newmax ? __fsub_rn(prev, sj) : live ? __fsub_rn(sj, prev) : 0.0f, tab);
                if (j < n_kv) { fa_s[j] = newmax ? 1.0f : live ? e : 0.0f; msv[j] = newmax ? e : 1.0f; }
            }
        }
        __syncthreads();
        if (tid < DV) {
            const char * vb = v.data + j_lo * v.nb[1] + hv * v.nb[2] + bv * v.nb[3] + 2 * (int64_t) tid;
            float acc = 0.0f, S = 0.0f;                                    // acc: an f16 value at all times (VKQ16)
#pragma unroll 4
            for (int64_t j = 0; j < n_kv; ++j) {
                const float vs = fa_s[j], ms = msv[j];
                if (vs != 0.0f || ms != 1.0f) {                             // (uniform) masked positions are skipped altogether
                    const float vv = __half2float(*(const __half *) (vb + j * v.nb[1]));
                    if (ms != 1.0f) acc = f16_round(__fmul_rn(acc, ms));         // ggml_vec_scale_f16
                    acc = f16_round(__builtin_fmaf(vv, vs, acc));                // ggml_vec_mad_f16
                    S = __fadd_rn(__fmul_rn(S, ms), vs);
                }
            }
            *(float *) (d.data + tid * d.nb[0] + h * d.nb[1] + t * d.nb[2] + b * d.nb[3]) = __fmul_rn(acc, __fdiv_rn(1.0f, S));
        }
        return;
    }
    // ---- row maximum, exponentials, their sum
    float mx = -INFINITY;
    for (int64_t j = tid; j < n_kv; j += FA_THREADS) mx = fmaxf(mx, fa_s[j]);
    mx = wave_max_f32(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w2 = 1; w2 < FA_WAVES; ++w2) mx = fmaxf(mx, red[w2]);
    float sum = 0.0f;
    for (int64_t j = tid; j < n_kv; j += FA_THREADS) { const float e = mx == -INFINITY ? 0.0f : expf(fa_s[j] - mx); fa_s[j] = e; sum += e; }   // (a fully masked piece contributes nothing)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) red[FA_WAVES + wave] = sum;
    __syncthreads();
    sum = 0.0f;
#pragma unroll
    for (int w2 = 0; w2 < FA_WAVES; ++w2) sum += red[FA_WAVES + w2];
    // ---- out[e] = sum_j p_j v[j][e] / sum: thread = (group g, element e); the groups split the kv positions
    const int per = (int) DV, groups = FA_THREADS / per, g = tid / per, e = tid - g * per;
    float acc = 0.0f;
    if (g < groups) {
        const char * vb = v.data + j_lo * v.nb[1] + hv * v.nb[2] + bv * v.nb[3] + 2 * (int64_t) e;
#pragma unroll 8
        for (int64_t j = g; j < n_kv; j += groups) acc += fa_s[j] * __half2float(*(const __half *) (vb + j * v.nb[1]));
    }
    __syncthreads();                                                        // everyone is done reading the probabilities
    float * part = fa_s;                                                     // (at least FA_THREADS floats of LDS are guaranteed by the launcher)
    part[tid] = acc;
    __syncthreads();
    if (tid < per) {
        float o = 0.0f;
        for (int gg = 0; gg < groups; ++gg) o += part[gg * per + tid];
        if (n_split == 1) *(float *) (d.data + tid * d.nb[0] + h * d.nb[1] + t * d.nb[2] + b * d.nb[3]) = o * (1.0f / sum);
        else {
            float * pr = part_out + ((((int64_t) b * gridDim.y + h) * (gridDim.x / n_split) + t) * n_split + sp) * (DV + 2);
            pr[tid] = o;
            if (tid == 0) { pr[DV] = mx; pr[DV + 1] = sum; }
        }
    }
}

// ---- FLASH_ATTN_EXT on a Q8_0 K / V cache (-ctk q8_0 -ctv q8_0; the reference: ggml-cuda/fattn.cu:244, CPU: ops.cpp:6686-6905) --------------
// One workgroup per (query row, head, batch), the CPU's arithmetic: q quantized to Q8_0 (the vec_dot_type of a Q8_0 K; the SIMD quantize_row_q8_0), scores by
// ggml_vec_dot_q8_0_q8_0 in its AVX2 order (per block d = d_k * d_q, eight f32 lane sums of four int8 products each fed by fma, hsum tree), the online
// softmax walked in order (prefix maximum, as in k_flash_attn_ext's seq mode), V dequantized (q * d) into an F32 accumulator: scale by f32 multiply when
// the maximum grows, v * weight by f32 fma per position (ggml_vec_scale_f32 / ggml_vec_mad_f32).
template <int KVT>                                                          // MI355Q_TYPE_Q8_0 (34-byte blocks) or MI355Q_TYPE_Q4_0 (18-byte blocks: f16 d, 16 bytes of nibbles, low = elements 0..15)
__global__ void __launch_bounds__(FA_THREADS) k_flash_attn_ext_q80(const TensorD q, const TensorD k, const TensorD v, const TensorD m, int has_mask, const TensorD d,
                                                                   float scale, float max_bias, float softcap, float m0, float m1, uint32_t n_head_log2) {
    extern __shared__ __attribute__((aligned(16))) float fq_s[];             // [n_kv] scores -> weights | [n_kv] rescale factors | q: [DK / 32] block scales, [DK / 4] packed quants
    constexpr int BB = KVT == MI355Q_TYPE_Q8_0 ? 34 : 18;                      // bytes per 32-element block of the cache
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t t = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int64_t DK = k.ne[0], DV = v.ne[0], n_kv = k.ne[1];
    const int64_t hk = h / (q.ne[2] / k.ne[2]), hv = h / (q.ne[2] / v.ne[2]), bk = b / (q.ne[3] / k.ne[3]), bv = b / (q.ne[3] / v.ne[3]);
    float * msv = fq_s + n_kv, * qd = msv + n_kv;
    uint32_t * qq = (uint32_t *) (qd + (DK >> 5));
    const float slope = max_bias > 0.0f ? ((uint32_t) h < n_head_log2 ? powf(m0, (float) (h + 1)) : powf(m1, (float) (2 * (h - n_head_log2) + 1))) : 1.0f;
    const __half * mp = has_mask ? (const __half *) (m.data + t * m.nb[1]) : nullptr;
    // ---- q -> Q8_0 (whole waves take part: the 8-lane reductions)
    if (tid < ((DK / 4 + 63) & ~63)) {
        const float * qp = (const float *) (q.data + t * q.nb[1] + h * q.nb[2] + b * q.nb[3]);
        const int e = 4 * tid;
        const float4 x = e < DK ? make_float4(qp[e], qp[e + 1], qp[e + 2], qp[e + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t pq; float dq; int sum;
        q80_group8<true>(x, pq, dq, sum);                                     // (round half to even: the CPU's SIMD quantize_row_q8_0, ggml-cpu-quants.c:842-845)
        if (e < DK) { qq[tid] = pq; if ((tid & 7) == 0) qd[tid >> 3] = __half2float(__float2half_rn(dq)); }
    }
    __syncthreads();
    // ---- scores: one position per thread
    const int nblk = (int) (DK >> 5);
    for (int64_t j = tid; j < n_kv; j += FA_THREADS) {
        const float mv = mp ? __fmul_rn(slope, __half2float(mp[j])) : 0.0f;
        float s = -INFINITY;
        if (mv != -INFINITY) {                                                // (a masked position is skipped by the CPU whatever its cache row holds)
            const char * row = k.data + j * k.nb[1] + hk * k.nb[2] + bk * k.nb[3];
            float acc[8] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
            for (int blk = 0; blk < nblk; ++blk) {
                const char * kb = row + BB * blk;
                const float dd = __fmul_rn(__half2float(*(const __half *) kb), qd[blk]);
                if constexpr (KVT == MI355Q_TYPE_Q8_0) {
#pragma unroll
                    for (int l = 0; l < 8; ++l) {
                        const uint32_t kw = (uint32_t) *(const uint16_t *) (kb + 2 + 4 * l) | ((uint32_t) *(const uint16_t *) (kb + 4 + 4 * l) << 16);
                        acc[l] = __builtin_fmaf(dd, (float) dot4((int) kw, (int) qq[8 * blk + l], 0), acc[l]);
                    }
                } else {
                    // ggml_vec_dot_q4_0_q8_0 (AVX2): bytes_from_nibbles_32 puts the low nibbles at elements 0..15 and the high ones at 16..31, minus 8 each
#pragma unroll
                    for (int l = 0; l < 4; ++l) {
                        const uint32_t kw = (uint32_t) *(const uint16_t *) (kb + 2 + 4 * l) | ((uint32_t) *(const uint16_t *) (kb + 4 + 4 * l) << 16);
                        const int qlo = (int) qq[8 * blk + l], qhi = (int) qq[8 * blk + 4 + l];
                        acc[l]     = __builtin_fmaf(dd, (float) (dot4((int) (kw & 0x0F0F0F0Fu), qlo, 0) - 8 * dot4(0x01010101, qlo, 0)), acc[l]);
                        acc[l + 4] = __builtin_fmaf(dd, (float) (dot4((int) ((kw >> 4) & 0x0F0F0F0Fu), qhi, 0) - 8 * dot4(0x01010101, qhi, 0)), acc[l + 4]);
                    }
                }
            }
            const float t0 = __fadd_rn(acc[0], acc[4]), t1 = __fadd_rn(acc[1], acc[5]), t2 = __fadd_rn(acc[2], acc[6]), t3 = __fadd_rn(acc[3], acc[7]);
            s = __fadd_rn(__fadd_rn(t0, t2), __fadd_rn(t1, t3));              // hsum_float_8
            s = __fmul_rn(s, scale);
            if (softcap != 0.0f) s = softcap * tanhf(s);
            s = __fadd_rn(s, mv);
        }
        fq_s[j] = s;
    }
    __syncthreads();
    // ---- online softmax in order: rescale factor and weight per position
    if (wave == 0) {
        const unsigned long long tab = PLAN_EXP2F_T[lane & 31];
        float carry = -INFINITY;
        for (int64_t j0 = 0; j0 < n_kv; j0 += 64) {
            const int64_t j = j0 + lane;
            const float sj = j < n_kv ? fq_s[j] : -INFINITY;
            float x = sj;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const float o = __shfl_up(x, off, 64); if (lane >= off) x = fmaxf(x, o); }
            float prev = __shfl_up(x, 1, 64);
            prev = fmaxf(lane == 0 ? -INFINITY : prev, carry);
            carry = fmaxf(carry, __shfl(x, 63, 64));
            const bool live = sj != -INFINITY, newmax = live && sj > prev;
            const float e = expf_libm(newmax ? __fsub_rn(prev, sj) : live ? __fsub_rn(sj, prev) : 0.0f, tab);
            if (j < n_kv) { fq_s[j] = newmax ? 1.0f : live ? e : 0.0f; msv[j] = newmax ? e : 1.0f; }
        }
    }
    __syncthreads();
    // ---- V . P in an F32 accumulator, a thread per output dim
    if (tid < DV) {
        const char * vb = v.data + hv * v.nb[2] + bv * v.nb[3] + BB * (tid >> 5);
        const int e = tid & 31;
        float acc = 0.0f, S = 0.0f;
#pragma unroll 4
        for (int64_t j = 0; j < n_kv; ++j) {
            const float vs = fq_s[j], ms = msv[j];
            if (vs != 0.0f || ms != 1.0f) {                                   // (uniform)
                const char * blk = vb + j * v.nb[1];
                int qv;
                if constexpr (KVT == MI355Q_TYPE_Q8_0) qv = (int) *(const int8_t *) (blk + 2 + e);
                else { const int byte = (int) *(const uint8_t *) (blk + 2 + (e & 15)); qv = ((e < 16 ? byte : byte >> 4) & 15) - 8; }
                const float vv = __fmul_rn((float) qv, __half2float(*(const __half *) blk));     // dequantize_row_q8_0 / _q4_0
                if (ms != 1.0f) acc = __fmul_rn(acc, ms);                     // ggml_vec_scale_f32
                acc = __builtin_fmaf(vv, vs, acc);                            // ggml_vec_mad_f32
                S = __fadd_rn(__fmul_rn(S, ms), vs);
            }
        }
        *(float *) (d.data + tid * d.nb[0] + h * d.nb[1] + t * d.nb[2] + b * d.nb[3]) = __fmul_rn(acc, __fdiv_rn(1.0f, S));
    }
}

// a quantized cache for the prefill path: K / V rows of all (kv head, batch) pairs -> F16 [B][H][n_kv][D] in scratch (value = q * d, rounded to f16), so that the
// matrix-core kernels of the f16 path can take them.  One thread per element pair.
template <int KVT>
__global__ void __launch_bounds__(256) k_dequant_cache_f16(const TensorD t, __half * __restrict__ out) {
    constexpr int BB = KVT == MI355Q_TYPE_Q8_0 ? 34 : 18;
    const int64_t D = t.ne[0], n_kv = t.ne[1], H = t.ne[2], B = t.ne[3], n = D * n_kv * H * B;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t e = i % D, r = i / D, j = r % n_kv, r2 = r / n_kv, h = r2 % H, b = r2 / H;
        const char * blk = t.data + j * t.nb[1] + h * t.nb[2] + b * t.nb[3] + BB * (e >> 5);
        const int ee = (int) (e & 31);
        int qv;
        if constexpr (KVT == MI355Q_TYPE_Q8_0) qv = (int) *(const int8_t *) (blk + 2 + ee);
        else { const int byte = (int) *(const uint8_t *) (blk + 2 + (ee & 15)); qv = ((ee < 16 ? byte : byte >> 4) & 15) - 8; }
        out[i] = __float2half_rn(__fmul_rn((float) qv, __half2float(*(const __half *) blk)));
    }
}

// merge the pieces of a split row: out = sum_s e^(m_s - M) o_s / sum_s e^(m_s - M) l_s
__global__ void __launch_bounds__(256) k_flash_attn_combine(const float * __restrict__ part, int n_split, int64_t DV, int64_t N, const TensorD d) {
    const int64_t t = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const float * pr = part + (((b * gridDim.y + h) * N + t) * n_split) * (DV + 2);
    float M = -INFINITY;
    for (int s = 0; s < n_split; ++s) M = fmaxf(M, pr[s * (DV + 2) + DV]);
    float den = 0.0f;
    for (int s = 0; s < n_split; ++s) { const float ms = pr[s * (DV + 2) + DV]; den += (ms == -INFINITY ? 0.0f : expf(ms - M)) * pr[s * (DV + 2) + DV + 1]; }
    for (int64_t e = threadIdx.x; e < DV; e += 256) {
        float o = 0.0f;
        for (int s = 0; s < n_split; ++s) { const float ms = pr[s * (DV + 2) + DV]; o += (ms == -INFINITY ? 0.0f : expf(ms - M)) * pr[s * (DV + 2) + e]; }
        *(float *) (d.data + e * d.nb[0] + h * d.nb[1] + t * d.nb[2] + b * d.nb[3]) = o / den;
    }
}


// ---- the same product for many src1 rows (prefill attention: KQ and KQV of a 512-token batch) on the matrix cores -------------------
// dst[n][m] = sum_k a[m][k] * f16(b[n][k]),  a f16.  64 x 64 output tile per workgroup (2 x 2 waves of 2 x 2 v_mfma_f32_16x16x32_f16), K in
// steps of 64 through LDS (a copied as it is, b rounded to f16 on the way in, rows padded by 16 B against bank conflicts).  Products of two
// f16 numbers are exact in f32 and the accumulation is f32, as in the CPU's f16 dot product; only the summation order differs.
typedef _Float16 mmf_h8 __attribute__((ext_vector_type(8)));
typedef float    mmf_f4 __attribute__((ext_vector_type(4)));
constexpr int MMF_T = 64, MMF_K = 64, MMF_LDS_ROW = MMF_K * 2 + 16;
// AT: src0 is given TRANSPOSED in memory -- a.data[k][m] with m contiguous (a.nb[1] = bytes between consecutive k): the V cache of the
// flash-attention layout, one row per position, read for out[n][m] = sum_k p[n][k] v[k][m].  a.ne still reads [K, M, ..].
template <bool AT>
__global__ void __launch_bounds__(256) k_mul_mat_f16_mfma(const TensorD a, const TensorD b, const TensorD d) {
    __shared__ __attribute__((aligned(16))) uint8_t As[MMF_T * MMF_LDS_ROW], Bs[MMF_T * MMF_LDS_ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int64_t M = d.ne[0], N = d.ne[1], K = a.ne[0];
    const int64_t m0 = (int64_t) blockIdx.x * MMF_T, n0 = (int64_t) blockIdx.y * MMF_T;
    const int64_t i12 = blockIdx.z % d.ne[2], i13 = blockIdx.z / d.ne[2];
    const int64_t i02 = i12 / (b.ne[2] / a.ne[2]), i03 = i13 / (b.ne[3] / a.ne[3]);
    const int row = tid >> 2, seg = tid & 3;                                  // staging: thread -> (tile row, 16-element k segment)
    const int64_t am = m0 + row < M ? m0 + row : M - 1, bn = n0 + row < N ? n0 + row : N - 1;
    const char * ap = AT ? a.data + i02 * a.nb[2] + i03 * a.nb[3] : a.data + am * a.nb[1] + i02 * a.nb[2] + i03 * a.nb[3];
    const char * bp = b.data + bn * b.nb[1] + i12 * b.nb[2] + i13 * b.nb[3];
    mmf_f4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (mmf_f4) { 0.f, 0.f, 0.f, 0.f };
    // the operands of step k0 + 64 are loaded while the matrix cores work on step k0
    uint4 a0, a1; float4 f0, f1, f2, f3;
    auto fetch = [&](int64_t k0) {
        const int64_t ks = k0 + 16 * seg;
        a0 = make_uint4(0, 0, 0, 0); a1 = a0;
        f0 = make_float4(0.f, 0.f, 0.f, 0.f); f1 = f0; f2 = f0; f3 = f0;
        if (ks < K) {                                                         // K is a multiple of 16 (checked by the launcher)
            if constexpr (!AT) { a0 = *(const uint4 *) (ap + 2 * ks); a1 = *(const uint4 *) (ap + 2 * ks + 16); }
            const float4 * bq = (const float4 *) (bp + 4 * ks);
            f0 = bq[0]; f1 = bq[1]; f2 = bq[2]; f3 = bq[3];
        }
        if constexpr (AT) {                                                   // thread -> (k = k0 + row, 16 consecutive m): transposed on the way into LDS
            if (k0 + row < K && m0 + 16 * seg < M) {                          // (M is a multiple of 16: checked by the launcher)
                const char * vp = ap + (k0 + row) * a.nb[1] + 2 * (m0 + 16 * seg);
                a0 = *(const uint4 *) vp; a1 = *(const uint4 *) (vp + 16);
            }
        }
    };
    fetch(0);
    for (int64_t k0 = 0; k0 < K; k0 += MMF_K) {
        __syncthreads();                                                      // the previous step's fragments have been read
        if constexpr (AT) {
            const uint32_t wv[8] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                *(uint16_t *) (As + (16 * seg + 2 * i) * MMF_LDS_ROW + 2 * row)     = (uint16_t) (wv[i] & 0xFFFFu);
                *(uint16_t *) (As + (16 * seg + 2 * i + 1) * MMF_LDS_ROW + 2 * row) = (uint16_t) (wv[i] >> 16);
            }
        } else {
            *(uint4 *) (As + row * MMF_LDS_ROW + 32 * seg) = a0; *(uint4 *) (As + row * MMF_LDS_ROW + 32 * seg + 16) = a1;
        }
        const mmf_h8 h0 = { (_Float16) f0.x, (_Float16) f0.y, (_Float16) f0.z, (_Float16) f0.w, (_Float16) f1.x, (_Float16) f1.y, (_Float16) f1.z, (_Float16) f1.w };
        const mmf_h8 h1 = { (_Float16) f2.x, (_Float16) f2.y, (_Float16) f2.z, (_Float16) f2.w, (_Float16) f3.x, (_Float16) f3.y, (_Float16) f3.z, (_Float16) f3.w };
        *(mmf_h8 *) (Bs + row * MMF_LDS_ROW + 32 * seg) = h0; *(mmf_h8 *) (Bs + row * MMF_LDS_ROW + 32 * seg + 16) = h1;
        __syncthreads();
        if (k0 + MMF_K < K) fetch(k0 + MMF_K);
#pragma unroll
        for (int kk = 0; kk < MMF_K / 32; ++kk) {
            const int koff = 2 * (32 * kk + 8 * (lane >> 4));
            mmf_h8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *(const mmf_h8 *) (As + (32 * wm + 16 * i + (lane & 15)) * MMF_LDS_ROW + koff);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *(const mmf_h8 *) (Bs + (32 * wn + 16 * j + (lane & 15)) * MMF_LDS_ROW + koff);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    // C rows = src0 rows m (4 consecutive per lane), C column = src1 row n
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t n = n0 + 32 * wn + 16 * j + (lane & 15);
        if (n >= N) continue;
        char * dr = d.data + n * d.nb[1] + i12 * d.nb[2] + i13 * d.nb[3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t m = m0 + 32 * wm + 16 * i + 4 * (lane >> 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) if (m + r < M) *(float *) (dr + (m + r) * d.nb[0]) = acc[i][j][r];
        }
    }
}

// ---- GET_ROWS (f32 / f16 rows -> f32): dst[:, i10, i11, i12] = src0[:, ids[i10, i11, i12], i11, i12]   ops.cpp:4272-4311 -------
__global__ void __launch_bounds__(256) k_get_rows(const TensorD a, const char * ids, int64_t nb10, int64_t nb11, int64_t nb12,
                                                  int64_t ne10, int64_t ne11, const TensorD d, int64_t n) {
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t i0 = i % d.ne[0], r = i / d.ne[0];
        const int64_t i12 = r / (ne11 * ne10), i11 = (r - i12 * ne11 * ne10) / ne10, i10 = r - i12 * ne11 * ne10 - i11 * ne10;
        const int64_t i01 = *(const int32_t *) (ids + i10 * nb10 + i11 * nb11 + i12 * nb12);
        if (i01 < 0 || i01 >= a.ne[1]) continue;                 // (the reference asserts; the row is left untouched here)
        const float v = ld_elem(a.data + i0 * a.nb[0] + i01 * a.nb[1] + i11 * a.nb[2] + i12 * a.nb[3], a.type);
        *(float *) (d.data + i0 * d.nb[0] + i10 * d.nb[1] + i11 * d.nb[2] + i12 * d.nb[3]) = v;
    }
}

// ---- SCALE: dst = a * s   (ops.cpp:3840-3878, ggml_vec_scale_f32) -----------------------------------------------------
// ---- ARGSORT (ggml-cpu/ops.cpp ggml_compute_forward_argsort_f32): dst[row][r] = index of the element of rank r.  One workgroup per row; the
// row sits in LDS and element i finds its rank by counting the elements that sort before it (ties keep their index order) -- a row is the expert
// scores of a token (8 .. 256 values), so the quadratic count is a handful of LDS reads per thread.
__global__ void __launch_bounds__(256) k_argsort(const TensorD a, const TensorD d, int desc) {
    extern __shared__ float row[];
    const int64_t r = blockIdx.x, n = a.ne[0];
    const int64_t i1 = r % a.ne[1], r2 = r / a.ne[1], i2 = r2 % a.ne[2], i3 = r2 / a.ne[2];
    const char * src = a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3];
    int32_t * dst = (int32_t *) (d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
    for (int64_t i = threadIdx.x; i < n; i += 256) row[i] = *(const float *) (src + i * a.nb[0]);
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float v = row[i];
        int rank = 0;
        for (int64_t j = 0; j < n; ++j) {
            const float w = row[j];
            rank += desc ? (w > v || (w == v && j < i)) : (w < v || (w == v && j < i));
        }
        dst[rank] = (int32_t) i;
    }
}

// ---- SUM_ROWS (ggml-cpu/ops.cpp ggml_compute_forward_sum_rows_f32: ggml_vec_sum_f32 accumulates in ggml_float = double): one wave per row
__global__ void __launch_bounds__(256) k_sum_rows(const TensorD a, const TensorD d, int64_t nrows) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int64_t i1 = r % a.ne[1], r2 = r / a.ne[1], i2 = r2 % a.ne[2], i3 = r2 / a.ne[2];
    const char * src = a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3];
    double s = 0.0;
    for (int64_t i = lane; i < a.ne[0]; i += 64) s += (double) *(const float *) (src + i * a.nb[0]);
    s = wave_sum_f64(s);
    if (lane == 0) *(float *) (d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]) = (float) s;
}

__global__ void __launch_bounds__(256) k_scale(const TensorD a, const TensorD d, float sc, int64_t n) {
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t i0 = i % d.ne[0], r = i / d.ne[0], i1 = r % d.ne[1], r2 = r / d.ne[1], i2 = r2 % d.ne[2], i3 = r2 / d.ne[2];
        const float x = *(const float *) (a.data + i0 * a.nb[0] + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
        *(float *) (d.data + i0 * d.nb[0] + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]) = __fmul_rn(x, sc);
    }
}

static int grid_for(int64_t n) { const int64_t g = (n + 255) / 256; return (int) (g < 1 ? 1 : (g > 65535 ? 65535 : g)); }

} // namespace mi355q

using namespace mi355q;

extern "C" {

void mi355q_set_error(const char * msg);          // api.hip

#define OPS_FAIL(code, msg) do { mi355q_set_error(msg); return code; } while (0)
#define OPS_LAUNCHED() return hipGetLastError() == hipSuccess ? MI355Q_OK : (mi355q_set_error("kernel launch failed"), MI355Q_ERR_HIP)

int mi355q_op_bin_bcast(int op, const mi355q_tensor * a, const mi355q_tensor * b, const mi355q_tensor * dst, void * stream) {
    if (!a || !b || !dst) OPS_FAIL(MI355Q_ERR_SHAPE, "op_bin_bcast: null tensor");
    if (!same_shape(a, dst)) OPS_FAIL(MI355Q_ERR_SHAPE, "op_bin_bcast: dst must have the shape of src0");
    for (int i = 0; i < 4; ++i) if (b->ne[i] <= 0 || a->ne[i] % b->ne[i]) OPS_FAIL(MI355Q_ERR_SHAPE, "op_bin_bcast: src1 does not broadcast over src0");
    const int64_t n = nelements(dst);
    if (n == 0) return MI355Q_OK;
    const dim3 g(grid_for(n)), t(256);
    switch (op) {
    case MI355Q_OP_ADD: hipLaunchKernelGGL(k_bin_bcast<BIN_ADD>, g, t, 0, (hipStream_t) stream, to_d(a), to_d(b), to_d(dst), n); break;
    case MI355Q_OP_SUB: hipLaunchKernelGGL(k_bin_bcast<BIN_SUB>, g, t, 0, (hipStream_t) stream, to_d(a), to_d(b), to_d(dst), n); break;
    case MI355Q_OP_MUL: hipLaunchKernelGGL(k_bin_bcast<BIN_MUL>, g, t, 0, (hipStream_t) stream, to_d(a), to_d(b), to_d(dst), n); break;
    case MI355Q_OP_DIV: hipLaunchKernelGGL(k_bin_bcast<BIN_DIV>, g, t, 0, (hipStream_t) stream, to_d(a), to_d(b), to_d(dst), n); break;
    default: OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_bin_bcast: unknown op");
    }
    OPS_LAUNCHED();
}

int mi355q_op_unary(int uop, const mi355q_tensor * a, const mi355q_tensor * dst, void * stream) {
    if (!a || !dst || !same_shape(a, dst)) OPS_FAIL(MI355Q_ERR_SHAPE, "op_unary: shapes differ");
    const int64_t n = nelements(dst);
    if (n == 0) return MI355Q_OK;
    const dim3 g(grid_for(n)), t(256);
#define U(OPV) case OPV: hipLaunchKernelGGL(k_unary<OPV>, g, t, 0, (hipStream_t) stream, to_d(a), to_d(dst), n); break;
    switch (uop) {
    U(MI355Q_UNARY_SILU) U(MI355Q_UNARY_RELU) U(MI355Q_UNARY_SIGMOID) U(MI355Q_UNARY_TANH) U(MI355Q_UNARY_NEG) U(MI355Q_UNARY_ABS)
    default: OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_unary: unknown op");
    }
#undef U
    OPS_LAUNCHED();
}

static void launch_rms(const mi355q_tensor * a, const mi355q_tensor * b, const mi355q_tensor * sum, const float * weight, const mi355q_tensor * dst,
                       float eps, int64_t nrows, hipStream_t st) {
    auto al16 = [](const mi355q_tensor * t) { return !t || ((((uintptr_t) t->data) | (uintptr_t) t->nb[1] | (uintptr_t) t->nb[2] | (uintptr_t) t->nb[3]) & 15) == 0; };
    const bool vec = a->ne[0] % 4 == 0 && al16(a) && al16(b) && al16(sum) && al16(dst) && (((uintptr_t) weight) & 15) == 0;
    const dim3 grid((unsigned) nrows), block(256);
    const TensorD ta = to_d(a), tb = b ? to_d(b) : to_d(a), ts = sum ? to_d(sum) : to_d(a), td = to_d(dst);
    if (vec && a->ne[0] <= 4096)       hipLaunchKernelGGL(k_add_rms_norm_mul_vec<4>,  grid, block, 0, st, ta, tb, b ? 1 : 0, ts, sum ? 1 : 0, weight, td, eps);
    else if (vec && a->ne[0] <= 16384) hipLaunchKernelGGL(k_add_rms_norm_mul_vec<16>, grid, block, 0, st, ta, tb, b ? 1 : 0, ts, sum ? 1 : 0, weight, td, eps);
    else                               hipLaunchKernelGGL(k_add_rms_norm_mul,        grid, block, 0, st, ta, tb, b ? 1 : 0, ts, sum ? 1 : 0, weight, td, eps);
}

int mi355q_op_rms_norm(const mi355q_tensor * a, const mi355q_tensor * dst, float eps, void * stream) {
    if (!a || !dst || !same_shape(a, dst) || a->type != 0 || dst->type != 0) OPS_FAIL(MI355Q_ERR_SHAPE, "op_rms_norm: f32 tensors of one shape");
    if (a->nb[0] != 4 || dst->nb[0] != 4) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_rms_norm: rows must be contiguous");
    const int64_t nrows = a->ne[1] * a->ne[2] * a->ne[3];
    if (nrows == 0 || a->ne[0] == 0) return MI355Q_OK;
    launch_rms(a, nullptr, nullptr, nullptr, dst, eps, nrows, (hipStream_t) stream);
    OPS_LAUNCHED();
}

static bool contiguous_f32(const mi355q_tensor * t) {
    return t->type == 0 && t->nb[0] == 4 && t->nb[1] == 4 * t->ne[0] && t->nb[2] == t->nb[1] * t->ne[1] && t->nb[3] == t->nb[2] * t->ne[2];
}

int mi355q_op_add_rms_norm_mul(const mi355q_tensor * a, const mi355q_tensor * b, const mi355q_tensor * sum, const float * weight,
                               const mi355q_tensor * dst, float eps, void * stream) {
    if (!a || !dst || !same_shape(a, dst) || a->type != 0 || dst->type != 0) OPS_FAIL(MI355Q_ERR_SHAPE, "op_add_rms_norm_mul: f32 tensors of one shape");
    if ((b && (!same_shape(a, b) || b->type != 0)) || (sum && (!b || !same_shape(a, sum) || sum->type != 0))) OPS_FAIL(MI355Q_ERR_SHAPE, "op_add_rms_norm_mul: addend / sum shape");
    if (a->nb[0] != 4 || dst->nb[0] != 4 || (b && b->nb[0] != 4) || (sum && sum->nb[0] != 4)) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_add_rms_norm_mul: rows must be contiguous");
    const int64_t nrows = a->ne[1] * a->ne[2] * a->ne[3];
    if (nrows == 0 || a->ne[0] == 0) return MI355Q_OK;
    launch_rms(a, b, sum, weight, dst, eps, nrows, (hipStream_t) stream);
    OPS_LAUNCHED();
}

int mi355q_op_unary_mul(int uop, const mi355q_tensor * a, const mi355q_tensor * b, const mi355q_tensor * dst, void * stream) {
    if (!a || !b || !dst || !same_shape(a, dst) || !same_shape(b, dst)) OPS_FAIL(MI355Q_ERR_SHAPE, "op_unary_mul: shapes differ");
    if (!contiguous_f32(a) || !contiguous_f32(b) || !contiguous_f32(dst)) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_unary_mul: contiguous f32 tensors only");
    const int64_t n = nelements(dst);
    if (n == 0) return MI355Q_OK;
    const dim3 g(grid_for(n)), t(256);
#define U(OPV) case OPV: hipLaunchKernelGGL(k_unary_mul<OPV>, g, t, 0, (hipStream_t) stream, (const float *) a->data, (const float *) b->data, (float *) dst->data, n); break;
    switch (uop) {
    U(MI355Q_UNARY_SILU) U(MI355Q_UNARY_RELU) U(MI355Q_UNARY_SIGMOID)
    default: OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_unary_mul: SiLU / ReLU / sigmoid only");
    }
#undef U
    OPS_LAUNCHED();
}

// f32 -> Q8_0: contiguous destination, source rows of whole blocks with unit stride along dim 0 (what ggml_compute_forward_dup_f32 requires too)
static int cpy_to_q8_0(const mi355q_tensor * a, const mi355q_tensor * dst, char * const * table, int index, hipStream_t st) {
    const int64_t n = nelements(dst);
    const int bb = dst->type == MI355Q_TYPE_Q8_0 ? 34 : 18;     // (Q8_0 or Q4_0 blocks)
    if (a->type != 0 || a->nb[0] != 4 || a->ne[0] % 32 != 0 || dst->ne[0] % 32 != 0 || dst->nb[0] != bb) return MI355Q_ERR_UNSUPPORTED;
    if (dst->nb[1] != dst->ne[0] / 32 * bb || dst->nb[2] != dst->nb[1] * dst->ne[1] || dst->nb[3] != dst->nb[2] * dst->ne[2]) return MI355Q_ERR_UNSUPPORTED;
    if (n == 0) return MI355Q_OK;
    if (dst->type == MI355Q_TYPE_Q8_0) hipLaunchKernelGGL(k_cpy_f32_q8_0, dim3(grid_for(n / 4)), dim3(256), 0, st, to_d(a), (char *) dst->data, n / 32, table, index);
    else                               hipLaunchKernelGGL(k_cpy_f32_q4_0, dim3(grid_for(n / 4)), dim3(256), 0, st, to_d(a), (char *) dst->data, n / 32, table, index);
    return MI355Q_OK;
}

int mi355q_op_cpy(const mi355q_tensor * a, const mi355q_tensor * dst, void * stream) {
    if (!a || !dst || nelements(a) != nelements(dst)) OPS_FAIL(MI355Q_ERR_SHAPE, "op_cpy: element counts differ");
    if (dst->type == MI355Q_TYPE_Q8_0 || dst->type == MI355Q_TYPE_Q4_0) {
        if (cpy_to_q8_0(a, dst, nullptr, 0, (hipStream_t) stream) != MI355Q_OK) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_cpy: f32 -> q8_0 / q4_0 needs a contiguous destination and source rows of whole 32-blocks");
        OPS_LAUNCHED();
    }
    if (a->type < 0 || a->type > 1 || dst->type < 0 || dst->type > 1) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_cpy: f32 / f16 only");
    const int64_t n = nelements(dst);
    if (n == 0) return MI355Q_OK;
    if (!cpy_fast(a, dst, nullptr, 0, (hipStream_t) stream))
        hipLaunchKernelGGL(k_cpy, dim3(grid_for(n)), dim3(256), 0, (hipStream_t) stream, to_d(a), to_d(dst), n, (char * const *) nullptr, 0);
    OPS_LAUNCHED();
}

int mi355q_op_cpy_indirect(const mi355q_tensor * a, const mi355q_tensor * dst, void * const * dest_table, int index, void * stream) {
    if (!a || !dst || !dest_table || index < 0 || nelements(a) != nelements(dst)) OPS_FAIL(MI355Q_ERR_SHAPE, "op_cpy_indirect: arguments");
    if (dst->type == MI355Q_TYPE_Q8_0 || dst->type == MI355Q_TYPE_Q4_0) {
        if (cpy_to_q8_0(a, dst, (char * const *) dest_table, index, (hipStream_t) stream) != MI355Q_OK) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_cpy_indirect: f32 -> q8_0 / q4_0 needs a contiguous destination and source rows of whole 32-blocks");
        OPS_LAUNCHED();
    }
    if (a->type < 0 || a->type > 1 || dst->type < 0 || dst->type > 1) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_cpy_indirect: f32 / f16 only");
    const int64_t n = nelements(dst);
    if (n == 0) return MI355Q_OK;
    if (!cpy_fast(a, dst, (char * const *) dest_table, index, (hipStream_t) stream))
        hipLaunchKernelGGL(k_cpy, dim3(grid_for(n)), dim3(256), 0, (hipStream_t) stream, to_d(a), to_d(dst), n, (char * const *) dest_table, index);
    OPS_LAUNCHED();
}

struct mi355q_graph { hipGraph_t graph; hipGraphExec_t exec; };

int mi355q_graph_capture_begin(void * stream) {
    if (hipStreamBeginCapture((hipStream_t) stream, hipStreamCaptureModeThreadLocal) != hipSuccess) OPS_FAIL(MI355Q_ERR_HIP, "graph_capture_begin failed");
    return MI355Q_OK;
}
int mi355q_graph_capture_end(void * stream, mi355q_graph ** out) {
    if (out) *out = nullptr;
    hipGraph_t g = nullptr;
    if (hipStreamEndCapture((hipStream_t) stream, &g) != hipSuccess || !g) { (void) hipGetLastError(); OPS_FAIL(MI355Q_ERR_HIP, "graph_capture_end: capture failed"); }
    hipGraphExec_t e = nullptr;
    if (hipGraphInstantiate(&e, g, nullptr, nullptr, 0) != hipSuccess || !e) { (void) hipGraphDestroy(g); (void) hipGetLastError(); OPS_FAIL(MI355Q_ERR_HIP, "graph_capture_end: instantiate failed"); }
    if (!out) { (void) hipGraphExecDestroy(e); (void) hipGraphDestroy(g); return MI355Q_OK; }
    *out = new mi355q_graph{ g, e };
    return MI355Q_OK;
}
int mi355q_graph_launch(mi355q_graph * graph, void * stream) {
    if (!graph) OPS_FAIL(MI355Q_ERR_SHAPE, "graph_launch: null graph");
    if (hipGraphLaunch(graph->exec, (hipStream_t) stream) != hipSuccess) OPS_FAIL(MI355Q_ERR_HIP, "graph_launch failed");
    return MI355Q_OK;
}
int mi355q_graph_destroy(mi355q_graph * graph) {
    if (!graph) return MI355Q_OK;
    (void) hipGraphExecDestroy(graph->exec); (void) hipGraphDestroy(graph->graph);
    delete graph;
    return MI355Q_OK;
}

int mi355q_op_soft_max(const mi355q_tensor * a, const mi355q_tensor * mask, const mi355q_tensor * dst, float scale, float max_bias, void * stream) {
    if (!a || !dst || !same_shape(a, dst) || a->type != 0 || dst->type != 0) OPS_FAIL(MI355Q_ERR_SHAPE, "op_soft_max: f32 tensors of one shape");
    // rows dense and equally spaced (the reference asserts contiguity of dst and walks src0 rows by nb[1])
    if (a->nb[0] != 4 || dst->nb[0] != 4 || a->nb[2] != a->nb[1] * a->ne[1] || a->nb[3] != a->nb[2] * a->ne[2] ||
        dst->nb[1] != dst->ne[0] * 4 || dst->nb[2] != dst->nb[1] * dst->ne[1] || dst->nb[3] != dst->nb[2] * dst->ne[2])
        OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_soft_max: rows must be equally spaced");
    if (mask && (mask->ne[0] != a->ne[0] || mask->ne[1] < a->ne[1] || mask->nb[0] != (mask->type == 0 ? 4 : 2) ||
                 mask->nb[1] != mask->ne[0] * mask->nb[0])) OPS_FAIL(MI355Q_ERR_SHAPE, "op_soft_max: mask must be contiguous [ne00, >= ne01]");
    const int64_t nrows = a->ne[1] * a->ne[2] * a->ne[3];
    if (nrows == 0 || a->ne[0] == 0) return MI355Q_OK;
    const uint32_t n_head = (uint32_t) a->ne[2];
    uint32_t n_head_log2 = 1; while (2 * n_head_log2 <= n_head) n_head_log2 *= 2;
    const float m0 = powf(2.0f, -(max_bias) / n_head_log2), m1 = powf(2.0f, -(max_bias / 2.0f) / n_head_log2);
    TensorD md = mask ? to_d(mask) : to_d(a);
    const dim3 grid((unsigned) ((nrows + 3) / 4)), block(256);
    if (a->ne[0] <= 512)       hipLaunchKernelGGL(k_soft_max_reg<8>,  grid, block, 0, (hipStream_t) stream, to_d(a), md, to_d(dst), scale, max_bias, m0, m1, n_head_log2, nrows, mask ? 1 : 0);
    else if (a->ne[0] <= 2048) hipLaunchKernelGGL(k_soft_max_reg<32>, grid, block, 0, (hipStream_t) stream, to_d(a), md, to_d(dst), scale, max_bias, m0, m1, n_head_log2, nrows, mask ? 1 : 0);
    else                       hipLaunchKernelGGL(k_soft_max,         grid, block, 0, (hipStream_t) stream, to_d(a), md, to_d(dst), scale, max_bias, m0, m1, n_head_log2, nrows, mask ? 1 : 0);
    OPS_LAUNCHED();
}

int mi355q_op_get_rows(const mi355q_tensor * a, const mi355q_tensor * ids, const mi355q_tensor * dst, void * stream) {
    if (!a || !ids || !dst) OPS_FAIL(MI355Q_ERR_SHAPE, "op_get_rows: null tensor");
    if ((a->type != 0 && a->type != 1) || dst->type != 0) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_get_rows: f32 / f16 rows to f32");
    if (dst->ne[0] != a->ne[0] || dst->ne[1] != ids->ne[0] || dst->ne[2] != ids->ne[1] || dst->ne[3] != ids->ne[2] || a->ne[2] != ids->ne[1])
        OPS_FAIL(MI355Q_ERR_SHAPE, "op_get_rows: shapes (dst [ne00, ne10, ne11, ne12], ne02 == ne11)");
    const int64_t n = nelements(dst);
    if (n == 0) return MI355Q_OK;
    hipLaunchKernelGGL(k_get_rows, dim3(grid_for(n)), dim3(256), 0, (hipStream_t) stream, to_d(a), (const char *) ids->data, ids->nb[0], ids->nb[1], ids->nb[2],
                       ids->ne[0], ids->ne[1], to_d(dst), n);
    OPS_LAUNCHED();
}

int mi355q_op_scale(const mi355q_tensor * a, const mi355q_tensor * dst, float scale, void * stream) {
    if (!a || !dst || !same_shape(a, dst) || a->type != 0 || dst->type != 0) OPS_FAIL(MI355Q_ERR_SHAPE, "op_scale: f32 tensors of one shape");
    const int64_t n = nelements(dst);
    if (n == 0) return MI355Q_OK;
    hipLaunchKernelGGL(k_scale, dim3(grid_for(n)), dim3(256), 0, (hipStream_t) stream, to_d(a), to_d(dst), scale, n);
    OPS_LAUNCHED();
}

int mi355q_op_argsort(const mi355q_tensor * a, const mi355q_tensor * dst, int descending, void * stream) {
    if (!a || !dst || !same_shape(a, dst) || a->type != 0) OPS_FAIL(MI355Q_ERR_SHAPE, "op_argsort: f32 source, i32 destination of the same shape");
    if (dst->nb[0] != 4) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_argsort: destination rows must be contiguous");
    if (a->ne[0] > 32768) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_argsort: rows of more than 32768 elements");
    const int64_t nrows = a->ne[1] * a->ne[2] * a->ne[3];
    if (nrows == 0 || a->ne[0] == 0) return MI355Q_OK;
    if (nrows > 0x7FFFFFFF) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_argsort: too many rows");
    if (a->ne[0] * 4 > 48 * 1024) {
        static bool big[64] = {};
        int dev = 0; (void) hipGetDevice(&dev); dev = dev >= 0 && dev < 64 ? dev : 0;
        if (!big[dev]) { if (hipFuncSetAttribute((const void *) k_argsort, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) OPS_FAIL(MI355Q_ERR_HIP, "op_argsort: LDS attribute"); big[dev] = true; }
    }
    hipLaunchKernelGGL(k_argsort, dim3((unsigned) nrows), dim3(256), (size_t) a->ne[0] * 4, (hipStream_t) stream, to_d(a), to_d(dst), descending ? 1 : 0);
    OPS_LAUNCHED();
}

int mi355q_op_sum_rows(const mi355q_tensor * a, const mi355q_tensor * dst, void * stream) {
    if (!a || !dst || a->type != 0 || dst->type != 0 || dst->ne[0] != 1 || dst->ne[1] != a->ne[1] || dst->ne[2] != a->ne[2] || dst->ne[3] != a->ne[3])
        OPS_FAIL(MI355Q_ERR_SHAPE, "op_sum_rows: f32 [ne0, ...] -> f32 [1, ...]");
    const int64_t nrows = a->ne[1] * a->ne[2] * a->ne[3];
    if (nrows == 0) return MI355Q_OK;
    if ((nrows + 3) / 4 > 0x7FFFFFFF) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_sum_rows: too many rows");
    hipLaunchKernelGGL(k_sum_rows, dim3((unsigned) ((nrows + 3) / 4)), dim3(256), 0, (hipStream_t) stream, to_d(a), to_d(dst), nrows);
    OPS_LAUNCHED();
}

int mi355q_op_rope(const mi355q_tensor * a, const int32_t * pos, const float * freq_factors, const mi355q_tensor * dst,
                   const mi355q_rope_params * p, void * stream) {
    if (!a || !dst || !pos || !p || !same_shape(a, dst) || a->type != 0 || dst->type != 0) OPS_FAIL(MI355Q_ERR_SHAPE, "op_rope: f32 tensors of one shape, positions");
    if (a->nb[0] != 4 || dst->nb[0] != 4) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_rope: rows must be contiguous");
    if (p->mode != 0 && p->mode != 2) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_rope: only the normal (0) and neox (2) modes");
    if (p->n_dims <= 0 || p->n_dims % 2 || p->n_dims > a->ne[0] || a->ne[0] % 2) OPS_FAIL(MI355Q_ERR_SHAPE, "op_rope: n_dims must be even and <= ne0");
    const int64_t n_pairs = nelements(a) / 2;
    if (n_pairs == 0) return MI355Q_OK;
    RopeP rp;
    rp.n_dims = p->n_dims; rp.neox = p->mode == 2; rp.freq_scale = p->freq_scale; rp.ext_factor = p->ext_factor; rp.attn_factor = p->attn_factor;
    rp.theta_scale = powf(p->freq_base, -2.0f / p->n_dims);
    // ggml_rope_yarn_corr_dims, ggml.c:3729-3743
    auto corr_dim = [&](float n_rot) { return p->n_dims * logf(p->n_ctx_orig / (n_rot * 2 * 3.14159265358979323846f)) / (2 * logf(p->freq_base)); };
    const float start = floorf(corr_dim(p->beta_fast)), end = ceilf(corr_dim(p->beta_slow));
    rp.corr0 = start > 0 ? start : 0; rp.corr1 = end < p->n_dims - 1 ? end : (float) (p->n_dims - 1);
    const int64_t tokens = a->ne[2] * a->ne[3];
    if (tokens >= 32 && a->ne[1] >= 2) {                        // a batch: the angle of a (position, pair) is shared by a group of heads
        int hg = 1;
        while (hg < 8 && 2 * hg <= a->ne[1] && tokens * ((a->ne[1] + 2 * hg - 1) / (2 * hg)) >= 2048) hg *= 2;
        const int64_t n_waves = tokens * ((a->ne[1] + hg - 1) / hg);
        hipLaunchKernelGGL(k_rope_rows, dim3((unsigned) ((n_waves + 3) / 4)), dim3(256), 0, (hipStream_t) stream, to_d(a), pos, freq_factors, to_d(dst), rp, hg, n_waves);
    } else
        hipLaunchKernelGGL(k_rope, dim3(grid_for(n_pairs)), dim3(256), 0, (hipStream_t) stream, to_d(a), pos, freq_factors, to_d(dst), rp, n_pairs);
    OPS_LAUNCHED();
}

int mi355q_op_mul_mat_f(const mi355q_tensor * a, const mi355q_tensor * b, const mi355q_tensor * dst, void * stream) {
    if (!a || !b || !dst) OPS_FAIL(MI355Q_ERR_SHAPE, "op_mul_mat_f: null tensor");
    if ((a->type != 0 && a->type != 1) || b->type != 0 || dst->type != 0) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_mul_mat_f: src0 f16/f32, src1 and dst f32");
    if (a->ne[0] != b->ne[0] || dst->ne[0] != a->ne[1] || dst->ne[1] != b->ne[1] || dst->ne[2] != b->ne[2] || dst->ne[3] != b->ne[3] ||
        a->ne[2] <= 0 || a->ne[3] <= 0 || b->ne[2] % a->ne[2] || b->ne[3] % a->ne[3]) OPS_FAIL(MI355Q_ERR_SHAPE, "op_mul_mat_f: shapes");
    if (a->nb[0] != (a->type == 0 ? 4 : 2) || b->nb[0] != 4) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_mul_mat_f: the k dimension must be contiguous in both operands");
    const int64_t n_out = nelements(dst);
    if (n_out == 0) return MI355Q_OK;
    // many src1 rows and an f16 src0 (prefill attention): matrix cores
    if (a->type == 1 && b->ne[1] >= 16 && dst->ne[0] >= 16 && a->ne[0] % 16 == 0 && dst->ne[2] * dst->ne[3] <= 65535 &&
        !(((uintptr_t) a->data | (uintptr_t) a->nb[1] | (uintptr_t) a->nb[2] | (uintptr_t) a->nb[3] | (uintptr_t) b->data | (uintptr_t) b->nb[1] | (uintptr_t) b->nb[2] | (uintptr_t) b->nb[3]) & 15)) {
        hipLaunchKernelGGL(k_mul_mat_f16_mfma<false>, dim3((unsigned) ((dst->ne[0] + MMF_T - 1) / MMF_T), (unsigned) ((dst->ne[1] + MMF_T - 1) / MMF_T), (unsigned) (dst->ne[2] * dst->ne[3])),
                           dim3(256), 0, (hipStream_t) stream, to_d(a), to_d(b), to_d(dst));
        OPS_LAUNCHED();
    }
    const int64_t groups_per_col = (dst->ne[0] + MMF_ROWS - 1) / MMF_ROWS, n_groups = groups_per_col * dst->ne[1] * dst->ne[2] * dst->ne[3];
    if ((n_groups + 3) / 4 > 0x7FFFFFFF) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_mul_mat_f: too many outputs");
    hipLaunchKernelGGL(k_mul_mat_f, dim3((unsigned) ((n_groups + 3) / 4)), dim3(256), 0, (hipStream_t) stream, to_d(a), to_d(b), to_d(dst), n_groups, groups_per_col);
    OPS_LAUNCHED();
}

size_t mi355q_op_flash_attn_ext_workspace(int64_t dv, int64_t n_q, int64_t n_head, int64_t n_batch, int64_t n_kv) {
    if (n_q >= 16) return (size_t) (n_q * n_kv * n_head * n_batch * 4) + 256 + (size_t) (4 * n_kv * n_head * n_batch * (dv > 256 ? dv : 256));     // prefill: the scores of the batch (+ an f16 copy of a quantized K / V cache, at most n_head kv heads of 256)
    if (n_q * n_head * n_batch > 256) return 0;
    return (size_t) (8 * n_q * n_head * n_batch * (dv + 2) * 4);            // up to 8 pieces per row
}

int mi355q_op_flash_attn_ext(const mi355q_tensor * q, const mi355q_tensor * k, const mi355q_tensor * v, const mi355q_tensor * mask,
                             const mi355q_tensor * dst, float scale, float max_bias, float logit_softcap,
                             void * workspace, size_t workspace_bytes, void * stream) {
    if (!q || !k || !v || !dst) OPS_FAIL(MI355Q_ERR_SHAPE, "op_flash_attn_ext: null tensor");
    const bool kv_q80 = (k->type == MI355Q_TYPE_Q8_0 && v->type == MI355Q_TYPE_Q8_0) || (k->type == MI355Q_TYPE_Q4_0 && v->type == MI355Q_TYPE_Q4_0);     // a quantized cache
    const int  kv_bb = k->type == MI355Q_TYPE_Q4_0 ? 18 : 34;
    if (q->type != 0 || !((k->type == 1 && v->type == 1) || kv_q80) || dst->type != 0 || (mask && mask->type != 1)) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_flash_attn_ext: q f32, k / v both f16, both q8_0 or both q4_0, mask f16, dst f32");
    const int64_t DK = k->ne[0], DV = v->ne[0], n_kv = k->ne[1], N = q->ne[1], n_head = q->ne[2], nb3 = q->ne[3];
    if (q->ne[0] != DK || v->ne[1] != n_kv || k->ne[2] <= 0 || v->ne[2] <= 0 || n_head % k->ne[2] || n_head % v->ne[2] || k->ne[3] <= 0 || v->ne[3] <= 0 ||
        nb3 % k->ne[3] || nb3 % v->ne[3] || dst->ne[0] != DV || dst->ne[1] != n_head || dst->ne[2] != N || dst->ne[3] != nb3)
        OPS_FAIL(MI355Q_ERR_SHAPE, "op_flash_attn_ext: q [DK, N, H, B], k [DK, n_kv, Hk, Bk], v [DV, n_kv, Hv, Bv], dst [DV, H, N, B]");
    if (q->nb[0] != 4 || k->nb[0] != (kv_q80 ? kv_bb : 2) || v->nb[0] != (kv_q80 ? kv_bb : 2) || dst->nb[0] != 4) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_flash_attn_ext: innermost dimensions must be contiguous");
    if (mask && (mask->ne[0] < n_kv || mask->ne[1] < N || mask->nb[0] != 2)) OPS_FAIL(MI355Q_ERR_SHAPE, "op_flash_attn_ext: mask f16 [>= n_kv, >= N]");
    if (DK > 256 || DV > 256 || DK < 1 || DV < 1) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_flash_attn_ext: head sizes up to 256");
    if (n_kv > 36864) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_flash_attn_ext: n_kv > 36864 (one row of scores lives in LDS)");
    if (N == 0 || n_head == 0 || nb3 == 0) return MI355Q_OK;
    if (n_kv == 0) OPS_FAIL(MI355Q_ERR_SHAPE, "op_flash_attn_ext: empty KV window");
    if (n_head > 65535 || nb3 > 65535) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_flash_attn_ext: too many heads / batches");
    if (logit_softcap != 0.0f) scale /= logit_softcap;                          // ops.cpp:6757-6759
    uint32_t n_head_log2 = 1; while (2 * n_head_log2 <= (uint32_t) n_head) n_head_log2 *= 2;
    const float m0 = powf(2.0f, -(max_bias) / n_head_log2), m1 = powf(2.0f, -(max_bias / 2.0f) / n_head_log2);
    mi355q_tensor k16, v16;                                   // (a quantized cache under a prefill batch: its f16 copy)
    if (kv_q80 && N >= 16 && logit_softcap == 0.0f && workspace && DK % 32 == 0 && DV % 32 == 0) {
        // Many query rows: one workgroup per row re-streams (and re-decodes) the cache for every row.  The cache is decoded ONCE into scratch as f16 and the
        // matrix-core path below takes it (q in f16 against the dequantized K instead of the CPU's Q8_0-quantized q: ~1e-3 of a score, NMSE 1e-6 of the
        // output; the decode path, N < 16, keeps the CPU's exact arithmetic).
        const size_t sc_bytes = ((size_t) N * (size_t) n_kv * (size_t) n_head * (size_t) nb3 * 4 + 255) & ~(size_t) 255;
        const size_t kb16 = (size_t) DK * n_kv * k->ne[2] * k->ne[3] * 2, vb16 = (size_t) DV * n_kv * v->ne[2] * v->ne[3] * 2;
        if (workspace_bytes >= sc_bytes + ((kb16 + 255) & ~(size_t) 255) + vb16) {
            __half * kp = (__half *) ((char *) workspace + sc_bytes), * vp = (__half *) ((char *) kp + ((kb16 + 255) & ~(size_t) 255));
            if (k->type == MI355Q_TYPE_Q8_0) {
                hipLaunchKernelGGL(k_dequant_cache_f16<MI355Q_TYPE_Q8_0>, dim3(grid_for((int64_t) (kb16 / 2))), dim3(256), 0, (hipStream_t) stream, to_d(k), kp);
                hipLaunchKernelGGL(k_dequant_cache_f16<MI355Q_TYPE_Q8_0>, dim3(grid_for((int64_t) (vb16 / 2))), dim3(256), 0, (hipStream_t) stream, to_d(v), vp);
            } else {
                hipLaunchKernelGGL(k_dequant_cache_f16<MI355Q_TYPE_Q4_0>, dim3(grid_for((int64_t) (kb16 / 2))), dim3(256), 0, (hipStream_t) stream, to_d(k), kp);
                hipLaunchKernelGGL(k_dequant_cache_f16<MI355Q_TYPE_Q4_0>, dim3(grid_for((int64_t) (vb16 / 2))), dim3(256), 0, (hipStream_t) stream, to_d(v), vp);
            }
            auto as16 = [](const mi355q_tensor * t, __half * p, mi355q_tensor & o) {
                o = *t; o.data = p; o.type = 1; o.nb[0] = 2; o.nb[1] = 2 * t->ne[0]; o.nb[2] = o.nb[1] * t->ne[1]; o.nb[3] = o.nb[2] * t->ne[2];
            };
            as16(k, kp, k16); as16(v, vp, v16);
            return mi355q_op_flash_attn_ext(q, &k16, &v16, mask, dst, logit_softcap != 0.0f ? scale * logit_softcap : scale, max_bias, logit_softcap, workspace, sc_bytes, stream);
        }
    }
    if (kv_q80) {                                            // a quantized cache: one workgroup per row with the CPU's arithmetic (k_flash_attn_ext_q80)
        if (DK % 32 || DV % 32 || n_kv > 8192) OPS_FAIL(MI355Q_ERR_UNSUPPORTED, "op_flash_attn_ext: q8_0 cache needs head sizes of whole 32-blocks and n_kv <= 8192");
        const size_t lds = (size_t) (2 * n_kv + DK / 32 + DK / 4) * 4;
        static bool attr_q80[64] = {};
        if (lds > 48 * 1024) {
            int dev = 0; (void) hipGetDevice(&dev); dev = dev >= 0 && dev < 64 ? dev : 0;
            if (!attr_q80[dev]) {
                if (hipFuncSetAttribute((const void *) k_flash_attn_ext_q80<MI355Q_TYPE_Q8_0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
                    hipFuncSetAttribute((const void *) k_flash_attn_ext_q80<MI355Q_TYPE_Q4_0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) OPS_FAIL(MI355Q_ERR_HIP, "op_flash_attn_ext: LDS attribute");
                attr_q80[dev] = true;
            }
        }
        if (k->type == MI355Q_TYPE_Q8_0)
            hipLaunchKernelGGL(k_flash_attn_ext_q80<MI355Q_TYPE_Q8_0>, dim3((unsigned) N, (unsigned) n_head, (unsigned) nb3), dim3(FA_THREADS), lds, (hipStream_t) stream,
                               to_d(q), to_d(k), to_d(v), mask ? to_d(mask) : to_d(q), mask ? 1 : 0, to_d(dst), scale, max_bias, logit_softcap, m0, m1, n_head_log2);
        else
            hipLaunchKernelGGL(k_flash_attn_ext_q80<MI355Q_TYPE_Q4_0>, dim3((unsigned) N, (unsigned) n_head, (unsigned) nb3), dim3(FA_THREADS), lds, (hipStream_t) stream,
                               to_d(q), to_d(k), to_d(v), mask ? to_d(mask) : to_d(q), mask ? 1 : 0, to_d(dst), scale, max_bias, logit_softcap, m0, m1, n_head_log2);
        OPS_LAUNCHED();
    }
    // Many query rows (prefill) and scratch for the scores: the three steps of the definition on the matrix-core kernels -- scores = K q
    // (k_mul_mat_f16_mfma), row softmax with the f16 mask (k_soft_max), out = P V with V read transposed (k_mul_mat_f16_mfma<true>).  One
    // workgroup per row (below) re-streams K and V for every row: 5x slower for a 512-token batch.
    {
        const size_t need = (size_t) N * (size_t) n_kv * (size_t) n_head * (size_t) nb3 * 4;
        auto al16 = [](const mi355q_tensor * t) { return ((((uintptr_t) t->data) | (uintptr_t) t->nb[1] | (uintptr_t) t->nb[2] | (uintptr_t) t->nb[3]) & 15) == 0; };
        if (N >= 16 && logit_softcap == 0.0f && workspace && workspace_bytes >= need && DK % 16 == 0 && DV % 16 == 0 && n_kv % 16 == 0 &&
            al16(q) && al16(k) && al16(v) && n_head * nb3 <= 65535 && (!mask || mask->nb[1] == 2 * n_kv)) {
            mi355q_tensor sc; sc.data = workspace; sc.type = 0;
            sc.ne[0] = n_kv; sc.ne[1] = N; sc.ne[2] = n_head; sc.ne[3] = nb3;
            sc.nb[0] = 4; sc.nb[1] = 4 * n_kv; sc.nb[2] = sc.nb[1] * N; sc.nb[3] = sc.nb[2] * n_head;
            const dim3 g1((unsigned) ((n_kv + MMF_T - 1) / MMF_T), (unsigned) ((N + MMF_T - 1) / MMF_T), (unsigned) (n_head * nb3));
            hipLaunchKernelGGL(k_mul_mat_f16_mfma<false>, g1, dim3(256), 0, (hipStream_t) stream, to_d(k), to_d(q), to_d(&sc));
            const int rc = mi355q_op_soft_max(&sc, mask, &sc, scale, max_bias, stream);     // in place, row by row
            if (rc != MI355Q_OK) return rc;
            mi355q_tensor vt = *v;                             // [K = n_kv, M = DV] with M contiguous: the transposed-source form
            vt.ne[0] = n_kv; vt.ne[1] = DV;
            mi355q_tensor od = *dst;                           // out[n][m]: dst is [DV, H, N, B] -> rows n with stride nb[2], heads with nb[1]
            od.ne[0] = DV; od.ne[1] = N; od.ne[2] = n_head; od.ne[3] = nb3;
            od.nb[1] = dst->nb[2]; od.nb[2] = dst->nb[1];
            const dim3 g2((unsigned) ((DV + MMF_T - 1) / MMF_T), (unsigned) ((N + MMF_T - 1) / MMF_T), (unsigned) (n_head * nb3));
            hipLaunchKernelGGL(k_mul_mat_f16_mfma<true>, g2, dim3(256), 0, (hipStream_t) stream, to_d(&vt), to_d(&sc), to_d(&od));
            OPS_LAUNCHED();
        }
    }
    // few rows (decode): split the KV range so that the chip is not left to n_head workgroups with long dependent chains -- unless the window is short
    // enough for ONE workgroup per row to follow the CPU's sequential F16 accumulation (k_flash_attn_ext, seq): then the result has the CPU's bits
    // (MI355Q_FA_EXACT=0 keeps the f32 form)
    static const bool fa_exact = !(getenv("MI355Q_FA_EXACT") && atoi(getenv("MI355Q_FA_EXACT")) == 0);
    const int seq = fa_exact && n_kv <= 1024 && DK % 32 == 0 ? 1 : 0;
    int n_split = 1;
    if (!seq && N * n_head * nb3 <= 256 && n_kv >= 256) { n_split = (int) ((n_kv + 127) / 128); if (n_split > 8) n_split = 8; }
    if (n_split > 1 && (!workspace || workspace_bytes < (size_t) (n_split * N * n_head * nb3 * (DV + 2) * 4))) n_split = 1;
    const int64_t chunk = (n_kv + n_split - 1) / n_split;
    const size_t lds = (size_t) (chunk > FA_THREADS ? chunk : FA_THREADS) * 4 + (seq ? (size_t) n_kv * 4 : 0);
    static bool attr_set[64] = {};
    if (lds > 48 * 1024) {
        int dev = 0; (void) hipGetDevice(&dev); dev = dev >= 0 && dev < 64 ? dev : 0;
        if (!attr_set[dev]) {
            if (hipFuncSetAttribute((const void *) k_flash_attn_ext, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) OPS_FAIL(MI355Q_ERR_HIP, "op_flash_attn_ext: LDS attribute");
            attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL(k_flash_attn_ext, dim3((unsigned) (N * n_split), (unsigned) n_head, (unsigned) nb3), dim3(FA_THREADS), lds, (hipStream_t) stream,
                       to_d(q), to_d(k), to_d(v), mask ? to_d(mask) : to_d(q), mask ? 1 : 0, to_d(dst), scale, max_bias, logit_softcap, m0, m1, n_head_log2,
                       n_split, (float *) workspace, seq);
    if (n_split > 1)
        hipLaunchKernelGGL(k_flash_attn_combine, dim3((unsigned) N, (unsigned) n_head, (unsigned) nb3), dim3(256), 0, (hipStream_t) stream,
                           (const float *) workspace, n_split, DV, N, to_d(dst));
    OPS_LAUNCHED();
}

} // extern "C"
