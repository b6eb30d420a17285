// mi355q_common.h -- shared host/device definitions for libmi355q (gfx950 only).
//
// Block byte layouts follow the reference's format spec, ggml/src/ggml-common.h:167-418
// (SURVEY.md 8a rows a1-a7).  Nothing here is taken from ggml-cuda: the device layout, the lane
// mappings and the kernels are designed for 64-wide wavefronts and 16-byte-per-lane global loads.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/mi355q.h"

namespace mi355q {

constexpr int WAVE = 64;

// ------------------------------------------------------------------------------------------------
// Type table.  A canonical ggml block of `bsize` bytes is cut into up to 4 "planes"
// (byte ranges).  In the PLANAR device layout a row of nb blocks stores plane 0 of all nb blocks,
// then plane 1 of all blocks, ... so that every plane is a dense, 16-byte-aligned array a wave can
// read with one global_load_dwordx4 per lane.  Row byte size and row stride equal the canonical ones.
// ------------------------------------------------------------------------------------------------
struct Plane { int src_off; int bytes; };

struct TypeInfo {
    int   type;        // ggml type id
    int   blck;        // elements per block
    int   bsize;       // bytes per block
    int   act;         // activation format the CPU pairs with it (vec_dot_type, ggml-cpu.c:211-376)
    int   nplanes;
    Plane planes[4];   // device order
    int   fast;        // 1: a planar GEMV kernel exists
};

inline const TypeInfo * type_info(int type) {   // host only
    // planes: payload first (always 16-B aligned for a 16-B aligned row), small fields after
    static const TypeInfo tab[] = {
        { MI355Q_TYPE_Q4_0,   32,  18, MI355Q_TYPE_Q8_0, 2, {{2, 16}, {0, 2}, {0, 0}, {0, 0}},            1 },
        { MI355Q_TYPE_Q4_1,   32,  20, MI355Q_TYPE_Q8_1, 2, {{4, 16}, {0, 4}, {0, 0}, {0, 0}},            0 },
        { MI355Q_TYPE_Q5_0,   32,  22, MI355Q_TYPE_Q8_0, 3, {{6, 16}, {2, 4}, {0, 2}, {0, 0}},            0 },
        { MI355Q_TYPE_Q5_1,   32,  24, MI355Q_TYPE_Q8_1, 3, {{8, 16}, {4, 4}, {0, 4}, {0, 0}},            0 },
        { MI355Q_TYPE_Q8_0,   32,  34, MI355Q_TYPE_Q8_0, 2, {{2, 32}, {0, 2}, {0, 0}, {0, 0}},            1 },
        { MI355Q_TYPE_Q2_K,   256, 84, MI355Q_TYPE_Q8_K, 3, {{16, 64}, {0, 16}, {80, 4}, {0, 0}},         0 },
        { MI355Q_TYPE_Q3_K,   256, 110, MI355Q_TYPE_Q8_K, 4, {{32, 64}, {0, 32}, {96, 12}, {108, 2}},     0 },
        { MI355Q_TYPE_Q4_K,   256, 144, MI355Q_TYPE_Q8_K, 2, {{16, 128}, {0, 16}, {0, 0}, {0, 0}},        1 },
        { MI355Q_TYPE_Q5_K,   256, 176, MI355Q_TYPE_Q8_K, 3, {{48, 128}, {16, 32}, {0, 16}, {0, 0}},      1 },
        { MI355Q_TYPE_Q6_K,   256, 210, MI355Q_TYPE_Q8_K, 4, {{0, 128}, {128, 64}, {192, 16}, {208, 2}},  1 },
        { MI355Q_TYPE_IQ4_NL, 32,  18, MI355Q_TYPE_Q8_0, 2, {{2, 16}, {0, 2}, {0, 0}, {0, 0}},            1 },
        { MI355Q_TYPE_IQ4_XS, 256, 136, MI355Q_TYPE_Q8_K, 2, {{8, 128}, {0, 8}, {0, 0}, {0, 0}},          1 },
        // code-book formats: canonical layout, generic tier (code books: iq_tables.h)
        { MI355Q_TYPE_IQ2_XXS, 256, 66,  MI355Q_TYPE_Q8_K, 1, {{0, 66}, {0, 0}, {0, 0}, {0, 0}},          0 },
        { MI355Q_TYPE_IQ2_XS,  256, 74,  MI355Q_TYPE_Q8_K, 1, {{0, 74}, {0, 0}, {0, 0}, {0, 0}},          0 },
        { MI355Q_TYPE_IQ2_S,   256, 82,  MI355Q_TYPE_Q8_K, 1, {{0, 82}, {0, 0}, {0, 0}, {0, 0}},          0 },
        { MI355Q_TYPE_IQ3_XXS, 256, 98,  MI355Q_TYPE_Q8_K, 1, {{0, 98}, {0, 0}, {0, 0}, {0, 0}},          0 },
        { MI355Q_TYPE_IQ3_S,   256, 110, MI355Q_TYPE_Q8_K, 1, {{0, 110}, {0, 0}, {0, 0}, {0, 0}},         0 },
        { MI355Q_TYPE_IQ1_S,   256, 50,  MI355Q_TYPE_Q8_K, 1, {{0, 50}, {0, 0}, {0, 0}, {0, 0}},          0 },
        { MI355Q_TYPE_IQ1_M,   256, 56,  MI355Q_TYPE_Q8_K, 1, {{0, 56}, {0, 0}, {0, 0}, {0, 0}},          0 },
        // activation-only formats (never src0)
        { MI355Q_TYPE_Q8_1,   32,  36, -1, 0, {{0, 0}, {0, 0}, {0, 0}, {0, 0}},                            0 },
        { MI355Q_TYPE_Q8_K,   256, 292, -1, 0, {{0, 0}, {0, 0}, {0, 0}, {0, 0}},                           0 },
    };
    for (unsigned i = 0; i < sizeof(tab) / sizeof(tab[0]); ++i)
        if (tab[i].type == type) return &tab[i];
    return nullptr;
}

// Planar layout is used iff a fast kernel exists AND every row / plane starts 16-byte aligned.
inline bool is_planar(const TypeInfo * ti, int64_t k) {   // host only
    if (!ti || !ti->fast || k % ti->blck) return false;
    const int64_t nb = k / ti->blck;
    if ((nb * ti->bsize) % 16) return false;
    int64_t off = 0;
    for (int p = 0; p < ti->nplanes; ++p) {
        if (off % 16) return false;
        off += (int64_t) ti->planes[p].bytes * nb;
    }
    return true;
}

// Grouped MUL_MAT_ID on the matrix-core tiers: the (token, slot) pairs are sorted by expert ON THE DEVICE into segments that start at
// multiples of the token-tile size; token tile t of the gathered rows belongs to expert tile_expert[t] (-1: padding beyond the last segment).
// A tile's workgroup takes the expert's matrix (w + e * expert_stride) and stores only the rows below seg_end[e].  tile_expert == nullptr:
// an ordinary dense matmul.
struct MoeTiles {
    const int32_t* tile_expert; const int32_t* seg_end; int64_t expert_stride; int tile_tokens; int pad;
    const int32_t* dst_row;      // slot -> row of y the result belongs to (-1: padding); nullptr: rows are stored in slot order
};

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
#ifdef __HIPCC__

__device__ __forceinline__ float h2f(uint32_t h16) {
    return __half2float(__ushort_as_half((unsigned short) h16));
}

// signed 4 x int8 dot-accumulate (v_dot4_i32_i8)
__device__ __forceinline__ int dot4(int a, int b, int c) {
    return __builtin_amdgcn_sdot4(a, b, c, false);
}

__device__ __forceinline__ int lane_id() { return (int) (threadIdx.x & 63); }

// Loads of WEIGHT bytes name the global address space.  A pointer that arrives through a descriptor is generic to the compiler, and a generic
// pointer costs a FLAT instruction: it may address LDS, so it counts on vmcnt AND lgkmcnt and can complete out of order with either queue -- with a
// single flat access pending the compiler answers every dependency with s_waitcnt vmcnt(0) lgkmcnt(0).  In a software-pipelined stream (8 loads in
// flight per wave, consumed one at a time) that waits for the load issued LAST before each step: the ring degenerates to depth 1.
#define MI355Q_GLOBAL __attribute__((address_space(1)))
typedef unsigned int mi355q_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int mi355q_u2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ T ldg(const void * p) { return *(const MI355Q_GLOBAL T *) p; }
template <> __device__ __forceinline__ uint2 ldg<uint2>(const void * p) { const mi355q_u2 v = *(const MI355Q_GLOBAL mi355q_u2 *) p; return make_uint2(v.x, v.y); }
// 16-byte global load, non-temporal (weights are streamed exactly once: MI355X guide "nt-weights")
__device__ __forceinline__ uint4 ldg16_nt(const void * p) {
    const MI355Q_GLOBAL unsigned int * q = (const MI355Q_GLOBAL unsigned int *) p;
    uint4 r;
    r.x = __builtin_nontemporal_load(q);
    r.y = __builtin_nontemporal_load(q + 1);
    r.z = __builtin_nontemporal_load(q + 2);
    r.w = __builtin_nontemporal_load(q + 3);
    return r;
}
__device__ __forceinline__ uint4 ldg16(const void * p) { const mi355q_u4 v = *(const MI355Q_GLOBAL mi355q_u4 *) p; return make_uint4(v.x, v.y, v.z, v.w); }

// ---- cross-lane reductions on DPP (VALU data-parallel primitives), not ds_bpermute: a 64-lane
// __shfl_xor tree is ~6 dependent LDS-crossbar round trips (~0.4 us per reduction on gfx950).
//   quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// sum over all 64 lanes, fixed (deterministic) tree; result uniform
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);   // each 16-lane row holds its sum
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
// max over all 64 lanes of an unsigned key; result uniform
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    int x = (int) v;
    auto mx = [](int a, int b) { return (int) max((uint32_t) a, (uint32_t) b); };
    x = mx(x, dpp_i<0xB1>(x)); x = mx(x, dpp_i<0x4E>(x)); x = mx(x, dpp_i<0x141>(x)); x = mx(x, dpp_i<0x140>(x));
    const uint32_t r0 = (uint32_t) __builtin_amdgcn_readlane(x, 0),  r1 = (uint32_t) __builtin_amdgcn_readlane(x, 16);
    const uint32_t r2 = (uint32_t) __builtin_amdgcn_readlane(x, 32), r3 = (uint32_t) __builtin_amdgcn_readlane(x, 48);
    return max(max(r0, r1), max(r2, r3));
}
// integer sum over each aligned group of 4 / 8 lanes (result in every lane of the group)
__device__ __forceinline__ int quad_sum(int v) { v += dpp_i<0xB1>(v); v += dpp_i<0x4E>(v); return v; }
__device__ __forceinline__ int oct_sum(int v)  { v = quad_sum(v); v += dpp_i<0x141>(v); return v; }
__device__ __forceinline__ float oct_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); return v;
}
// value of the neighbouring lane (lane ^ 1)
__device__ __forceinline__ int pair_swap(int v) { return dpp_i<0xB1>(v); }

// expf as the C library of the reference's host computes it (glibc >= 2.27, sysdeps/ieee754/flt-32/e_expf.c): x / ln2 split into k / 32 + r, a 32-entry
// table of 2^(i/32) and a cubic in r, everything in f64, rounded once to f32 -- so an f64 restatement gives the library's bits (the device library's
// expf is a different algorithm, 1 ulp apart in ~10 % of the arguments: enough to flip an f16 rounding of the flash accumulator now and then).
// The table lives in the lanes of the calling wave (lane i holds entry i % 32: `tab`), read with ds_bpermute: no memory access on the serial path.
static __device__ const unsigned long long PLAN_EXP2F_T[32] = {      // bits(2^(i/32)) - (i << 47), correctly rounded (generated with 60-digit decimal arithmetic)
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull };
__device__ __forceinline__ float expf_libm(float x, unsigned long long tab) {      // (every lane of the wave must be active: ds_bpermute)
    const double xd = (double) x;
    const double z = 0x1.71547652b82fep+5 * xd;                 // 32 / ln 2
    double kd = z + 0x1.8p+52;
    const unsigned long long ki = (unsigned long long) __double_as_longlong(kd);
    kd -= 0x1.8p+52;
    const double r = z - kd;
    const int src = 4 * (int) (ki & 31ull);
    const unsigned lo = (unsigned) __builtin_amdgcn_ds_bpermute(src, (int) (unsigned) tab), hi = (unsigned) __builtin_amdgcn_ds_bpermute(src, (int) (unsigned) (tab >> 32));
    const unsigned long long t = (((unsigned long long) hi << 32) | lo) + (ki << 47);
    const double sd = __longlong_as_double((long long) t);
    const double zz = 0x1.c6af84b912394p-20 * r + 0x1.ebfce50fac4f3p-13;
    const double r2 = r * r;
    double y = 0x1.62e42ff0c52d6p-6 * r + 1.0;
    y = zz * r2 + y;
    y = y * sd;
    float res = (float) y;
    if (x < -0x1.9fe368p6f) res = 0.0f;                         // (underflow, -inf included)
    if (x > 0x1.62e42ep6f) res = INFINITY;
    if (x != x) res = x;
    return res;
}

// f32 -> f16 -> f32 of a value that was ROUNDED TO F32 FIRST.  Written plainly, fptrunc(fma(a, b, c)) is fused by the compiler into v_fma_mixlo_f16,
// which rounds the exact a * b + c to f16 ONCE; the CPU rounds to f32 and then to f16, and the two differ whenever the f32 result lands on an f16 tie
// (2^-13 of the operations: one or two elements of an attention output).  The empty asm pins the f32 value in a register between the two roundings.
__device__ __forceinline__ float f16_round(float x) { asm volatile("" : "+v"(x)); return __half2float(__float2half_rn(x)); }

// K-quant 6-bit (scale, min) pair j out of the 12-byte field given as three dwords
// (get_scale_min_k4, ggml/src/ggml-quants.c:631-638)
__device__ __forceinline__ uint32_t byte_of3(uint32_t w0, uint32_t w1, uint32_t w2, int idx) {
    const uint32_t w = idx < 4 ? w0 : (idx < 8 ? w1 : w2);
    return (w >> (8 * (idx & 3))) & 0xFFu;
}
__device__ __forceinline__ void k4_scale_min(uint32_t w0, uint32_t w1, uint32_t w2, int j, int & sc, int & mn) {
    if (j < 4) {
        sc = (int) (byte_of3(w0, w1, w2, j) & 63u);
        mn = (int) (byte_of3(w0, w1, w2, j + 4) & 63u);
    } else {
        const uint32_t a = byte_of3(w0, w1, w2, j + 4);
        sc = (int) ((a & 0x0Fu) | ((byte_of3(w0, w1, w2, j - 4) >> 6) << 4));
        mn = (int) ((a >> 4)    | ((byte_of3(w0, w1, w2, j)     >> 6) << 4));
    }
}

#endif // __HIPCC__

} // namespace mi355q
