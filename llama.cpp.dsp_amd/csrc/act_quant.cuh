// act_quant.cuh -- wave-level activation quantizers, bit-identical to the reference CPU ones.
//
//   Q8_K : quantize_row_q8_K_ref   ggml/src/ggml-quants.c:2479-2516   (256 elements, f32 scale, bsums)
//   Q8_0 : quantize_row_q8_0_ref   ggml/src/ggml-quants.c:194-217     (32 elements, f16 scale)
//   Q8_1 : quantize_row_q8_1_ref   ggml/src/ggml-quants.c:220-253     (32 elements, f16 scale + f16 d*sum)
//
// Mapping (wave64): one wave owns 256 consecutive activations, lane l owns x[4l .. 4l+3] (one float4,
// a fully coalesced 1 KiB wave load).  A Q8_K block is the whole wave; a 32-element block is 8 lanes.
// Every f32 operation that the CPU performs is done with the same operands and explicit
// round-to-nearest intrinsics (no FMA contraction), so blocks come out bit-for-bit equal.
#pragma once
#include "mi355q_common.h"

namespace mi355q {

// nearest_int(): round-half-even through the 1.5*2^23 constant (ggml-quants.c:372-377)
__device__ __forceinline__ int nearest_int_magic(float v) {
    const float t = __fadd_rn(v, 12582912.0f);
    return (int) ((__float_as_uint(t) & 0x007FFFFFu)) - 0x00400000;
}

__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) {
    return (uint32_t) (a & 0xFF) | ((uint32_t) (b & 0xFF) << 8) | ((uint32_t) (c & 0xFF) << 16) | ((uint32_t) (d & 0xFF) << 24);
}

// ---- Q8_K: whole wave = one block ------------------------------------------------------------
// in : v = this lane's 4 activations.   out: q = 4 packed int8, d = block scale (all lanes),
//      bsum = sum of the 16 quants of this lane's 16-group (valid in all 4 lanes of the group).
__device__ __forceinline__ void q8k_wave(const float4 v, uint32_t & q, float & d, int & bsum) {
    const int lane = lane_id();
    // signed value of the FIRST element with the largest magnitude ("if (ax > amax)")
    const float xs[4] = { v.x, v.y, v.z, v.w };
    float amax = 0.0f; int best = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float a = fabsf(xs[i]); if (a > amax) { amax = a; best = i; } }
    // key: larger |x| wins, then the smaller element index
    uint32_t khi = __float_as_uint(amax);
    uint32_t klo = 0xFFFFFFFFu - (uint32_t) (4 * lane + best);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t ohi = __shfl_xor(khi, o, 64);
        const uint32_t olo = __shfl_xor(klo, o, 64);
        const bool take = (ohi > khi) || (ohi == khi && olo > klo);
        khi = take ? ohi : khi;
        klo = take ? olo : klo;
    }
    const float wamax = __uint_as_float(khi);
    const int   widx  = (int) (0xFFFFFFFFu - klo);
    const int   wsub  = widx & 3;
    const float mine  = wsub == 0 ? v.x : (wsub == 1 ? v.y : (wsub == 2 ? v.z : v.w));
    const float vmax  = __shfl(mine, widx >> 2, 64);
    if (wamax == 0.0f) {                       // all-zero block: d = 0, quants 0 (bsums undefined in the reference; 0 here)
        q = 0; d = 0.0f; bsum = 0;
        return;
    }
    const float iscale = __fdiv_rn(-127.0f, vmax);
    int qi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = nearest_int_magic(__fmul_rn(iscale, xs[i]));
        qi[i] = t > 127 ? 127 : t;
    }
    q = pack4(qi[0], qi[1], qi[2], qi[3]);
    int s = qi[0] + qi[1] + qi[2] + qi[3];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    bsum = s;
    d = __fdiv_rn(1.0f, iscale);
}

// ---- Q8_0 / Q8_1: 8 lanes = one 32-element block ---------------------------------------------
// out: q = 4 packed int8, d = f32 scale BEFORE f16 rounding (the quants use this one, as the CPU does),
//      sum = sum of the 32 quants of the block (valid in all 8 lanes).
template <bool ROUND_EVEN>
__device__ __forceinline__ void q80_group8(const float4 v, uint32_t & q, float & d, int & sum) {
    float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    d = __fdiv_rn(amax, 127.0f);
    const float id = d != 0.0f ? __fdiv_rn(1.0f, d) : 0.0f;
    const float xs[4] = { v.x, v.y, v.z, v.w };
    int qi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float p = __fmul_rn(xs[i], id);
        qi[i] = (int) (ROUND_EVEN ? rintf(p) : roundf(p));
    }
    q = pack4(qi[0], qi[1], qi[2], qi[3]);
    int s = qi[0] + qi[1] + qi[2] + qi[3];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    sum = s;
}

// float4 load of this lane's 4 activations (zero beyond k).  `vec` = row is 16-byte aligned.
__device__ __forceinline__ float4 load_x4(const float * row, int64_t e0, int64_t k, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e0 + 3 < k) {
        if (vec) v = *(const float4 *) (row + e0);
        else { v.x = row[e0]; v.y = row[e0 + 1]; v.z = row[e0 + 2]; v.w = row[e0 + 3]; }
    }
    return v;
}

} // namespace mi355q
