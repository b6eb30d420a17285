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

// nearest_int(): round-half-even through the 1.5*2^23 constant (ggml-quants.c:372-377):
//   t = v + 12582912.f ; (bits(t) & 0x007fffff) - 0x00400000
// For |v| < 2^22 (the reference asserts it) t lies in [2^23, 2^24), its exponent field is fixed, and the
// masked expression equals bits(t) - bits(12582912.f) -- one integer subtract.
__device__ __forceinline__ int nearest_int_magic(float v) {
    return (int) (__float_as_uint(__fadd_rn(v, 12582912.0f)) - 0x4B400000u);
}

// four ints in [-128,127] -> packed bytes (a = byte 0)
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) {
    const uint32_t lo = (uint32_t) (a & 0xFFFF) | ((uint32_t) b << 16);        // i16 pairs
    const uint32_t hi = (uint32_t) (c & 0xFFFF) | ((uint32_t) d << 16);
    return __builtin_amdgcn_perm(hi, lo, 0x06040200u);                          // bytes 0,2 of lo then 0,2 of hi
}

// ---- Q8_K: whole wave = one block ------------------------------------------------------------
// in : v = this lane's 4 activations.   out: q = 4 packed int8, d = block scale (all lanes),
//      bsum = sum of the 16 quants of this lane's 16-group (valid in all 4 lanes of the group).
// Written for instruction count: every workgroup of a GEMV launch runs this for the whole activation
// vector, so it is executed (#CUs x K/256) times per launch.
__device__ __forceinline__ void q8k_wave(const float4 v, uint32_t & q, float & d, int & bsum) {
    const float ax = fabsf(v.x), ay = fabsf(v.y), az = fabsf(v.z), aw = fabsf(v.w);
    const float amax = fmaxf(fmaxf(ax, ay), fmaxf(az, aw));
    // wave maximum of |x| (non-negative floats order like their bit patterns), then the LOWEST lane that
    // holds it (ballot + find-first-set) and, inside that lane, the first of its 4 elements with that
    // magnitude = the first element in memory order, as the CPU loop "if (ax > amax)" picks it
    const uint32_t wbits = wave_max_u32(__float_as_uint(amax));
    const unsigned long long holders = __ballot(__float_as_uint(amax) == wbits);
    const int   src  = __builtin_amdgcn_readfirstlane(__ffsll((long long) holders) - 1);
    const float mine = ax == amax ? v.x : (ay == amax ? v.y : (az == amax ? v.z : v.w));
    const float vmax = readlane_f(mine, src);
    if (wbits == 0u) {                         // all-zero block: d = 0, quants 0 (bsums undefined in the reference; 0 here)
        q = 0; d = 0.0f; bsum = 0;
        return;
    }
    const float iscale = __fdiv_rn(-127.0f, vmax);
    const int q0 = min(127, nearest_int_magic(__fmul_rn(iscale, v.x)));
    const int q1 = min(127, nearest_int_magic(__fmul_rn(iscale, v.y)));
    const int q2 = min(127, nearest_int_magic(__fmul_rn(iscale, v.z)));
    const int q3 = min(127, nearest_int_magic(__fmul_rn(iscale, v.w)));
    q = pack4(q0, q1, q2, q3);
    bsum = quad_sum((q0 + q1) + (q2 + q3));
    d = __fdiv_rn(1.0f, iscale);
}

// ---- Q8_0 / Q8_1: 8 lanes = one 32-element block ---------------------------------------------
// out: q = 4 packed int8, d = f32 scale BEFORE f16 rounding (the quants use this one, as the CPU does),
//      sum = sum of the 32 quants of the block (valid in all 8 lanes).
template <bool ROUND_EVEN>
__device__ __forceinline__ void q80_group8(const float4 v, uint32_t & q, float & d, int & sum) {
    const float amax = oct_max(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    d = __fdiv_rn(amax, 127.0f);
    const float id = d != 0.0f ? __fdiv_rn(1.0f, d) : 0.0f;
    auto rnd = [](float p) { return (int) (ROUND_EVEN ? rintf(p) : roundf(p)); };
    const int q0 = rnd(__fmul_rn(v.x, id)), q1 = rnd(__fmul_rn(v.y, id)), q2 = rnd(__fmul_rn(v.z, id)), q3 = rnd(__fmul_rn(v.w, id));
    q = pack4(q0, q1, q2, q3);
    sum = oct_sum((q0 + q1) + (q2 + q3));
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
