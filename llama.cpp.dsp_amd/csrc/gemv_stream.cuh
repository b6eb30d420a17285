// gemv_stream.cuh -- device-side building blocks of the planar-row weight streamers (gemv_fast.hip: one
// launch per matmul; plan.hip: one persistent launch per token): the LDS image of the quantized
// activations, the in-flight payload chunk of each weight type and its consumer, and the fused
// activation-quantization prologue.
#pragma once
#include "act_quant.cuh"

namespace mi355q {

constexpr int GEMV_THREADS  = 1024;           // 16 waves = ONE workgroup per CU: the activation quantization (done by every
                                              // workgroup for itself) then runs once per CU instead of twice
constexpr int GEMV_WAVES    = GEMV_THREADS / WAVE;
constexpr int GEMV_MAX_MATS = 4;

// ------------------------------------------------------------------------------------------------
// LDS image of the quantized activations, per column n (all offsets in bytes from the column base)
//   Q8_K family: q8[k] | d f32 [k/256] | bsums i32 [k/16]
//   Q8_0 family: q8[k] | d f32 [k/32] (already f16-rounded) | sums i32 [k/32]
// ------------------------------------------------------------------------------------------------
// kernel families: which weight types one kernel instantiation can stream (they share the activation image)
//   FAM_Q8K : Q4_K, Q5_K, Q6_K, IQ4_XS (Q8_K activations)   FAM_Q80 : Q8_0, Q4_0, IQ4_NL (Q8_0 activations)
enum { FAM_Q8K = 0, FAM_Q80 = 1 };
__host__ __device__ constexpr bool fam_is_q8k(int fam) { return fam != FAM_Q80; }

__host__ __device__ __forceinline__ int lds_col_bytes(int fam, int k) {
    return fam_is_q8k(fam) ? k + (k / 256) * 4 + (k / 16) * 4 : k + (k / 32) * 8;
}

struct ActView {
    const uint8_t * base;   // column base in LDS
    int             k;
    __device__ __forceinline__ uint4 q16(int e) const { return *(const uint4 *) (base + e); }       // 16 int8, e % 16 == 0
    // Q8_K
    __device__ __forceinline__ float dK(int b) const { return *(const float *) (base + k + 4 * b); }
    __device__ __forceinline__ int   bsum(int g16) const { return *(const int *) (base + k + (k >> 6) + 4 * g16); }
    // Q8_0
    __device__ __forceinline__ float d0(int b) const { return *(const float *) (base + k + 4 * b); }
    __device__ __forceinline__ int   sum0(int b) const { return *(const int *) (base + k + (k >> 3) + 4 * b); }
};

__device__ __forceinline__ int dot16(const uint32_t w[4], const uint4 a) {
    int s = dot4((int) w[0], (int) a.x, 0);
    s = dot4((int) w[1], (int) a.y, s);
    s = dot4((int) w[2], (int) a.z, s);
    s = dot4((int) w[3], (int) a.w, s);
    return s;
}

// ------------------------------------------------------------------------------------------------
// One in-flight payload chunk (16 bytes of packed quants) plus the side fields its lane needs.
// chunk_load<T>() only issues global loads; Consume<T,N>::run() unpacks and accumulates.
// A row is cut into steps of 64 chunks; in step s lane l owns chunk c = 64*s + l.  `row` and `s` are
// wave-uniform (SGPRs); everything derived from the lane id alone is loop-invariant.
// ------------------------------------------------------------------------------------------------
struct Chunk {
    uint4    q;        // payload
    uint4    a;        // Q4_K/Q5_K: header (d, dmin, 12 scale bytes)   Q6_K: qh bytes
    uint4    b;        // Q5_K: qh bytes
    uint32_t sc;       // Q6_K: the two int8 sub-block scales (bytes 0 and 1)
    uint32_t dh;       // f16 super-scale (Q6_K, Q8_0, Q4_0)
};

template <int T> __device__ __forceinline__ void chunk_load(Chunk & ch, const uint8_t * row, int nb, int s, int lane);
template <int T, int NCOLS> struct Consume;

// K-quant 6-bit scale/min pairs of sub-blocks (2g, 2g+1) from the 12-byte field (w0,w1,w2), g lane-invariant.
//   j <  4: sc = q[j] & 63,                     m = q[j+4] & 63
//   j >= 4: sc = (q[j+4] & 15) | (q[j-4]>>6)<<4, m = (q[j+4] >> 4) | (q[j]>>6)<<4      (ggml-quants.c:631-638)
__device__ __forceinline__ void k4_pairs(uint32_t w0, uint32_t w1, uint32_t w2, int g, int & sc0, int & sc1, int & mn0, int & mn1) {
    const int sh = 16 * (g & 1);                                 // the pair sits in bytes (2g&3, 2g&3+1) of its dword
    const uint32_t x0 = (w0 >> sh) & 0xFFFFu, x1 = (w1 >> sh) & 0xFFFFu, x2 = (w2 >> sh) & 0xFFFFu;
    const uint32_t sc_lo = x0 & 0x3F3Fu, mn_lo = x1 & 0x3F3Fu;
    const uint32_t sc_hi = (x2 & 0x0F0Fu) | (((x0 >> 6) & 0x0303u) << 4);
    const uint32_t mn_hi = ((x2 >> 4) & 0x0F0Fu) | (((x1 >> 6) & 0x0303u) << 4);
    const uint32_t sc = g < 2 ? sc_lo : sc_hi, mn = g < 2 ? mn_lo : mn_hi;
    sc0 = (int) (sc & 0xFFu); sc1 = (int) (sc >> 8); mn0 = (int) (mn & 0xFFu); mn1 = (int) (mn >> 8);
}

// ---- Q4_K planar: [qs 128*nb][hdr(d,dmin,scales) 16*nb]                          ggml-common.h:285-296
template <> __device__ __forceinline__ void chunk_load<MI355Q_TYPE_Q4_K>(Chunk & ch, const uint8_t * row, int nb, int s, int lane) {
    ch.q = ldg16_nt(row + 1024 * s + 16 * lane);
    ch.a = ldg16(row + 128 * nb + 128 * s + 16 * (lane >> 3));
}
template <int NCOLS> __device__ __forceinline__ void consume_q4k_q5k(const uint32_t lo[4], const uint32_t hi[4], const uint4 h,
                                                                        int s, int lane, const ActView * av, float * acc) {
    const int g = (lane >> 1) & 3, half = lane & 1;
    const int b = 8 * s + (lane >> 3);
    const float d = h2f(h.x & 0xFFFFu), dmin = h2f(h.x >> 16);
    int sc0, mn0, sc1, mn1;
    k4_pairs(h.y, h.z, h.w, g, sc0, sc1, mn0, mn1);
    const int e  = 2048 * s + (256 * (lane >> 3) + 64 * g + 16 * half);    // low nibbles -> e.., high nibbles -> e+32..
    const int bi = 128 * s + (16 * (lane >> 3) + 4 * g + half);            // bsum index of the low half; high half = +2
#pragma unroll
    for (int n = 0; n < NCOLS; ++n) {
        const int s0 = dot16(lo, av[n].q16(e)), s1 = dot16(hi, av[n].q16(e + 32));
        const int m  = mn0 * av[n].bsum(bi) + mn1 * av[n].bsum(bi + 2);
        const float yd = av[n].dK(b);
        acc[n] += (d * yd) * (float) (sc0 * s0 + sc1 * s1) - (dmin * yd) * (float) m;
    }
}
template <int NCOLS> struct Consume<MI355Q_TYPE_Q4_K, NCOLS> {
    static __device__ __forceinline__ void run(const Chunk & ch, int s, int lane, const ActView * av, float * acc) {
        const uint32_t lo[4] = { ch.q.x & 0x0F0F0F0Fu, ch.q.y & 0x0F0F0F0Fu, ch.q.z & 0x0F0F0F0Fu, ch.q.w & 0x0F0F0F0Fu };
        const uint32_t hi[4] = { (ch.q.x >> 4) & 0x0F0F0F0Fu, (ch.q.y >> 4) & 0x0F0F0F0Fu, (ch.q.z >> 4) & 0x0F0F0F0Fu, (ch.q.w >> 4) & 0x0F0F0F0Fu };
        consume_q4k_q5k<NCOLS>(lo, hi, ch.a, s, lane, av, acc);
    }
};

// ---- Q5_K planar: [qs 128*nb][qh 32*nb][hdr 16*nb]                                ggml-common.h:302-314
template <> __device__ __forceinline__ void chunk_load<MI355Q_TYPE_Q5_K>(Chunk & ch, const uint8_t * row, int nb, int s, int lane) {
    ch.q = ldg16_nt(row + 1024 * s + 16 * lane);
    ch.b = ldg16(row + 128 * nb + 256 * s + (32 * (lane >> 3) + 16 * (lane & 1)));
    ch.a = ldg16(row + 160 * nb + 128 * s + 16 * (lane >> 3));
}
template <int NCOLS> struct Consume<MI355Q_TYPE_Q5_K, NCOLS> {
    static __device__ __forceinline__ void run(const Chunk & ch, int s, int lane, const ActView * av, float * acc) {
        const int g = (lane >> 1) & 3;
        const uint32_t qw[4] = { ch.q.x, ch.q.y, ch.q.z, ch.q.w }, hw[4] = { ch.b.x, ch.b.y, ch.b.z, ch.b.w };
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[i] = (qw[i] & 0x0F0F0F0Fu)        | (((hw[i] >> (2 * g))     & 0x01010101u) << 4);
            hi[i] = ((qw[i] >> 4) & 0x0F0F0F0Fu) | (((hw[i] >> (2 * g + 1)) & 0x01010101u) << 4);
        }
        consume_q4k_q5k<NCOLS>(lo, hi, ch.a, s, lane, av, acc);
    }
};

// ---- Q6_K planar: [ql 128*nb][qh 64*nb][scales 16*nb, PERMUTED][d 2*nb]           ggml-common.h:320-326
// chunk j = 4h + 2cc + p of a block holds ql bytes 16j..16j+15; its two sub-block scales
// (8h+2cc+p and 8h+4+2cc+p) are stored adjacently at bytes (2j, 2j+1) of the scale plane (see layout.hip).
template <> __device__ __forceinline__ void chunk_load<MI355Q_TYPE_Q6_K>(Chunk & ch, const uint8_t * row, int nb, int s, int lane) {
    const int j = lane & 7;
    ch.q  = ldg16_nt(row + 1024 * s + 16 * lane);
    ch.a  = ldg16_nt(row + 128 * nb + 512 * s + (64 * (lane >> 3) + 32 * (j >> 2) + 16 * (j & 1)));
    ch.sc = ldg<uint16_t>(row + 192 * nb + 128 * s + 2 * lane);
    ch.dh = ldg<uint16_t>(row + 208 * nb + 16 * s + 2 * (lane >> 3));
}
template <int NCOLS> struct Consume<MI355Q_TYPE_Q6_K, NCOLS> {
    static __device__ __forceinline__ void run(const Chunk & ch, int s, int lane, const ActView * av, float * acc) {
        const int j = lane & 7, cc = (j >> 1) & 1;
        const int b = 8 * s + (lane >> 3);
        const uint32_t lw[4] = { ch.q.x, ch.q.y, ch.q.z, ch.q.w }, hw[4] = { ch.a.x, ch.a.y, ch.a.z, ch.a.w };
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t hs = hw[i] >> (2 * cc);
            lo[i] = (lw[i] & 0x0F0F0F0Fu)        | ((hs & 0x03030303u) << 4);
            hi[i] = ((lw[i] >> 4) & 0x0F0F0F0Fu) | (hs & 0x30303030u);
        }
        const float d = h2f(ch.dh);
        const int sc0 = (int) (int8_t) (ch.sc & 0xFFu), sc1 = (int) (int8_t) (ch.sc >> 8);
        const int e  = 2048 * s + (256 * (lane >> 3) + 128 * (j >> 2) + 32 * cc + 16 * (j & 1));   // low -> e.., high -> e+64..
        const int bi = 128 * s + (16 * (lane >> 3) + 8 * (j >> 2) + 2 * cc + (j & 1));
#pragma unroll
        for (int n = 0; n < NCOLS; ++n) {
            // sum (q-32)*y = sum q*y - 32*sum y ; sum y over the 16 elements is exactly a Q8_K bsum
            const int s0 = dot16(lo, av[n].q16(e))      - 32 * av[n].bsum(bi);
            const int s1 = dot16(hi, av[n].q16(e + 64)) - 32 * av[n].bsum(bi + 4);
            acc[n] += (d * av[n].dK(b)) * (float) (sc0 * s0 + sc1 * s1);
        }
    }
};

// ---- Q8_0 planar: [qs 32*nb][d 2*nb]; a chunk is HALF a block                      ggml-common.h:209-214
template <> __device__ __forceinline__ void chunk_load<MI355Q_TYPE_Q8_0>(Chunk & ch, const uint8_t * row, int nb, int s, int lane) {
    ch.q  = ldg16_nt(row + 1024 * s + 16 * lane);
    ch.dh = ldg<uint16_t>(row + 32 * nb + 64 * s + 2 * (lane >> 1));
}
template <int NCOLS> struct Consume<MI355Q_TYPE_Q8_0, NCOLS> {
    static __device__ __forceinline__ void run(const Chunk & ch, int s, int lane, const ActView * av, float * acc) {
        const uint32_t w[4] = { ch.q.x, ch.q.y, ch.q.z, ch.q.w };
        const float dw = h2f(ch.dh);
        const int c = 64 * s + lane;
#pragma unroll
        for (int n = 0; n < NCOLS; ++n) {
            int t = dot16(w, av[n].q16(16 * c));
            t += pair_swap(t);                                // exact int32 sum of the whole 32-block, as the CPU forms it
            if ((lane & 1) == 0) acc[n] += (float) t * (dw * av[n].d0(c >> 1));
        }
    }
};

// ---- Q4_0 planar: [qs 16*nb][d 2*nb]; a chunk is one block                         ggml-common.h:167-172
template <> __device__ __forceinline__ void chunk_load<MI355Q_TYPE_Q4_0>(Chunk & ch, const uint8_t * row, int nb, int s, int lane) {
    ch.q  = ldg16_nt(row + 1024 * s + 16 * lane);
    ch.dh = ldg<uint16_t>(row + 16 * nb + 128 * s + 2 * lane);
}
template <int NCOLS> struct Consume<MI355Q_TYPE_Q4_0, NCOLS> {
    static __device__ __forceinline__ void run(const Chunk & ch, int s, int lane, const ActView * av, float * acc) {
        const uint32_t lo[4] = { ch.q.x & 0x0F0F0F0Fu, ch.q.y & 0x0F0F0F0Fu, ch.q.z & 0x0F0F0F0Fu, ch.q.w & 0x0F0F0F0Fu };
        const uint32_t hi[4] = { (ch.q.x >> 4) & 0x0F0F0F0Fu, (ch.q.y >> 4) & 0x0F0F0F0Fu, (ch.q.z >> 4) & 0x0F0F0F0Fu, (ch.q.w >> 4) & 0x0F0F0F0Fu };
        const float dw = h2f(ch.dh);
        const int c = 64 * s + lane;
#pragma unroll
        for (int n = 0; n < NCOLS; ++n) {
            // sum (q-8)*y = sum q*y - 8*sum y
            const int t = dot16(lo, av[n].q16(32 * c)) + dot16(hi, av[n].q16(32 * c + 16)) - 8 * av[n].sum0(c);
            acc[n] += (float) t * dw * av[n].d0(c);           // CPU order: sumi*d_x*d_y (ggml-cpu-quants.c:2604)
        }
    }
};

// ---- the non-linear 4-bit formats: a nibble indexes the 16-entry code book kvalues_iq4nl (ggml-common.h; decoded by
// dequantize_row_iq4_nl / _xs, ggml-quants.c:2434-2475).  Four nibbles held as bytes of a dword become four int8 with two v_perm_b32
// (each picks among 8 table bytes by the low 3 bits) and one select on bit 3 -- no LDS table, no per-nibble loads.
__device__ __forceinline__ uint32_t iq4_lut4(uint32_t nib) {
    // the code book {-127,-104,-83,-65 | -49,-35,-22,-10 | 1,13,25,38 | 53,69,89,113} as little-endian dwords
    const uint32_t sel = nib & 0x07070707u;
    const uint32_t a = __builtin_amdgcn_perm(0xF6EADDCFu, 0xBFAD9881u, sel);      // entries 0..7
    const uint32_t b = __builtin_amdgcn_perm(0x71594535u, 0x26190D01u, sel);      // entries 8..15
    const uint32_t m = ((nib >> 3) & 0x01010101u) * 0xFFu;
    return (a & ~m) | (b & m);
}

// ---- IQ4_NL planar: [qs 16*nb][d 2*nb]; a chunk is one 32-block (low nibbles = elements 0..15, high = 16..31)      ggml-common.h block_iq4_nl
template <> __device__ __forceinline__ void chunk_load<MI355Q_TYPE_IQ4_NL>(Chunk & ch, const uint8_t * row, int nb, int s, int lane) {
    ch.q  = ldg16_nt(row + 1024 * s + 16 * lane);
    ch.dh = ldg<uint16_t>(row + 16 * nb + 128 * s + 2 * lane);
}
template <int NCOLS> struct Consume<MI355Q_TYPE_IQ4_NL, NCOLS> {
    static __device__ __forceinline__ void run(const Chunk & ch, int s, int lane, const ActView * av, float * acc) {
        const uint32_t qw[4] = { ch.q.x, ch.q.y, ch.q.z, ch.q.w };
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { lo[i] = iq4_lut4(qw[i] & 0x0F0F0F0Fu); hi[i] = iq4_lut4((qw[i] >> 4) & 0x0F0F0F0Fu); }
        const float dw = h2f(ch.dh);
        const int c = 64 * s + lane;
#pragma unroll
        for (int n = 0; n < NCOLS; ++n) {
            const int t = dot16(lo, av[n].q16(32 * c)) + dot16(hi, av[n].q16(32 * c + 16));
            acc[n] += (av[n].d0(c) * dw) * (float) t;         // CPU: d = d_y * d_x; sumf += d * (sumi1 + sumi2)   (ggml-cpu-quants.c ggml_vec_dot_iq4_nl_q8_0)
        }
    }
};

// ---- IQ4_XS planar: [qs 128*nb][hdr(d, scales_h, scales_l[4]) 8*nb]; a chunk is one 32-element sub-block            ggml-common.h block_iq4_xs
template <> __device__ __forceinline__ void chunk_load<MI355Q_TYPE_IQ4_XS>(Chunk & ch, const uint8_t * row, int nb, int s, int lane) {
    ch.q = ldg16_nt(row + 1024 * s + 16 * lane);
    const uint2 h = ldg<uint2>(row + 128 * nb + 64 * s + 8 * (lane >> 3));
    ch.a.x = h.x; ch.a.y = h.y;
}
template <int NCOLS> struct Consume<MI355Q_TYPE_IQ4_XS, NCOLS> {
    static __device__ __forceinline__ void run(const Chunk & ch, int s, int lane, const ActView * av, float * acc) {
        const int j = lane & 7, b = 8 * s + (lane >> 3);
        const uint32_t qw[4] = { ch.q.x, ch.q.y, ch.q.z, ch.q.w };
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { lo[i] = iq4_lut4(qw[i] & 0x0F0F0F0Fu); hi[i] = iq4_lut4((qw[i] >> 4) & 0x0F0F0F0Fu); }
        const float d = h2f(ch.a.x & 0xFFFFu);
        const uint32_t sh = ch.a.x >> 16;                      // scales_h
        const int ls = (int) ((ch.a.y >> (4 * j)) & 0xFu) | (int) (((sh >> (2 * j)) & 3u) << 4);     // scales_l nibble j | 2 high bits
        const int e = 2048 * s + 256 * (lane >> 3) + 32 * j;
#pragma unroll
        for (int n = 0; n < NCOLS; ++n) {
            const int t = dot16(lo, av[n].q16(e)) + dot16(hi, av[n].q16(e + 16));
            acc[n] += ((d * av[n].dK(b)) * (float) (ls - 32)) * (float) t;     // CPU: d4d8 = d_x * d_y; d1 = d4d8 * (ls - 32); sumf += d1 * (sumi1 + sumi2)
        }
    }
};

// chunks per row for a type
__host__ __device__ __forceinline__ int row_chunks(int type, int k) {
    return type == MI355Q_TYPE_Q8_0 ? k / 16 : k / 32;       // Q4_0 / IQ4_NL: one per 32-block; K-quants / IQ4_XS: 8 per 256-block
}

// ------------------------------------------------------------------------------------------------
// activation quantization into LDS (once per workgroup).  A wave issues the activation loads of a
// pass (PRO spans of 256 floats) before it waits for any of them.  Kept deliberately compact (one
// copy per kernel, short loops): these kernels run for a few microseconds, instruction fetch counts.
// ------------------------------------------------------------------------------------------------
template <int FAM, bool ROUND_EVEN>
__device__ __forceinline__ void quantize_span_to_lds(const float4 v, int span, uint8_t * col, int k, int lane) {
    const int e0 = span * 256 + 4 * lane;
    if constexpr (FAM == FAM_Q8K) {
        uint32_t q; float d; int bsum;
        q8k_wave(v, q, d, bsum);
        *(uint32_t *) (col + e0) = q;
        if (lane == 0) *(float *) (col + k + 4 * span) = d;
        if ((lane & 3) == 0) *(int *) (col + k + (k >> 6) + 4 * (16 * span + (lane >> 2))) = bsum;
    } else {
        uint32_t q; float d; int sum;
        q80_group8<ROUND_EVEN>(v, q, d, sum);
        if (e0 < k) {
            *(uint32_t *) (col + e0) = q;
            if ((lane & 7) == 0) {
                const int b = e0 >> 5;
                *(float *) (col + k + 4 * b) = __half2float(__float2half_rn(d));
                *(int *) (col + k + (k >> 3) + 4 * b) = sum;
            }
        }
    }
}

// Spans of 256 activations are dealt round-robin to the waves, two per pass; the loads of pass i+1 are
// issued before pass i is processed, so only the first memory round trip is exposed.  No integer
// divisions, 32-bit indices only: this code runs once per workgroup in kernels that last microseconds.
//
// The first pass of column 0 is fetched by the CALLER (act_fetch) BEFORE it primes the weight ring:
// vmcnt retires in issue order, so activation loads issued behind 16 weight loads could only be consumed
// after those had landed; issued first, they are waited for with the weight loads still in flight.
__device__ __forceinline__ float4 act_fetch(const float * xr, int span, int k, int x_vec, int lane) {
    const int e0 = span * 256 + 4 * lane;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e0 < k) {                                                 // k is a multiple of 32: whole float4 in range
        if (x_vec) v = *(const float4 *) (xr + e0);
        else { v.x = xr[e0]; v.y = xr[e0 + 1]; v.z = xr[e0 + 2]; v.w = xr[e0 + 3]; }
    }
    return v;
}

template <int FAM, int NCOLS, bool ROUND_EVEN>
__device__ __forceinline__ void quantize_columns_to_lds(uint8_t * lds, int colb, const char * xbase, int64_t x_stride,
                                                         int k, int x_vec, int wave, int lane, float4 c0, float4 c1) {
    const int spans = (k + 255) >> 8;
#pragma unroll 1
    for (int n = 0; n < NCOLS; ++n) {
        const float * xr = (const float *) (xbase + (int64_t) n * x_stride);
        uint8_t * col = lds + n * colb;
        int span = wave;
        if (n > 0) { c0 = act_fetch(xr, span, k, x_vec, lane); c1 = act_fetch(xr, span + GEMV_WAVES, k, x_vec, lane); }
#pragma unroll 1
        while (span < spans) {
            const float4 n0 = act_fetch(xr, span + 2 * GEMV_WAVES, k, x_vec, lane), n1 = act_fetch(xr, span + 3 * GEMV_WAVES, k, x_vec, lane);
            quantize_span_to_lds<FAM, ROUND_EVEN>(c0, span, col, k, lane);
            if (span + GEMV_WAVES < spans) quantize_span_to_lds<FAM, ROUND_EVEN>(c1, span + GEMV_WAVES, col, k, lane);
            c0 = n0; c1 = n1; span += 2 * GEMV_WAVES;
        }
    }
    __syncthreads();
}

} // namespace mi355q
