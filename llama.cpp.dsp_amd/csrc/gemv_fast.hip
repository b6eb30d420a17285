// gemv_fast.hip -- the token-generation hot loop: y[n][m] = W_q[m][k] . x[n][k], n <= 8, for
// PLANAR device rows (see mi355q_common.h).  Replaces the reference's quantize_q8_1 + mul_mat_vec_q
// pair (ggml-cuda/quantize.cu:4-38, mmvq.cu:130-288) with ONE launch, designed for gfx950:
//
//  * fused prologue: every workgroup quantizes the activation columns itself (Q8_K / Q8_0, exactly
//    the CPU's arithmetic, act_quant.cuh) into LDS -- no separate quantize launch, no HBM round trip;
//    the f32 activations (<= 8 x 56 KiB) come out of L2.  The first weight loads are already in
//    flight while this runs.
//  * one wave = one weight row at a time; a row is cut into STEPS of 64 payload chunks of 16 bytes:
//    in a step lane l owns chunk 64*s + l, so a wave-wide global_load_dwordx4 reads 1 KiB of contiguous
//    packed quants (non-temporal: weights are streamed exactly once); the small per-block fields come
//    from their own dense planes.  Lane roles inside a block (which sub-block, which nibble half) are
//    loop-invariant because 64 is a multiple of the 8 chunks of a block.
//  * a ring of D steps per wave is kept in flight across row boundaries (the (row, step) items of a
//    wave form one flat stream): consume item i, then immediately re-issue the load of item i+D.
//    With <=128 VGPRs there are 16 waves per CU, i.e. ~16*D KiB of weight reads in flight per CU.
//  * nibble / 6-bit unpack with full-dword bit ops, v_dot4_i32_i8 against int8 activations from LDS
//    (ds_read_b128), exact int32 block sums, f32 scale, 64-lane shuffle reduction per row.
//  * several matrices that share the activations (wq/wk/wv, gate/up) run in ONE launch (row ranges
//    are concatenated; the type switch is wave-uniform), removing launch gaps from the token loop.
//
// Bound: HBM read of W.  Algorithmic bytes per launch = sum_i m_i * row_size(type_i, k).
#include "act_quant.cuh"

namespace mi355q {

constexpr int GEMV_THREADS  = 1024;           // 16 waves = ONE workgroup per CU: the activation quantization (done by every
                                              // workgroup for itself) then runs once per CU instead of twice
constexpr int GEMV_WAVES    = GEMV_THREADS / WAVE;
constexpr int GEMV_MAX_MATS = 4;

struct GemvMat {
    const uint8_t * w;
    float *         y;
    int64_t         w_stride;
    int64_t         y_stride;   // bytes between activation columns in y
    int64_t         row_begin;  // first concatenated row index of this matrix
    int             type;
    int             pad;
};

#ifdef MI355Q_STAMPS
// Diagnostic build only (libmi355q_dbg.so, -DMI355Q_STAMPS): lane 0 of every wave records 100 MHz wall-clock
// stamps at fixed points into g_stamps[(block*WAVES + wave)*8 + i].  The product library contains none of this.
__device__ unsigned long long * g_stamps = nullptr;
#define MI355Q_STAMP(i) do { if (g_stamps && lane == 0) g_stamps[((size_t) blockIdx.x * GEMV_WAVES + wave) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MI355Q_STAMP(i) do { } while (0)
#endif

struct GemvArgs {
    GemvMat       mats[GEMV_MAX_MATS];
    const float * x;
    int64_t       x_stride;     // bytes
    int64_t       total_rows;
    int64_t       rows_per_wg;
    int           n_mats;
    int           k;
    int           x_vec;        // x rows 16-byte aligned
    int           pad;
    // MUL_MAT_ID mode (ids != nullptr): blockIdx.y = (token t, slot u) pair; one matrix, one column.
    //   W = mats[0].w + ids[t][u]*expert_stride ; x = x + t*x_stride2 + (u % x_ne1)*x_stride ; y = mats[0].y + pair*m*4
    const int32_t * ids;
    int64_t       ids_stride;   // bytes between token rows of ids
    int64_t       expert_stride;
    int64_t       x_stride2;
    int           n_used;
    int           x_ne1;
    int           n_expert;
    int           pad2;
};

// ------------------------------------------------------------------------------------------------
// LDS image of the quantized activations, per column n (all offsets in bytes from the column base)
//   Q8_K family: q8[k] | d f32 [k/256] | bsums i32 [k/16]
//   Q8_0 family: q8[k] | d f32 [k/32] (already f16-rounded) | sums i32 [k/32]
// ------------------------------------------------------------------------------------------------
// kernel families: which weight types one kernel instantiation can stream (they share the activation image)
//   FAM_Q8K : Q4_K, Q6_K (Q8_K activations)   FAM_Q80 : Q8_0, Q4_0 (Q8_0 activations)   FAM_Q5K : Q5_K (Q8_K activations;
//   kept apart so that the in-flight ring of the hot Q4_K/Q6_K kernel needs 10 instead of 14 VGPRs per slot)
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
    ch.sc = *(const uint16_t *) (row + 192 * nb + 128 * s + 2 * lane);
    ch.dh = *(const uint16_t *) (row + 208 * nb + 16 * s + 2 * (lane >> 3));
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
    ch.dh = *(const uint16_t *) (row + 32 * nb + 64 * s + 2 * (lane >> 1));
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
    ch.dh = *(const uint16_t *) (row + 16 * nb + 128 * s + 2 * lane);
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

// chunks per row for a type
__host__ __device__ __forceinline__ int row_chunks(int type, int k) {
    return type == MI355Q_TYPE_Q8_0 ? k / 16 : k / 32;       // Q4_0: one per 32-block; K-quants: 8 per 256-block
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

// ------------------------------------------------------------------------------------------------
// Kernel.  WT = the single weight type of the launch (compile time), or WT_MIXED.
//
//  The rows of all matrices of the launch (all of type WT) form one concatenated range; each wave walks its
//  rows (r_begin+wave, +WAVES, ...) as one flat stream of (row, step) items with a ring of D items in
//  flight.  The first D items are requested BEFORE the activation quantization, so they fly during it.
//  (A launch of mixed types, e.g. Q4_K wq/wk + Q6_K wv of a Q4_K_M layer, is split per type by the host.)
// All cursor state is wave-uniform (SGPRs); descriptor fields are read with constant kernarg indices only,
// so they arrive in one batch of scalar loads at kernel entry.
// ------------------------------------------------------------------------------------------------

struct MatSel { const uint8_t * w; float * y; int64_t w_stride, y_stride; int rb, re, type; };

// descriptor of matrix i (i wave-uniform) through constant-index selects
template <bool MULTI>
__device__ __forceinline__ MatSel select_mat(const GemvArgs & a, int i) {
    MatSel m = { a.mats[0].w, a.mats[0].y, a.mats[0].w_stride, a.mats[0].y_stride, 0, (int) a.total_rows, a.mats[0].type };
    if constexpr (MULTI) {
        if (a.n_mats > 1) m.re = (int) a.mats[1].row_begin;
#pragma unroll
        for (int j = 1; j < GEMV_MAX_MATS; ++j) {
            const bool is = i == j;
            m.w = is ? a.mats[j].w : m.w; m.y = is ? a.mats[j].y : m.y;
            m.w_stride = is ? a.mats[j].w_stride : m.w_stride; m.y_stride = is ? a.mats[j].y_stride : m.y_stride;
            m.rb = is ? (int) a.mats[j].row_begin : m.rb; m.type = is ? a.mats[j].type : m.type;
            const int nxt = (j + 1 < GEMV_MAX_MATS && j + 1 < a.n_mats) ? (int) a.mats[j + 1 < GEMV_MAX_MATS ? j + 1 : j].row_begin : (int) a.total_rows;
            m.re = is ? nxt : m.re;
        }
    }
    return m;
}
// index of the matrix that owns concatenated row gr
template <bool MULTI>
__device__ __forceinline__ int mat_of_row(const GemvArgs & a, int gr) {
    int mi = 0;
    if constexpr (MULTI) {
#pragma unroll
        for (int j = 1; j < GEMV_MAX_MATS; ++j) if (j < a.n_mats && gr >= (int) a.mats[j].row_begin) mi = j;
    }
    return mi;
}

// Stream the concatenated rows [r_lo, r_hi) restricted to this wave (r_lo + wave, + WAVES, ...), all of type T.
// PRIME_ONLY / RUN_ONLY let the caller put the activation prologue between the two halves.
template <int T, int NCOLS, int D, bool MULTI>
struct Streamer {
    int ld_gr, ld_s, cs_gr, cs_s, r_hi, nb, nchunks, steps;
    const uint8_t * ld_row;
    float * cs_y; int64_t cs_ys;
    Chunk ring[D];

    __device__ __forceinline__ void resolve_ld(const GemvArgs & a, int64_t w_off) {
        const MatSel m = select_mat<MULTI>(a, mat_of_row<MULTI>(a, ld_gr));
        ld_row = m.w + w_off + (int64_t) (ld_gr - m.rb) * m.w_stride;
    }
    __device__ __forceinline__ void resolve_cs(const GemvArgs & a, int64_t y_off) {
        const MatSel m = select_mat<MULTI>(a, mat_of_row<MULTI>(a, cs_gr));
        cs_y = (float *) ((char *) m.y + y_off) + (cs_gr - m.rb); cs_ys = m.y_stride;
    }
    __device__ __forceinline__ void issue(Chunk & slot, const GemvArgs & a, int64_t w_off, int lane) {
        if (ld_gr < r_hi) {                                   // wave-uniform
            if (64 * ld_s + lane < nchunks) chunk_load<T>(slot, ld_row, nb, ld_s, lane);
            if (++ld_s == steps) { ld_s = 0; ld_gr += GEMV_WAVES; if (ld_gr < r_hi) resolve_ld(a, w_off); }
        }
    }
    __device__ __forceinline__ void prime(const GemvArgs & a, int r_lo, int r_hi_, int k, int wave, int lane, int64_t w_off, int64_t y_off) {
        r_hi = r_hi_;
        nb = (T == MI355Q_TYPE_Q8_0 || T == MI355Q_TYPE_Q4_0) ? k >> 5 : k >> 8;
        nchunks = row_chunks(T, k); steps = (nchunks + 63) >> 6;
        ld_gr = cs_gr = r_lo + wave; ld_s = cs_s = 0;
        if (ld_gr < r_hi) { resolve_ld(a, w_off); resolve_cs(a, y_off); }
#pragma unroll
        for (int d = 0; d < D; ++d) issue(ring[d], a, w_off, lane);
    }
    __device__ __forceinline__ void run(const GemvArgs & a, const ActView * av, int lane, int64_t w_off, int64_t y_off) {
        float acc[NCOLS];
#pragma unroll
        for (int n = 0; n < NCOLS; ++n) acc[n] = 0.0f;
        while (cs_gr < r_hi) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (cs_gr < r_hi) {                           // wave-uniform
                    if (64 * cs_s + lane < nchunks) Consume<T, NCOLS>::run(ring[d], cs_s, lane, av, acc);
                    if (++cs_s == steps) {                    // row finished: reduce, store, next row
#pragma unroll
                        for (int n = 0; n < NCOLS; ++n) {
                            const float t = wave_sum(acc[n]);
                            if (lane == 0) *(float *) ((char *) cs_y + (int64_t) n * cs_ys) = t;
                            acc[n] = 0.0f;
                        }
                        cs_s = 0; cs_gr += GEMV_WAVES;
                        if (cs_gr < r_hi) resolve_cs(a, y_off);
                    }
                    issue(ring[d], a, w_off, lane);           // refill the slot just consumed
                }
            }
        }
    }
};

template <int FAM, int WT, int NCOLS, int D, bool ROUND_EVEN, bool MULTI>
__global__ void __launch_bounds__(GEMV_THREADS)
k_gemv_fast(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));   // wave-uniform -> SGPR
    const int k    = a.k;
    const int colb = (lds_col_bytes(FAM, k) + 15) & ~15;
    MI355Q_STAMP(0);

    const char * xbase = (const char *) a.x;
    int64_t w_off = 0, y_off = 0;
    if (a.ids) {                                              // MoE: resolve this pair's expert on the device
        const int pair = (int) blockIdx.y, t = pair / a.n_used, u = pair - t * a.n_used;
        const int e = *(const int32_t *) ((const char *) a.ids + (int64_t) t * a.ids_stride + 4 * u);
        if (e < 0 || e >= a.n_expert) return;                 // (the reference asserts; we leave the row untouched)
        xbase += (int64_t) t * a.x_stride2 + (int64_t) (u % a.x_ne1) * a.x_stride;
        w_off = (int64_t) e * a.expert_stride;
        y_off = (int64_t) pair * a.total_rows * 4;
    }

    // this workgroup's contiguous range of the concatenated rows
    const int r_begin = (int) blockIdx.x * (int) a.rows_per_wg;
    int       r_end   = r_begin + (int) a.rows_per_wg;
    if (r_end > (int) a.total_rows) r_end = (int) a.total_rows;

    ActView av[NCOLS];
#pragma unroll
    for (int n = 0; n < NCOLS; ++n) { av[n].base = lds + n * colb; av[n].k = k; }

    Streamer<WT, NCOLS, D, MULTI> st;
    MI355Q_STAMP(1);
    // order matters (vmcnt retires in issue order): activation loads first, then the weight ring
    const float4 c0 = act_fetch((const float *) xbase, wave, k, a.x_vec, lane);
    const float4 c1 = act_fetch((const float *) xbase, wave + GEMV_WAVES, k, a.x_vec, lane);
    st.prime(a, r_begin, r_end, k, wave, lane, w_off, y_off);              // first D items in flight ...
    MI355Q_STAMP(2);
    quantize_columns_to_lds<FAM, NCOLS, ROUND_EVEN>(lds, colb, xbase, a.x_stride, k, a.x_vec, wave, lane, c0, c1);   // ... while the activations are quantized
    MI355Q_STAMP(3);
    st.run(a, av, lane, w_off, y_off);
    MI355Q_STAMP(5);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int family_of(int type) {
    switch (type) {
    case MI355Q_TYPE_Q4_K: case MI355Q_TYPE_Q5_K: case MI355Q_TYPE_Q6_K: return FAM_Q8K;
    case MI355Q_TYPE_Q8_0: case MI355Q_TYPE_Q4_0: return FAM_Q80;
    default: return -1;
    }
}

int gemv_fast_family(int type) { return family_of(type); }

// largest number of activation columns whose LDS image fits in the CU's 160 KiB (one workgroup per CU), <= 8
int gemv_fast_max_cols(int type, int64_t k) {
    const int fam = family_of(type);
    if (fam < 0) return 0;
    const int colb = (lds_col_bytes(fam, (int) k) + 15) & ~15;
    const int n = (160 * 1024 - 1024) / colb;
    return n > 8 ? 8 : n;
}

template <int FAM, int WT, int NCOLS, int D, bool EVEN, bool MULTI>
static int launch_one_m(const GemvArgs & a, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    auto kern = k_gemv_fast<FAM, WT, NCOLS, D, EVEN, MULTI>;
    static size_t lds_enabled = 48 * 1024;                 // per kernel instantiation
    if (lds_bytes > lds_enabled) {
        if (hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return MI355Q_ERR_HIP;
        lds_enabled = 160 * 1024;
    }
    hipLaunchKernelGGL(kern, grid, dim3(GEMV_THREADS), lds_bytes, stream, a);
    return hipGetLastError() == hipSuccess ? MI355Q_OK : MI355Q_ERR_HIP;
}

template <int FAM, int WT, int NCOLS, int D, bool EVEN>
static int launch_one(const GemvArgs & a, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    if (a.n_mats > 1) return launch_one_m<FAM, WT, NCOLS, D, EVEN, true>(a, grid, lds_bytes, stream);
    return launch_one_m<FAM, WT, NCOLS, D, EVEN, false>(a, grid, lds_bytes, stream);
}

// ring depth D (1-KiB steps in flight per wave): 8 for one column (a whole K=14336 row), shallower as columns eat VGPRs.
// Every N=1 launch of a token of one weight type runs the SAME kernel, so its code stays hot in the
// instruction cache across the ~130 launches of a token.
template <int FAM, int WT, bool EVEN>
static int launch_cols(const GemvArgs & a, int ncols, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    switch (ncols) {
    case 1: return launch_one<FAM, WT, 1, 8, EVEN>(a, grid, lds_bytes, stream);
    case 2: return launch_one<FAM, WT, 2, 6, EVEN>(a, grid, lds_bytes, stream);
    case 3: return launch_one<FAM, WT, 3, 3, EVEN>(a, grid, lds_bytes, stream);
    case 4: return launch_one<FAM, WT, 4, 3, EVEN>(a, grid, lds_bytes, stream);
    case 5: return launch_one<FAM, WT, 5, 2, EVEN>(a, grid, lds_bytes, stream);
    case 6: return launch_one<FAM, WT, 6, 2, EVEN>(a, grid, lds_bytes, stream);
    case 7: return launch_one<FAM, WT, 7, 2, EVEN>(a, grid, lds_bytes, stream);
    case 8: return launch_one<FAM, WT, 8, 2, EVEN>(a, grid, lds_bytes, stream);
    default: return MI355Q_ERR_UNSUPPORTED;
    }
}

struct MoeArgs {
    const int32_t * ids; int64_t ids_stride; int64_t expert_stride; int64_t x_stride2;
    int n_used; int x_ne1; int n_expert; int n_pairs;
};

static int launch_gemv_fast_typed(const mi355q_mat * mats, int n_mats, const float * x, int64_t x_stride,
                                  int ncols, int64_t k, int flags, int n_cu, hipStream_t stream, const MoeArgs * moe);

// mats: planar device rows of one activation family, same k.  ncols <= gemv_fast_max_cols().
// Matrices of different weight types are grouped per type (one launch per type).
int launch_gemv_fast(const mi355q_mat * mats, int n_mats, const float * x, int64_t x_stride,
                     int ncols, int64_t k, int flags, int n_cu, hipStream_t stream, const MoeArgs * moe = nullptr) {
    if (n_mats < 1 || n_mats > GEMV_MAX_MATS) return MI355Q_ERR_UNSUPPORTED;
    bool done[GEMV_MAX_MATS] = { false, false, false, false };
    for (int i = 0; i < n_mats; ++i) {
        if (done[i]) continue;
        mi355q_mat grp[GEMV_MAX_MATS]; int ng = 0;
        for (int j = i; j < n_mats; ++j) if (!done[j] && mats[j].type == mats[i].type) { grp[ng++] = mats[j]; done[j] = true; }
        const int rc = launch_gemv_fast_typed(grp, ng, x, x_stride, ncols, k, flags, n_cu, stream, moe);
        if (rc != MI355Q_OK) return rc;
    }
    return MI355Q_OK;
}

static int launch_gemv_fast_typed(const mi355q_mat * mats, int n_mats, const float * x, int64_t x_stride,
                                  int ncols, int64_t k, int flags, int n_cu, hipStream_t stream, const MoeArgs * moe) {
    if (n_mats < 1 || n_mats > GEMV_MAX_MATS) return MI355Q_ERR_UNSUPPORTED;
    const int fam = family_of(mats[0].type);
    if (fam < 0) return MI355Q_ERR_UNSUPPORTED;
    GemvArgs a = {};
    int64_t rows = 0;
    int max_steps = 0;
    for (int i = 0; i < n_mats; ++i) {
        if (family_of(mats[i].type) != fam) return MI355Q_ERR_UNSUPPORTED;
        if (((uintptr_t) mats[i].w | (uintptr_t) mats[i].w_stride) & 15) return MI355Q_ERR_ALIGN;
        a.mats[i].w = (const uint8_t *) mats[i].w; a.mats[i].y = mats[i].y;
        a.mats[i].w_stride = mats[i].w_stride; a.mats[i].y_stride = mats[i].y_stride;
        a.mats[i].row_begin = rows; a.mats[i].type = mats[i].type;
        rows += mats[i].m;
        const int st = (row_chunks(mats[i].type, (int) k) + 63) / 64;
        if (st > max_steps) max_steps = st;
    }
    if (rows == 0) return MI355Q_OK;
    if (rows > 0x7FFFFFF0) return MI355Q_ERR_UNSUPPORTED;
    a.x = x; a.x_stride = x_stride; a.total_rows = rows; a.n_mats = n_mats; a.k = (int) k;
    a.x_vec = (((uintptr_t) x | (uintptr_t) x_stride) & 15) == 0 ? 1 : 0;
    int pairs = 1;
    if (moe) {
        if (n_mats != 1 || ncols != 1) return MI355Q_ERR_UNSUPPORTED;
        a.ids = moe->ids; a.ids_stride = moe->ids_stride; a.expert_stride = moe->expert_stride; a.x_stride2 = moe->x_stride2;
        a.n_used = moe->n_used; a.x_ne1 = moe->x_ne1; a.n_expert = moe->n_expert;
        pairs = moe->n_pairs;
        if (pairs <= 0) return MI355Q_OK;
        if (pairs > 65535) return MI355Q_ERR_UNSUPPORTED;
    }
    // grid: one 16-wave workgroup per CU, each a contiguous row range
    int64_t grid = (int64_t) n_cu / pairs;
    if (grid < 1) grid = 1;
    int64_t rpw = (rows + grid - 1) / grid;
    if (rpw < 1) rpw = 1;
    grid = (rows + rpw - 1) / rpw;
    a.rows_per_wg = rpw;
    const int colb = (lds_col_bytes(fam, (int) k) + 15) & ~15;
    const size_t lds_bytes = (size_t) colb * ncols;
    const bool even = (flags & MI355Q_FLAG_ROUND_EVEN) != 0;
    const dim3 g((unsigned) grid, (unsigned) pairs);
    const int wt = mats[0].type;                               // the caller groups by type
    if (fam == FAM_Q8K) {                                      // Q8_K activations have one rounding rule
        switch (wt) {
        case MI355Q_TYPE_Q4_K: return launch_cols<FAM_Q8K, MI355Q_TYPE_Q4_K, false>(a, ncols, g, lds_bytes, stream);
        case MI355Q_TYPE_Q6_K: return launch_cols<FAM_Q8K, MI355Q_TYPE_Q6_K, false>(a, ncols, g, lds_bytes, stream);
        default:               return launch_cols<FAM_Q8K, MI355Q_TYPE_Q5_K, false>(a, ncols, g, lds_bytes, stream);
        }
    }
    if (even) return wt == MI355Q_TYPE_Q8_0 ? launch_cols<FAM_Q80, MI355Q_TYPE_Q8_0, true>(a, ncols, g, lds_bytes, stream)
                                            : launch_cols<FAM_Q80, MI355Q_TYPE_Q4_0, true>(a, ncols, g, lds_bytes, stream);
    return wt == MI355Q_TYPE_Q8_0 ? launch_cols<FAM_Q80, MI355Q_TYPE_Q8_0, false>(a, ncols, g, lds_bytes, stream)
                                  : launch_cols<FAM_Q80, MI355Q_TYPE_Q4_0, false>(a, ncols, g, lds_bytes, stream);
}

#ifdef MI355Q_STAMPS
extern "C" int mi355q_debug_set_stamps(void * dev_buf) {
    unsigned long long * p = (unsigned long long *) dev_buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -4;
}
#endif

} // namespace mi355q
