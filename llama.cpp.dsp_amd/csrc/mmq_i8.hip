// mmq_i8.hip -- the batched (prefill) tier for Q4_K on the INTEGER matrix cores: y[n][m] = W_q4k[m][k] . x[n][k], n > 8.
//
// The CPU backend multiplies a Q4_K row with Q8_K-quantized activations super-block by super-block (ggml-cpu-quants.c
// ggml_vec_dot_q4_K_q8_K; block formats ggml-common.h:289-303, 329-334):
//      y += d*dx * sum_g sc_g * (q_g . x_g)  -  dmin*dx * sum_g m_g * bsum_g          (g = the 8 groups of 32 weights)
// with exact integer sums.  This kernel evaluates exactly that expression -- same activation quantizer (act_quant.cuh, bit-equal to
// quantize_row_q8_K), same integers -- with the two integer sums on the matrix cores:
//   * sum_g sc_g (q_g . x_g): the 6-bit group scale is folded into the weights BEFORE the multiply, split in two int8 planes,
//     sc = 8*sh + sl:  q*sl <= 105 and q*sh <= 105 both fit a signed byte, so one v_mfma_i32_16x16x64_i8 per plane covers two
//     groups with different scales (64 contraction slots) and the whole super-block accumulates in two i32 tiles:
//     isum = 8*acc_h + acc_l.  No per-group rescaling of outputs; the nibble -> plane work is one v_pk_mul_lo_u16 per 4 weights.
//   * sum_g m_g * bsum_g: one v_mfma_f32_16x16x32_bf16 with small exact integers: bsum = 64*hi + lo (hi, lo, m_g, 64*m_g are all
//     exactly representable in bf16; the f32 accumulation of 16 products < 2^24 is exact).
// Per super-block and 16x16 output tile: 8 integer MFMAs (K = 64) + 1 bf16 MFMA, then ONE f32 update d*dx*isum - dmin*dx*msum.
// Replaces the reference's mul_mat_q tiles (ggml-cuda/mmq.cuh:2595-2674, dp4a on CDNA) and its dequantize + hipBLAS route.
//
// Data flow: each wave owns RT*16 weight rows and dequantizes them from global memory straight into MFMA B-operand registers (no LDS
// round trip, no duplicate dequantization inside a workgroup); the quantized activation tile of the super-block (BN tokens x 256 int8,
// + scales and group sums) is DMA-copied global -> LDS (global_load_lds_dwordx4, double buffered) and read as A operands by all waves.
// Workgroup = 4 waves = (64*RT) rows x (16*TT) tokens; accumulators RT*TT*4 VGPRs.
// Bound: MFMA (9 x 16 cycles per 16x16x256 tile-block = 0.89 of the dense bf16 rate at best); HBM: packed W once per token tile.
#include "mi355q_common.h"
#include "act_quant.cuh"

namespace mi355q {

typedef __attribute__((ext_vector_type(8))) __bf16 i8q_bf16x8;
typedef __attribute__((ext_vector_type(4))) float  i8q_f32x4;
typedef __attribute__((ext_vector_type(4))) int    i8q_i32x4;
typedef __attribute__((ext_vector_type(2))) unsigned short i8q_u16x2;
typedef __attribute__((address_space(3))) void * i8q_lds_ptr;

// activation image, per super-block b and token n (tokens padded to a multiple of 128 with zero records):
//   XQ [nb][n_pad][272 B] : 256 int8 in MFMA slot order (16-byte piece P = 8a+4b+2c+d of the block  ->  step s = 2a+c, lane quarter kq = 2b+d,
//                           at 64*s + 16*kq) + 16 B of padding (LDS bank spread); the record IS the LDS row.
//   XS [nb][n_pad][ 32 B] : 16 bf16: [lo0 lo1 lo4 lo5 hi0 hi1 hi4 hi5 | lo2 lo3 lo6 lo7 hi2 hi3 hi6 hi7], bsum32_g = 64*hi_g + lo_g, 0 <= lo < 64
//   XD [nb][n_pad] f32    : the Q8_K block scale
constexpr int I8Q_REC = 272, I8Q_XS = 32, I8Q_TOK_ALIGN = 128;
__host__ __device__ inline int64_t i8q_npad(int64_t n) { return (n + I8Q_TOK_ALIGN - 1) / I8Q_TOK_ALIGN * I8Q_TOK_ALIGN; }

__global__ void __launch_bounds__(256)
k_mmq_i8_prep(const float * __restrict__ x, int64_t x_stride, uint8_t * __restrict__ xq, uint8_t * __restrict__ xs, float * __restrict__ xd,
              int n, int n_pad, int nb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x, tok = 4 * blockIdx.y + wave;
    if (tok >= n_pad) return;
    uint8_t * rec = xq + ((int64_t) b * n_pad + tok) * I8Q_REC;
    uint8_t * srec = xs + ((int64_t) b * n_pad + tok) * I8Q_XS;
    uint32_t q = 0; float d = 0.0f; int bsum = 0;
    if (tok < n) {
        const float4 v = *(const float4 *) ((const char *) x + (int64_t) tok * x_stride + 4 * (256 * (int64_t) b + 4 * lane));
        q8k_wave(v, q, d, bsum);
    }
    // Record byte 4*L comes from source dword D(L): the record is [step s][lane quarter kq][16 B] and holds source piece P = 8a+4b+2c+d at
    // s = 2a+c, kq = 2b+d.  The quants are exchanged between lanes (one ds_bpermute) so that the wave writes its 256 bytes as ONE contiguous
    // store -- written from the natural order the record is 16 scattered 16-byte segments and the kernel runs at a third of the speed.
    const int s = lane >> 4, kq = (lane >> 2) & 3, P = 8 * (s >> 1) + 4 * (kq >> 1) + 2 * (s & 1) + (kq & 1);
    const uint32_t qp = (uint32_t) __shfl((int) q, 4 * P + (lane & 3), 64);
    *(uint32_t *) (rec + 4 * lane) = qp;
    if (lane < 4) *(uint32_t *) (rec + 256 + 4 * lane) = 0u;
    if (lane == 0) xd[(int64_t) b * n_pad + tok] = d;
    // group sums: the 32-group g lives in lanes 8g..8g+7; lanes 0..7 each assemble one dword (two bf16) of the 32-byte record
    const int bs32 = bsum + __shfl_xor(bsum, 4);
    const int k8 = lane & 7;
    const int g0 = (k8 & 4 ? 2 : 0) + (k8 & 1 ? 4 : 0), want_hi = (k8 >> 1) & 1;      // dwords: (lo0 lo1)(lo4 lo5)(hi0 hi1)(hi4 hi5) | (lo2 lo3)(lo6 lo7)(hi2 hi3)(hi6 hi7)
    const int sa = __shfl(bs32, 8 * g0, 64), sb = __shfl(bs32, 8 * (g0 + 1), 64);
    const int va = want_hi ? sa >> 6 : sa & 63, vb = want_hi ? sb >> 6 : sb & 63;       // bs32 = 64*hi + lo (arithmetic shift), 0 <= lo < 64, |hi| <= 64
    if (lane < 8) *(uint32_t *) (srec + 4 * lane) = (__float_as_uint((float) va) >> 16) | (__float_as_uint((float) vb) & 0xFFFF0000u);   // small integers: exact in bf16
}

__device__ __forceinline__ uint32_t i8q_mul_bytes(uint32_t nib, uint32_t mul16x2) {     // 4 bytes (each <= 15) times a multiplier <= 7: one v_pk_mul_lo_u16
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(i8q_u16x2, nib) * __builtin_bit_cast(i8q_u16x2, mul16x2));
}
__device__ __forceinline__ uint32_t i8q_bf16_pair(int a, int b) {                       // two small non-negative integers -> packed bf16 (exact)
    return (__float_as_uint((float) a) >> 16) | (__float_as_uint((float) b) & 0xFFFF0000u);
}

// Up to 4 matrices multiplied with the same activations in ONE grid (wq|wk|wv, ffn_gate|ffn_up): their row blocks are concatenated, so a
// 1024-row wk does not become a launch of its own that leaves most of the chip idle.  rb_begin[i] = first row block of matrix i.
struct I8Mats { const uint8_t * w[4]; int64_t w_stride[4]; float * y[4]; int64_t y_stride[4]; int m[4]; int rb_begin[4]; };

template <int RT, int TT, bool STAMPS>
__global__ void __launch_bounds__(256, 2)
k_mmq_i8_q4k(const I8Mats mats, const uint8_t * __restrict__ xq, const uint8_t * __restrict__ xs, const float * __restrict__ xd,
             int n, int n_pad, int nb, int n_tok_tiles, int total_tiles, int per_xcd, int n_split, int64_t split_stride, const MoeTiles moe) {
    constexpr int BN = 16 * TT, BM = 64 * RT;
    constexpr int XQ_BYTES = BN * I8Q_REC, XS_BYTES = BN * I8Q_XS, XD_BYTES = BN * 4, BUF = XQ_BYTES + XS_BYTES + XD_BYTES;
    constexpr int XQ_PIECES = XQ_BYTES / 1024, XS_PIECES = XS_BYTES / 1024;          // 1 KiB per wave-instruction (64 lanes x 16 B)
    static_assert(XQ_BYTES % 1024 == 0 && XS_BYTES % 1024 == 0 && BN * 4 <= XD_BYTES, "DMA pieces");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    // workgroups that share weight rows (the token tiles of one row block) are consecutive on ONE XCD: its L2 serves the re-reads
    const int v = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (v >= total_tiles) return;
    const int rb_all = v / n_tok_tiles, tt0 = v - rb_all * n_tok_tiles;
    // which matrix (scalar selects on kernel arguments; rb_begin of unused slots is INT_MAX)
    const int mi = (rb_all >= mats.rb_begin[1]) + (rb_all >= mats.rb_begin[2]) + (rb_all >= mats.rb_begin[3]);
    const uint8_t * __restrict__ w = mi == 0 ? mats.w[0] : mi == 1 ? mats.w[1] : mi == 2 ? mats.w[2] : mats.w[3];
    if (moe.tile_expert) {                                    // grouped MUL_MAT_ID: this token tile's expert (uniform per workgroup)
        const int e = moe.tile_expert[tt0 * (16 * TT) / moe.tile_tokens];
        if (e < 0) return;
        w += (int64_t) e * moe.expert_stride;
        n = moe.seg_end[e];
    }
    const int64_t w_stride = mi == 0 ? mats.w_stride[0] : mi == 1 ? mats.w_stride[1] : mi == 2 ? mats.w_stride[2] : mats.w_stride[3];
    float * __restrict__ y = mi == 0 ? mats.y[0] : mi == 1 ? mats.y[1] : mi == 2 ? mats.y[2] : mats.y[3];
    const int64_t y_stride = mi == 0 ? mats.y_stride[0] : mi == 1 ? mats.y_stride[1] : mi == 2 ? mats.y_stride[2] : mats.y_stride[3];
    const int m = mi == 0 ? mats.m[0] : mi == 1 ? mats.m[1] : mi == 2 ? mats.m[2] : mats.m[3];
    const int rb = rb_all - (mi == 0 ? 0 : mi == 1 ? mats.rb_begin[1] : mi == 2 ? mats.rb_begin[2] : mats.rb_begin[3]);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // scalar: the DMA piece loop branches on it
    const int l16 = lane & 15, kq = lane >> 4, h16 = 16 * (kq >> 1);
    const int m0 = rb * BM + wave * 16 * RT, n0 = tt0 * BN;
    const uint32_t live = (kq & 1) ? 0u : 0xFFFFFFFFu;       // lane quarters 0 and 2 carry the slots of the bf16 (group minima) MFMA

    const uint8_t * wrow[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { const int r = m0 + 16 * rt + l16; wrow[rt] = w + (int64_t) (r < m ? r : 0) * w_stride; }

    i8q_f32x4 facc[RT][TT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < TT; ++t) facc[rt][t] = (i8q_f32x4) { 0.f, 0.f, 0.f, 0.f };

    uint4 rq0[RT], rq1[RT], rh[RT];
    auto fetch_w = [&](int b) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const uint8_t * p = wrow[rt] + 128 * (int64_t) b + 16 * kq;
            rq0[rt] = ldg16(p); rq1[rt] = ldg16(p + 64);              // (kept in L2: the other token tiles of this row block read the same bytes)
            rh[rt]  = ldg16(wrow[rt] + 128 * (int64_t) nb + 16 * b);
        }
    };
    auto dma_x = [&](int b, int buf) {                        // the activation tile of super-block b -> LDS buffer buf (lane-linear 1 KiB pieces)
        uint8_t * dst = lds + buf * BUF;
        const uint8_t * gq = xq + ((int64_t) b * n_pad + n0) * I8Q_REC;
        const uint8_t * gs = xs + ((int64_t) b * n_pad + n0) * I8Q_XS;
        const uint8_t * gd = (const uint8_t *) (xd + (int64_t) b * n_pad + n0);
#pragma unroll
        for (int i = 0; i < (XQ_PIECES + 3) / 4; ++i) {
            const int p = wave + 4 * i;
            if (p < XQ_PIECES) __builtin_amdgcn_global_load_lds(gq + 1024 * p + 16 * lane, (i8q_lds_ptr) (dst + 1024 * p), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < (XS_PIECES + 3) / 4; ++i) {
            const int p = wave + 4 * i;
            if (p < XS_PIECES) __builtin_amdgcn_global_load_lds(gs + 1024 * p + 16 * lane, (i8q_lds_ptr) (dst + XQ_BYTES + 1024 * p), 16, 0, 0);
        }
        if (wave == 3 && 16 * lane < BN * 4) __builtin_amdgcn_global_load_lds(gd + 16 * lane, (i8q_lds_ptr) (dst + XQ_BYTES + XS_BYTES), 16, 0, 0);
    };

    // split K (blockIdx.y, single-matrix launches): piece z covers super-blocks [b_lo, b_hi) and writes partial sums to y + z * split_stride
    const int b_lo = (int) ((int64_t) nb * blockIdx.y / n_split), b_hi = (int) ((int64_t) nb * (blockIdx.y + 1) / n_split);
    fetch_w(b_lo);
    dma_x(b_lo, 0);

    [[maybe_unused]] uint64_t st_wait = 0, st_deq = 0, st_tiles = 0, st_t0 = 0, st_t1 = 0, st_t2 = 0;
    for (int b = b_lo; b < b_hi; ++b) {
        if constexpr (STAMPS) st_t0 = __builtin_amdgcn_s_memrealtime();
        const int buf = (b - b_lo) & 1;
        // everything this wave issued (its share of tile b, its packed weights of block b) has landed; after the barrier so has every
        // other wave's share, and nobody still reads the other buffer
        __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0); expcnt / lgkmcnt untouched
        __syncthreads();
        if constexpr (STAMPS) { st_t1 = __builtin_amdgcn_s_memrealtime(); st_wait += st_t1 - st_t0; }

        // ---- block b: packed nibbles -> two int8 planes with the group scale folded in; group minima -> a bf16 B operand ----
        i8q_i32x4 pl[RT][4], ph[RT][4];                       // [step] : B operand (16 bytes = this lane's 16 contraction slots) of plane l / h
        i8q_bf16x8 mfrag[RT];
        float dW[RT], dmW[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const uint4 hd = rh[rt];
            dW[rt] = h2f(hd.x & 0xFFFFu); dmW[rt] = h2f(hd.x >> 16);
            // this lane's groups: 2h, 2h+1 (scale bytes 2h, 2h+1 / min bytes 4+2h, 5+2h) and 4+2h, 5+2h (bytes 8+2h, 9+2h + the top bits)   [get_scale_min_k4]
            const uint32_t a = (hd.y >> h16) & 0xFFFFu, bb = (hd.z >> h16) & 0xFFFFu, c = (hd.w >> h16) & 0xFFFFu;
            const int sc0 = a & 63, sc1 = (a >> 8) & 63, mn0 = bb & 63, mn1 = (bb >> 8) & 63;
            const int sc2 = (c & 0xF) | (((a >> 6) & 3) << 4), sc3 = ((c >> 8) & 0xF) | (((a >> 14) & 3) << 4);
            const int mn2 = ((c >> 4) & 0xF) | (((bb >> 6) & 3) << 4), mn3 = ((c >> 12) & 0xF) | (((bb >> 14) & 3) << 4);
            const int scs[4] = { sc0, sc1, sc2, sc3 };
            const uint32_t raw[2][4] = { { rq0[rt].x, rq0[rt].y, rq0[rt].z, rq0[rt].w }, { rq1[rt].x, rq1[rt].y, rq1[rt].z, rq1[rt].w } };
#pragma unroll
            for (int s = 0; s < 4; ++s) {                     // step s: load s/2, low (s even) or high (s odd) nibbles
                const uint32_t sl = (uint32_t) (scs[s] & 7) * 0x00010001u, sh = (uint32_t) (scs[s] >> 3) * 0x00010001u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t nib = (s & 1) ? (raw[s >> 1][j] >> 4) & 0x0F0F0F0Fu : raw[s >> 1][j] & 0x0F0F0F0Fu;
                    pl[rt][s][j] = (int) i8q_mul_bytes(nib, sl);
                    ph[rt][s][j] = (int) i8q_mul_bytes(nib, sh);
                }
            }
            // slots of the bf16 MFMA (lane quarters 0 and 2 only): [m m m m 64m 64m 64m 64m] of this lane's 4 groups
            const uint32_t f0 = i8q_bf16_pair(mn0, mn1) & live, f1 = i8q_bf16_pair(mn2, mn3) & live;
            const uint32_t f2 = i8q_bf16_pair(mn0 << 6, mn1 << 6) & live, f3 = i8q_bf16_pair(mn2 << 6, mn3 << 6) & live;
            const i8q_i32x4 fm = { (int) f0, (int) f1, (int) f2, (int) f3 };
            mfrag[rt] = __builtin_bit_cast(i8q_bf16x8, fm);
        }
        // ---- prefetch block b+1 (weights into the registers just consumed, activations into the other LDS buffer) ----
        if (b + 1 < b_hi) { fetch_w(b + 1); dma_x(b + 1, buf ^ 1); }
        __builtin_amdgcn_sched_barrier(0);                      // keep the phases apart: the scheduler otherwise mixes them into a slower order

        if constexpr (STAMPS) { st_t2 = __builtin_amdgcn_s_memrealtime(); st_deq += st_t2 - st_t1; }
        const uint8_t * xq_l = lds + buf * BUF, * xs_l = xq_l + XQ_BYTES, * xd_l = xs_l + XS_BYTES;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const uint8_t * trow = xq_l + (16 * t + l16) * I8Q_REC + 16 * kq;
            i8q_i32x4 af[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) af[s] = *(const i8q_i32x4 *) (trow + 64 * s);
            const i8q_bf16x8 sf = *(const i8q_bf16x8 *) (xs_l + (16 * t + l16) * I8Q_XS + 16 * (kq >> 1));   // odd quarters: finite filler against zero weights
            const i8q_f32x4 dx = *(const i8q_f32x4 *) (xd_l + 4 * (16 * t + 4 * kq));                          // the 4 tokens of this lane's C rows
            // the 2*RT accumulator chains of this token tile advance together: a dependent v_mfma issues only every ~4th slot
            i8q_i32x4 il[RT], ih[RT];
            i8q_f32x4 ms[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) { il[rt] = (i8q_i32x4) { 0, 0, 0, 0 }; ih[rt] = (i8q_i32x4) { 0, 0, 0, 0 }; }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    il[rt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[s], pl[rt][s], il[rt], 0, 0, 0);
                    ih[rt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[s], ph[rt][s], ih[rt], 0, 0, 0);
                }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) ms[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf, mfrag[rt], (i8q_f32x4) { 0.f, 0.f, 0.f, 0.f }, 0, 0, 0);
            // C rows = tokens 4*kq + r, C column = weight row l16:  y += dx * (d * isum - dmin * msum)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float isum = (float) ((ih[rt][r] << 3) + il[rt][r]);
                    facc[rt][t][r] = __builtin_fmaf(dx[r], __builtin_fmaf(dW[rt], isum, -(dmW[rt] * ms[rt][r])), facc[rt][t][r]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (STAMPS) { st_tiles += __builtin_amdgcn_s_memrealtime() - st_t2; }
    }

    // ---- epilogue: lane holds 4 tokens x 1 weight row per tile ----
    [[maybe_unused]] uint64_t st_end = 0;
    if constexpr (STAMPS) st_end = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int row = m0 + 16 * rt + l16;
        if (row >= m) continue;
        char * ycol = (char *) (y + (int64_t) blockIdx.y * split_stride) + 4 * (int64_t) row;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tok = n0 + 16 * t + 4 * kq + r;
                if (tok < n) {
                    const int dr = moe.dst_row ? moe.dst_row[tok] : tok;      // grouped MUL_MAT_ID: straight to the pair's row of the result
                    if (dr >= 0) *(float *) (ycol + (int64_t) dr * y_stride) = facc[rt][t][r];
                }
            }
        }
    }
    if constexpr (STAMPS) {
        if (v == 0 && threadIdx.x == 0) { y[0] = (float) st_wait * 0.01f; y[1] = (float) st_deq * 0.01f; y[2] = (float) st_tiles * 0.01f; y[3] = (float) (__builtin_amdgcn_s_memrealtime() - st_end) * 0.01f; }
    }
}

// ---- host side ------------------------------------------------------------------------------------
bool mmq_i8_supported(int type, int64_t k) { return type == MI355Q_TYPE_Q4_K && k % 256 == 0; }

size_t mmq_i8_workspace(int64_t n, int64_t k) {
    return (size_t) (i8q_npad(n) * (k / 256) * (I8Q_REC + I8Q_XS + 4) + 1024 + 255) & ~(size_t) 255;      // + slack: the XD piece is copied as a full KiB
}

// w: planar device rows; x f32 [n][k] (row stride x_stride, 16-byte aligned rows); workspace >= mmq_i8_workspace(n,k); y f32 [n][m]
// One grid for up to 4 Q4_K matrices on the same activations.  w: planar device rows; x f32 [n][k] (16-byte aligned rows);
// workspace >= mmq_i8_workspace(n,k); y_i f32 [n][m_i].  prepare: write the activation image first.
void launch_mmq_reduce(const float * part, int n_split, int64_t split_stride, float * y, int64_t y_stride, int64_t m, int64_t n, hipStream_t stream);

// K pieces for a single matrix whose 128 x 128 tiles do not fill the chip (as the bf16 tier does): 0/1 = no split
static int mmq_i8_splits(int64_t m, int64_t n, int64_t k, int n_cu) {
    const int64_t t128 = ((m + 127) / 128) * (i8q_npad(n) / 128);
    if (2 * t128 >= 3 * (int64_t) n_cu || k < 8192 || m % 4) return 1;           // (K = 4096: measured no gain over 64 x 64 tiles)
    int s = (int) ((2 * (int64_t) n_cu + t128 - 1) / t128);
    const int max_s = (int) (k / 256 / 4);                                     // at least 4 super-blocks per piece
    if (s > max_s) s = max_s;
    if (s > 8) s = 8;
    return s < 2 ? 1 : s;
}
size_t mmq_i8_split_workspace(int64_t m, int64_t n, int64_t k, int n_cu) {
    const int s = mmq_i8_splits(m, n, k, n_cu);
    return s > 1 ? (size_t) s * (size_t) n * (size_t) m * 4 + 256 : 0;
}

int launch_mmq_i8_multi(const mi355q_mat * mt, int n_mats, const float * x, int64_t x_stride, int64_t n, int64_t k,
                        void * workspace, size_t workspace_bytes, int n_cu, hipStream_t stream, bool prepare, const MoeTiles * moe_p = nullptr) {
    MoeTiles moe = {}; if (moe_p) moe = *moe_p;
    if (n_mats < 1 || n_mats > 4) return MI355Q_ERR_SHAPE;
    for (int i = 0; i < n_mats; ++i) if (!mmq_i8_supported(mt[i].type, k)) return MI355Q_ERR_UNSUPPORTED;
    if (n <= 0) return MI355Q_OK;
    if ((x_stride & 15) || ((uintptr_t) x & 15)) return MI355Q_ERR_ALIGN;
    const int nb = (int) (k / 256);
    const int64_t n_pad = i8q_npad(n);
    uint8_t * xq = (uint8_t *) workspace;
    uint8_t * xs = xq + (size_t) nb * n_pad * I8Q_REC;
    float   * xd = (float *) (xs + (size_t) nb * n_pad * I8Q_XS);
    if (prepare) hipLaunchKernelGGL(k_mmq_i8_prep, dim3((unsigned) nb, (unsigned) (n_pad / 4)), dim3(256), 0, stream, x, x_stride, xq, xs, xd, (int) n, (int) n_pad, nb);
    // 128 rows x 128 tokens per workgroup (the dequantization is shared by most tokens) when that still gives a CU 1.5 workgroups;
    // otherwise 64 x 64 (four times the workgroups, 3-4 per CU): measured at N = 512, 4096 x 4096: 47 -> 42 us, 4096 x 14336: 147 -> 128 us
    int64_t rb128 = 0; for (int i = 0; i < n_mats; ++i) rb128 += (mt[i].m + 127) / 128;
    // ... or, for a single matrix, 128 x 128 tiles over K pieces (partial sums behind the activation image, added up by k_mmq_reduce)
    int splits = n_mats == 1 && !moe.tile_expert ? mmq_i8_splits(mt[0].m, n, k, n_cu) : 1;
    const size_t part_off = mmq_i8_workspace(n, k);
    if (splits > 1 && (workspace_bytes < part_off + (size_t) splits * (size_t) n * (size_t) mt[0].m * 4 || (mt[0].y_stride & 15) || ((uintptr_t) mt[0].y & 15))) splits = 1;
    mi355q_mat part_mat;
    if (splits > 1) { part_mat = mt[0]; part_mat.y = (float *) ((char *) workspace + part_off); part_mat.y_stride = 4 * mt[0].m; }
    const mi355q_mat * src = splits > 1 ? &part_mat : mt;
    const int64_t split_stride = splits > 1 ? n * mt[0].m : 0;
    const bool wide = splits > 1 || 2 * rb128 * (n_pad / 128) >= 3 * (int64_t) n_cu;
#define MI355Q_I8_LAUNCH(RT, TT, STAMPS) {                                                                                         \
        constexpr int bn = 16 * TT, bm = 64 * RT;                                                                                  \
        const size_t lds_bytes = 2 * (size_t) (bn * (I8Q_REC + I8Q_XS + 4));                                                       \
        static bool attr_set[64] = {};          /* the attribute is per device: the plugin drives every visible GPU from one process */ \
        int dev_ = 0; (void) hipGetDevice(&dev_); dev_ = dev_ >= 0 && dev_ < 64 ? dev_ : 0;                                          \
        if (!attr_set[dev_]) {                                                                                                     \
            if (hipFuncSetAttribute((const void *) k_mmq_i8_q4k<RT, TT, STAMPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes) != hipSuccess) \
                return MI355Q_ERR_HIP;                                                                                             \
            attr_set[dev_] = true;                                                                                                 \
        }                                                                                                                          \
        I8Mats im; int rbs = 0;                                                                                                    \
        for (int i = 0; i < 4; ++i) {                                                                                              \
            const int j = i < n_mats ? i : 0;                                                                                      \
            im.w[i] = (const uint8_t *) src[j].w; im.w_stride[i] = src[j].w_stride; im.y[i] = src[j].y; im.y_stride[i] = src[j].y_stride; im.m[i] = (int) src[j].m; \
            im.rb_begin[i] = i < n_mats ? rbs : 0x7FFFFFFF;                                                                        \
            if (i < n_mats) rbs += (int) ((mt[i].m + bm - 1) / bm);                                                                \
        }                                                                                                                          \
        const int n_tok_tiles = (int) (n_pad / bn), total = rbs * n_tok_tiles, per_xcd = (total + 7) / 8;                         \
        if (total > 0)                                                                                                             \
            hipLaunchKernelGGL((k_mmq_i8_q4k<RT, TT, STAMPS>), dim3((unsigned) (8 * per_xcd), (unsigned) splits), dim3(256), lds_bytes, stream, im, \
                               (const uint8_t *) xq, (const uint8_t *) xs, (const float *) xd, (int) n, (int) n_pad, nb, n_tok_tiles, total, per_xcd, splits, split_stride, moe); }
    static const bool stamps = getenv("MI355Q_I8_STAMPS") != nullptr;       // dev: phase times of workgroup 0 into y[0][0..3] (tools/pp_shape.py)
    static const int force = getenv("MI355Q_I8_CFG") ? atoi(getenv("MI355Q_I8_CFG")) : 0;      // dev: 28 / 24 / 18 / 14 = RT, TT
    const int cfg = force ? force : (moe.tile_expert ? (moe.tile_tokens == 128 ? 28 : 14) : (wide ? 28 : 14));   // (grouped: the tile the segments were aligned to)
    if (cfg == 28) { if (stamps) MI355Q_I8_LAUNCH(2, 8, true) else MI355Q_I8_LAUNCH(2, 8, false) }
    else if (cfg == 24) MI355Q_I8_LAUNCH(2, 4, false)
    else if (cfg == 18) MI355Q_I8_LAUNCH(1, 8, false)
    else MI355Q_I8_LAUNCH(1, 4, false)
#undef MI355Q_I8_LAUNCH
    if (splits > 1) launch_mmq_reduce(part_mat.y, splits, split_stride, mt[0].y, mt[0].y_stride, mt[0].m, n, stream);
    return hipGetLastError() == hipSuccess ? MI355Q_OK : MI355Q_ERR_HIP;
}

int launch_mmq_i8(int type, const void * w, int64_t w_stride, const float * x, int64_t x_stride,
                  float * y, int64_t y_stride, int64_t m, int64_t n, int64_t k, void * workspace, size_t workspace_bytes, int n_cu, hipStream_t stream, bool prepare) {
    if (m <= 0) return mmq_i8_supported(type, k) ? MI355Q_OK : MI355Q_ERR_UNSUPPORTED;
    const mi355q_mat mt = { type, w, w_stride, y, y_stride, m };
    return launch_mmq_i8_multi(&mt, 1, x, x_stride, n, k, workspace, workspace_bytes, n_cu, stream, prepare);
}

} // namespace mi355q
