// mmq_q80.hip -- the batched (prefill) tier for Q8_0 on the INTEGER matrix cores, with the CPU's arithmetic: y[n][m] = W_q8_0[m][k] . x[n][k], n > 8.
//
// The CPU backend quantizes the activations to Q8_0 (32-element blocks, f16 scale; quantize_row_q8_0, ggml-quants.c:194-217) and multiplies block by
// block (ggml_vec_dot_q8_0_q8_0, ggml-cpu-quants.c scalar tail):      sumf += (float) sumi * (d_w * d_x),   sumi = sum_{j<32} qw_j * qx_j  (exact int32),
// blocks in ascending order.  This kernel evaluates exactly that: the same activation quantizer (act_quant.cuh q80_group8, bit-equal to the CPU's, both
// rounding rules), one v_mfma_i32_16x16x32_i8 per 32-block and 16 x 16 output tile for the exact integer sums, then the CPU's two f32 operations per
// block in the CPU's block order.  The result equals the scalar CPU backend BIT FOR BIT for any input -- which is what north_star asks of the Q8_0
// configuration ("bit-exact for Q8_0 integer dot") also at prefill sizes, where round 1 used the bf16 tier (NMSE 1e-5 class).
// Replaces the reference's mul_mat_q Q8_0 tiles (ggml-cuda/mmq.cuh:563-760, dp4a on CDNA) / its dequantize + hipBLAS route.
//
// Data flow (as mmq_i8.hip): a wave owns RT*16 weight rows and reads them from global memory straight into MFMA B-operand registers (8 bytes per
// lane and block: lane = (row l16, k-quarter kq)); the quantized activation tile of the 256-k step (BN tokens x 256 int8 + 8 block scales) is DMA-copied
// global -> LDS (global_load_lds_dwordx4, double buffered) and read as A operands.  Workgroup = 4 waves = (64*RT) rows x (16*TT) tokens.
// Bound: vector issue, not the matrix pipe -- every 16x16x32 MFMA (16 cycles) is followed by 4 conversions and 12 f32 operations per lane, which is
// the price of applying both block scales in the CPU's order; HBM: packed W once per token tile.
#include "mi355q_common.h"
#include "act_quant.cuh"

namespace mi355q {

typedef __attribute__((ext_vector_type(4))) float  q80_f32x4;
typedef __attribute__((ext_vector_type(2))) float  q80_f32x2;
struct Q80Acc { q80_f32x2 lo, hi; };
typedef __attribute__((ext_vector_type(4))) int    q80_i32x4;
typedef __attribute__((address_space(3))) void * q80_lds_ptr;

// activation image per 256-k step b and token n (tokens padded to a multiple of 128):
//   XQ [nb256][n_pad][272 B] : 8 blocks x 32 int8 in natural order + 16 B of padding (LDS bank spread); the record IS the LDS row
//   XS [nb256][n_pad/64][8 blocks][64 tokens] f32 : the block scales (already rounded to f16, as the CPU stores them), block-major inside a
//      64-token tile so that one 16-byte LDS read gives a lane the scales of its 4 C-row tokens of one block -- as two register pairs for v_pk_mul_f32
constexpr int Q80_REC = 272, Q80_XS = 32, Q80_TOK_ALIGN = 128, Q80_XS_TILE = 64;
__host__ __device__ inline int64_t q80_npad(int64_t n) { return (n + Q80_TOK_ALIGN - 1) / Q80_TOK_ALIGN * Q80_TOK_ALIGN; }

template <bool ROUND_EVEN>
__global__ void __launch_bounds__(256)
k_mmq_q80_prep(const float * __restrict__ x, int64_t x_stride, uint8_t * __restrict__ xq, uint8_t * __restrict__ xs, int n, int n_pad, int nb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x, tok = 4 * blockIdx.y + wave;
    if (tok >= n_pad) return;
    uint8_t * rec = xq + ((int64_t) b * n_pad + tok) * Q80_REC;
    float * srec = (float *) (xs + ((int64_t) b * n_pad + (tok & ~(Q80_XS_TILE - 1))) * Q80_XS) + (tok & (Q80_XS_TILE - 1));
    uint32_t q = 0; float d = 0.0f; int sum = 0;
    if (tok < n) {
        const float4 v = *(const float4 *) ((const char *) x + (int64_t) tok * x_stride + 4 * (256 * (int64_t) b + 4 * lane));
        q80_group8<ROUND_EVEN>(v, q, d, sum);
        d = __half2float(__float2half_rn(d));
    }
    *(uint32_t *) (rec + 4 * lane) = q;
    if (lane < 4) *(uint32_t *) (rec + 256 + 4 * lane) = 0u;
    if ((lane & 7) == 0) srec[Q80_XS_TILE * (lane >> 3)] = d;
}

struct Q80Mats { const uint8_t * w[4]; int64_t w_stride[4]; float * y[4]; int64_t y_stride[4]; int m[4]; int rb_begin[4]; };

template <int RT, int TT, int NW>
__global__ void __launch_bounds__(64 * NW, RT == 1 ? 16 / NW : 8 / NW)
k_mmq_q80(const Q80Mats mats, const uint8_t * __restrict__ xq, const uint8_t * __restrict__ xs, int n, int n_pad, int nb /* 256-k steps */,
          int n_tok_tiles, int total_tiles, int per_xcd, const MoeTiles moe) {
    constexpr int BN = 16 * TT, BM = 16 * NW * RT;
    constexpr int XQ_BYTES = BN * Q80_REC, XS_BYTES = BN * Q80_XS, BUF = XQ_BYTES + XS_BYTES;
    constexpr int XQ_PIECES = XQ_BYTES / 1024, XS_PIECES = (XS_BYTES + 1023) / 1024;
    static_assert(XQ_BYTES % 1024 == 0, "DMA pieces");
    static_assert(BN == Q80_XS_TILE, "the scale image is laid out per 64-token tile");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    // workgroups that share weight rows (the token tiles of one row block) are consecutive on ONE XCD: its L2 serves the re-reads
    const int v = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (v >= total_tiles) return;
    const int rb_all = v / n_tok_tiles, tt0 = v - rb_all * n_tok_tiles;
    const int mi = (rb_all >= mats.rb_begin[1]) + (rb_all >= mats.rb_begin[2]) + (rb_all >= mats.rb_begin[3]);
    const uint8_t * __restrict__ w = mi == 0 ? mats.w[0] : mi == 1 ? mats.w[1] : mi == 2 ? mats.w[2] : mats.w[3];
    if (moe.tile_expert) {                                    // grouped MUL_MAT_ID: this token tile's expert (uniform per workgroup)
        const int e = moe.tile_expert[tt0 * BN / moe.tile_tokens];
        if (e < 0) return;
        w += (int64_t) e * moe.expert_stride;
        n = moe.seg_end[e];
    }
    const int64_t w_stride = mi == 0 ? mats.w_stride[0] : mi == 1 ? mats.w_stride[1] : mi == 2 ? mats.w_stride[2] : mats.w_stride[3];
    float * __restrict__ y = mi == 0 ? mats.y[0] : mi == 1 ? mats.y[1] : mi == 2 ? mats.y[2] : mats.y[3];
    const int64_t y_stride = mi == 0 ? mats.y_stride[0] : mi == 1 ? mats.y_stride[1] : mi == 2 ? mats.y_stride[2] : mats.y_stride[3];
    const int m = mi == 0 ? mats.m[0] : mi == 1 ? mats.m[1] : mi == 2 ? mats.m[2] : mats.m[3];
    const int rb = rb_all - (mi == 0 ? 0 : mi == 1 ? mats.rb_begin[1] : mi == 2 ? mats.rb_begin[2] : mats.rb_begin[3]);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l16 = lane & 15, kq = lane >> 4;
    const int m0 = rb * BM + wave * 16 * RT, n0 = tt0 * BN;
    const int nb32 = nb * 8;

    const uint8_t * wrow[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { const int r = m0 + 16 * rt + l16; wrow[rt] = w + (int64_t) (r < m ? r : 0) * w_stride; }

    Q80Acc facc[RT][TT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < TT; ++t) { facc[rt][t].lo = (q80_f32x2) { 0.f, 0.f }; facc[rt][t].hi = (q80_f32x2) { 0.f, 0.f }; }

    // packed weights of the step: block j of row rt -> this lane's 8 contraction slots (bytes 8*kq .. 8*kq+7 of the block); the 8 f16 block scales
    long wq[RT][8]; uint4 wd[RT];
    auto fetch_w = [&](int b) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const uint8_t * p = wrow[rt] + 256 * (int64_t) b + 8 * kq;
#pragma unroll
            for (int j = 0; j < 8; ++j) wq[rt][j] = *(const long *) (p + 32 * j);        // (kept in L2: the other token tiles of this row block read the same bytes)
            wd[rt] = ldg16(wrow[rt] + 32 * (int64_t) nb32 + 16 * b);
        }
    };
    auto dma_x = [&](int b, int buf) {                        // the activation tile of step b -> LDS buffer buf (lane-linear 1 KiB pieces)
        uint8_t * dst = lds + buf * BUF;
        const uint8_t * gq = xq + ((int64_t) b * n_pad + n0) * Q80_REC;
        const uint8_t * gs = xs + ((int64_t) b * n_pad + n0) * Q80_XS;
#pragma unroll
        for (int i = 0; i < (XQ_PIECES + NW - 1) / NW; ++i) {
            const int p = wave + NW * i;
            if (p < XQ_PIECES) __builtin_amdgcn_global_load_lds(gq + 1024 * p + 16 * lane, (q80_lds_ptr) (dst + 1024 * p), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < (XS_PIECES + NW - 1) / NW; ++i) {
            const int p = wave + NW * i;
            if (p < XS_PIECES && 1024 * p + 16 * lane < XS_BYTES) __builtin_amdgcn_global_load_lds(gs + 1024 * p + 16 * lane, (q80_lds_ptr) (dst + XQ_BYTES + 1024 * p), 16, 0, 0);
        }
    };

    fetch_w(0);
    dma_x(0, 0);
    for (int b = 0; b < nb; ++b) {
        const int buf = b & 1;
        __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0): this wave's share of tile b and its weights of step b have landed
        __syncthreads();
        // the step's operands move to fresh registers so that the loads of step b+1 can be issued before the MFMAs
        long bw[RT][8]; float dW[RT][8];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bw[rt][j] = wq[rt][j];
            const uint32_t dd[4] = { wd[rt].x, wd[rt].y, wd[rt].z, wd[rt].w };
#pragma unroll
            for (int j = 0; j < 8; ++j) dW[rt][j] = h2f((dd[j >> 1] >> (16 * (j & 1))) & 0xFFFFu);
        }
        if (b + 1 < nb) { fetch_w(b + 1); dma_x(b + 1, buf ^ 1); }
        __builtin_amdgcn_sched_barrier(0);

        const uint8_t * xq_l = lds + buf * BUF, * xs_l = xq_l + XQ_BYTES;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            __builtin_amdgcn_sched_barrier(0);                  // one token tile at a time: hoisting the next tiles' LDS reads costs 32 registers each
            const uint8_t * trow = xq_l + (16 * t + l16) * Q80_REC + 8 * kq;
            const float * sp = (const float *) xs_l + 16 * t + 4 * kq;       // the block scales of this lane's 4 C-row tokens (16 t + 4 kq + r), block j at +64 j
#pragma unroll
            for (int j = 0; j < 8; ++j) {                     // blocks in ascending order, as the CPU adds them
                const long af = *(const long *) (trow + 32 * j);
                const q80_f32x4 dx = *(const q80_f32x4 *) (sp + Q80_XS_TILE * j);
                const q80_f32x2 dx01 = { dx[0], dx[1] }, dx23 = { dx[2], dx[3] };
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const q80_i32x4 is = __builtin_amdgcn_mfma_i32_16x16x32_i8(af, bw[rt][j], (q80_i32x4) { 0, 0, 0, 0 }, 0, 0, 0);
                    // sumf += sumi * (d_w * d_x): three correctly rounded f32 operations per output, two outputs per packed instruction
                    const q80_f32x2 dw2 = { dW[rt][j], dW[rt][j] };
                    const q80_f32x2 s01 = { (float) is[0], (float) is[1] }, s23 = { (float) is[2], (float) is[3] };
                    facc[rt][t].lo = facc[rt][t].lo + s01 * (dw2 * dx01);
                    facc[rt][t].hi = facc[rt][t].hi + s23 * (dw2 * dx23);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: lane holds 4 tokens x 1 weight row per tile ----
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int row = m0 + 16 * rt + l16;
        if (row >= m) continue;
        char * ycol = (char *) y + 4 * (int64_t) row;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tok = n0 + 16 * t + 4 * kq + r;
                if (tok < n) {
                    const int dr = moe.dst_row ? moe.dst_row[tok] : tok;      // grouped MUL_MAT_ID: straight to the pair's row of the result
                    if (dr >= 0) *(float *) (ycol + (int64_t) dr * y_stride) = r < 2 ? facc[rt][t].lo[r & 1] : facc[rt][t].hi[r & 1];
                }
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------
bool mmq_q80_supported(int type, int64_t k) { return type == MI355Q_TYPE_Q8_0 && k % 256 == 0; }

size_t mmq_q80_workspace(int64_t n, int64_t k) {
    return (size_t) (q80_npad(n) * (k / 256) * (Q80_REC + Q80_XS) + 1024 + 255) & ~(size_t) 255;
}

// One grid for up to 4 Q8_0 matrices on the same activations.  w: planar device rows; x f32 [n][k] (16-byte aligned rows);
// workspace >= mmq_q80_workspace(n,k); y_i f32 [n][m_i].  prepare: write the activation image first.
int launch_mmq_q80_multi(const mi355q_mat * mt, int n_mats, const float * x, int64_t x_stride, int64_t n, int64_t k,
                         void * workspace, size_t workspace_bytes, int n_cu, hipStream_t stream, bool prepare, bool round_even, const MoeTiles * moe_p = nullptr) {
    MoeTiles moe = {}; if (moe_p) moe = *moe_p;
    if (n_mats < 1 || n_mats > 4) return MI355Q_ERR_SHAPE;
    for (int i = 0; i < n_mats; ++i) if (!mmq_q80_supported(mt[i].type, k)) return MI355Q_ERR_UNSUPPORTED;
    if (n <= 0) return MI355Q_OK;
    if ((x_stride & 15) || ((uintptr_t) x & 15)) return MI355Q_ERR_ALIGN;
    if (workspace_bytes < mmq_q80_workspace(n, k)) return MI355Q_ERR_WORKSPACE;
    const int nb = (int) (k / 256);
    const int64_t n_pad = q80_npad(n);
    uint8_t * xq = (uint8_t *) workspace;
    uint8_t * xs = xq + (size_t) nb * n_pad * Q80_REC;
    if (prepare) {
        if (round_even) hipLaunchKernelGGL(k_mmq_q80_prep<true>, dim3((unsigned) nb, (unsigned) (n_pad / 4)), dim3(256), 0, stream, x, x_stride, xq, xs, (int) n, (int) n_pad, nb);
        else            hipLaunchKernelGGL(k_mmq_q80_prep<false>, dim3((unsigned) nb, (unsigned) (n_pad / 4)), dim3(256), 0, stream, x, x_stride, xq, xs, (int) n, (int) n_pad, nb);
    }
    (void) n_cu;
#define MI355Q_Q80_LAUNCH(RT, TT, NW) {                                                                                                \
        constexpr int bn = 16 * TT, bm = 16 * NW * RT;                                                                                  \
        const size_t lds_bytes = 2 * (size_t) (bn * (Q80_REC + Q80_XS));                                                           \
        static bool attr_set[64] = {};                                                                                             \
        int dev_ = 0; (void) hipGetDevice(&dev_); dev_ = dev_ >= 0 && dev_ < 64 ? dev_ : 0;                                          \
        if (!attr_set[dev_]) {                                                                                                     \
            if (hipFuncSetAttribute((const void *) k_mmq_q80<RT, TT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes) != hipSuccess) \
                return MI355Q_ERR_HIP;                                                                                             \
            attr_set[dev_] = true;                                                                                                 \
        }                                                                                                                          \
        Q80Mats im; int rbs = 0;                                                                                                   \
        for (int i = 0; i < 4; ++i) {                                                                                              \
            const int j = i < n_mats ? i : 0;                                                                                      \
            im.w[i] = (const uint8_t *) mt[j].w; im.w_stride[i] = mt[j].w_stride; im.y[i] = mt[j].y; im.y_stride[i] = mt[j].y_stride; im.m[i] = (int) mt[j].m; \
            im.rb_begin[i] = i < n_mats ? rbs : 0x7FFFFFFF;                                                                        \
            if (i < n_mats) rbs += (int) ((mt[i].m + bm - 1) / bm);                                                                \
        }                                                                                                                          \
        const int n_tok_tiles = (int) (n_pad / bn), total = rbs * n_tok_tiles, per_xcd = (total + 7) / 8;                         \
        if (total > 0)                                                                                                             \
            hipLaunchKernelGGL((k_mmq_q80<RT, TT, NW>), dim3((unsigned) (8 * per_xcd)), dim3(64 * NW), lds_bytes, stream, im,              \
                               (const uint8_t *) xq, (const uint8_t *) xs, (int) n, (int) n_pad, nb, n_tok_tiles, total, per_xcd, moe); }
    // 128 rows x 64 tokens when that fills the chip, else 64 x 64 (a 128-token tile's accumulators leave no room for the operands: 48 spilled
    // registers when tried); a grouped launch's segments are aligned to 64 or 128 tokens, both multiples of the token tile
    // 64 rows x 64 tokens per 4-wave workgroup, 4 workgroups per CU.  Measured at N = 512 (TFLOP/s for 14336x4096 / 4096x4096 / 4096x14336):
    // this shape 321 / 246 / 283; 128 rows (two row tiles per wave, 176 VGPRs, 2 waves per SIMD) 275 / 195 / -; 8-wave workgroups sharing the
    // activation tile 303 / 233 / 263: occupancy, not operand traffic, is what the per-block f32 chain needs.
    static const int forced = getenv("MI355Q_Q80_CFG") ? atoi(getenv("MI355Q_Q80_CFG")) : 0;       // dev: 14 / 24 (row tiles per wave, token tiles)
    if (forced == 24) MI355Q_Q80_LAUNCH(2, 4, 4)
    else MI355Q_Q80_LAUNCH(1, 4, 4)
#undef MI355Q_Q80_LAUNCH
    return hipGetLastError() == hipSuccess ? MI355Q_OK : MI355Q_ERR_HIP;
}

} // namespace mi355q
