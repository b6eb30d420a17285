// mmq_generic.hip -- batched (prefill) tier for the weight types that keep their CANONICAL ggml rows on the device (no planar layout, no streaming
// kernel): Q4_1, Q5_0, Q5_1, Q2_K, Q3_K, IQ2_XXS, IQ2_XS, IQ2_S, IQ3_XXS, IQ3_S, IQ1_S, IQ1_M -- and Q4_0 / Q8_0 / IQ4_NL rows whose K keeps them
// canonical.  y[n][m] = W[m][k] . x[n][k] for n > 8.
//
// Until round 2 these types ran one GEMV column per token at any batch size (k_gemv_generic): a 512-token prompt re-read and re-decoded every
// weight 512 times.  Here a workgroup decodes a 64-row x 128-k weight tile ONCE per 64 tokens into bf16 in LDS -- the values of dequantize_row_<type>
// (ggml-quants.c; restated in oracle/orc_core.c, oracle/orc_iq.c), each computed as the CPU computes it and then rounded to bf16 -- and multiplies
// it with the bf16 activations on v_mfma_f32_16x16x32_bf16 (f32 accumulate), exactly the arithmetic of the planar bf16 tier (mmq_bf16.hip):
// NMSE against the exact product <= 2e-5, against the CPU backend <= 5e-4 (the reference's op bound, tests/test-backend-ops.cpp:1990).
// Not a roofline path: the decode reads the canonical blocks with byte loads and looks the code-book types up in device memory (csrc/iq_tables.h);
// what it buys is the 64-fold reuse of every decoded weight.
#include "mi355q_common.h"
#include "iq_tables.h"

namespace mi355q {

typedef __attribute__((ext_vector_type(8))) __bf16 gq_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 gq_bf16x2;
typedef __attribute__((ext_vector_type(2))) float  gq_f32x2;
typedef __attribute__((ext_vector_type(4))) float  gq_f32x4;

__device__ __constant__ int8_t c_gq_iq4[16] = { -127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113 };

__device__ __forceinline__ float gq_h(const uint8_t * p) { return h2f((uint32_t) p[0] | ((uint32_t) p[1] << 8)); }
__device__ __forceinline__ uint32_t gq_u32(const uint8_t * p) { return (uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16) | ((uint32_t) p[3] << 24); }
__device__ __forceinline__ float gq_sgn(uint32_t s, int t) { return ((s >> t) & 1u) ? -1.0f : 1.0f; }

// The 32 weights of sub-block s (32 consecutive k) of one canonical row, as dequantize_row_<type> produces them.
template <int TYPE>
__device__ __forceinline__ void deq32(const uint8_t * wrow, int64_t s, float * o) {
    if constexpr (TYPE == MI355Q_TYPE_Q4_0) {                        // ggml-quants.c:255-273
        const uint8_t * w = wrow + s * 18; const float d = gq_h(w);
#pragma unroll
        for (int l = 0; l < 16; ++l) { o[l] = (float) ((w[2 + l] & 0x0F) - 8) * d; o[l + 16] = (float) ((w[2 + l] >> 4) - 8) * d; }
    } else if constexpr (TYPE == MI355Q_TYPE_Q4_1) {                 // :275-294
        const uint8_t * w = wrow + s * 20; const float d = gq_h(w), m = gq_h(w + 2);
#pragma unroll
        for (int l = 0; l < 16; ++l) { o[l] = (float) (w[4 + l] & 0x0F) * d + m; o[l + 16] = (float) (w[4 + l] >> 4) * d + m; }
    } else if constexpr (TYPE == MI355Q_TYPE_Q5_0 || TYPE == MI355Q_TYPE_Q5_1) {   // :296-345
        constexpr bool one = TYPE == MI355Q_TYPE_Q5_1;
        const uint8_t * w = wrow + s * (one ? 24 : 22);
        const float d = gq_h(w), m = one ? gq_h(w + 2) : 0.0f;
        const uint32_t qh = gq_u32(w + (one ? 4 : 2)); const uint8_t * qs = w + (one ? 8 : 6);
#pragma unroll
        for (int l = 0; l < 16; ++l) {
            const int x0 = (qs[l] & 0x0F) | (int) (((qh >> l) << 4) & 0x10);
            const int x1 = (qs[l] >> 4)   | (int) ((qh >> (l + 12)) & 0x10);
            if constexpr (one) { o[l] = (float) x0 * d + m; o[l + 16] = (float) x1 * d + m; }
            else               { o[l] = (float) (x0 - 16) * d; o[l + 16] = (float) (x1 - 16) * d; }
        }
    } else if constexpr (TYPE == MI355Q_TYPE_Q8_0) {                 // :347-363
        const uint8_t * w = wrow + s * 34; const float d = gq_h(w);
#pragma unroll
        for (int l = 0; l < 32; ++l) o[l] = (float) (int8_t) w[2 + l] * d;
    } else if constexpr (TYPE == MI355Q_TYPE_IQ4_NL) {               // :2436-2452
        const uint8_t * w = wrow + s * 18; const float d = gq_h(w);
#pragma unroll
        for (int l = 0; l < 16; ++l) { o[l] = d * (float) c_gq_iq4[w[2 + l] & 0x0F]; o[l + 16] = d * (float) c_gq_iq4[w[2 + l] >> 4]; }
    } else {
        const int64_t b = s >> 3; const int j = (int) (s & 7);      // super-block b, its 32-element sub-block j = 4 h + c
        const int h = j >> 2, c = j & 3;
        if constexpr (TYPE == MI355Q_TYPE_Q2_K) {                    // :712-745   block: scales[16] qs[64] d dmin
            const uint8_t * w = wrow + b * 84; const uint8_t * qs = w + 16 + 32 * h;
            const float d = gq_h(w + 80), mn = gq_h(w + 82);
            const int s0 = w[2 * j], s1 = w[2 * j + 1];
            const float dl0 = d * (float) (s0 & 0x0F), ml0 = mn * (float) (s0 >> 4), dl1 = d * (float) (s1 & 0x0F), ml1 = mn * (float) (s1 >> 4);
#pragma unroll
            for (int l = 0; l < 32; ++l) { const float q = (float) ((qs[l] >> (2 * c)) & 3); o[l] = l < 16 ? dl0 * q - ml0 : dl1 * q - ml1; }
        } else if constexpr (TYPE == MI355Q_TYPE_Q3_K) {             // :1056-1100 block: hmask[32] qs[64] scales[12] d
            const uint8_t * w = wrow + b * 110; const uint8_t * qs = w + 32 + 32 * h; const uint8_t * hm = w; const uint8_t * sp = w + 96;
            const float d = gq_h(w + 108);
            float dl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int jj = 2 * j + t;
                const int lo = jj < 8 ? (sp[jj] & 0x0F) : (sp[jj - 8] >> 4);
                const int hi = (sp[8 + (jj & 3)] >> (2 * (jj >> 2))) & 3;
                dl[t] = d * (float) ((lo | (hi << 4)) - 32);
            }
#pragma unroll
            for (int l = 0; l < 32; ++l) {
                const int q = ((qs[l] >> (2 * c)) & 3) - (((hm[l] >> (4 * h + c)) & 1) ? 0 : 4);
                o[l] = dl[l >> 4] * (float) q;
            }
        } else if constexpr (TYPE == MI355Q_TYPE_IQ2_XXS) {          // :2197-2222 block: d qs u16[32]
            const uint8_t * w = wrow + b * 66; const uint8_t * q2 = w + 2 + 8 * j;
            const uint32_t aux1 = gq_u32(q2 + 4);
            const float db = gq_h(w) * (0.5f + (float) (aux1 >> 28)) * 0.25f;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const uint8_t * g = c_grid_iq2xxs[q2[l]]; const uint32_t sg = c_signs_iq2[(aux1 >> (7 * l)) & 127];
#pragma unroll
                for (int t = 0; t < 8; ++t) o[8 * l + t] = db * (float) g[t] * gq_sgn(sg, t);
            }
        } else if constexpr (TYPE == MI355Q_TYPE_IQ2_XS) {           // :2225-2248 block: d qs u16[32] scales[8]
            const uint8_t * w = wrow + b * 74; const uint8_t * q2 = w + 2 + 8 * j; const int sc = w[66 + j];
            const float d = gq_h(w);
            const float db0 = d * (0.5f + (float) (sc & 0xf)) * 0.25f, db1 = d * (0.5f + (float) (sc >> 4)) * 0.25f;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const uint32_t q = (uint32_t) q2[2 * l] | ((uint32_t) q2[2 * l + 1] << 8);
                const uint8_t * g = c_grid_iq2xs[q & 511]; const uint32_t sg = c_signs_iq2[q >> 9];
#pragma unroll
                for (int t = 0; t < 8; ++t) o[8 * l + t] = (l < 2 ? db0 : db1) * (float) g[t] * gq_sgn(sg, t);
            }
        } else if constexpr (TYPE == MI355Q_TYPE_IQ2_S) {            // :2252-2280 block: d qs[32] signs[32] qh[8] scales[8]
            const uint8_t * w = wrow + b * 82; const uint8_t * qs = w + 2 + 4 * j; const uint8_t * sgp = w + 34 + 4 * j;
            const int qh = w[66 + j], sc = w[74 + j];
            const float d = gq_h(w);
            const float db0 = d * (0.5f + (float) (sc & 0xf)) * 0.25f, db1 = d * (0.5f + (float) (sc >> 4)) * 0.25f;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const uint8_t * g = c_grid_iq2s[qs[l] | ((qh << (8 - 2 * l)) & 0x300)];
#pragma unroll
                for (int t = 0; t < 8; ++t) o[8 * l + t] = (l < 2 ? db0 : db1) * (float) g[t] * gq_sgn(sgp[l], t);
            }
        } else if constexpr (TYPE == MI355Q_TYPE_IQ3_XXS) {          // :2284-2312 block: d qs[64] scales_and_signs[32]
            const uint8_t * w = wrow + b * 98; const uint8_t * q3 = w + 2 + 8 * j;
            const uint32_t aux = gq_u32(w + 66 + 4 * j);
            const float db = gq_h(w) * (0.5f + (float) (aux >> 28)) * 0.5f;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const uint8_t * g1 = c_grid_iq3xxs[q3[2 * l]]; const uint8_t * g2 = c_grid_iq3xxs[q3[2 * l + 1]];
                const uint32_t sg = c_signs_iq2[(aux >> (7 * l)) & 127];
#pragma unroll
                for (int t = 0; t < 4; ++t) { o[8 * l + t] = db * (float) g1[t] * gq_sgn(sg, t); o[8 * l + t + 4] = db * (float) g2[t] * gq_sgn(sg, t + 4); }
            }
        } else if constexpr (TYPE == MI355Q_TYPE_IQ3_S) {            // :2316-2357 block: d qs[64] qh[8] signs[32] scales[4]
            const uint8_t * w = wrow + b * 110; const uint8_t * qs = w + 2 + 8 * j; const int qh = w[66 + j]; const uint8_t * sgp = w + 74 + 4 * j;
            const int nib = (j & 1) ? (w[106 + (j >> 1)] >> 4) : (w[106 + (j >> 1)] & 0xf);
            const float db = gq_h(w) * (float) (1 + 2 * nib);
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const uint8_t * g1 = c_grid_iq3s[qs[2 * l] | ((qh << (8 - 2 * l)) & 256)]; const uint8_t * g2 = c_grid_iq3s[qs[2 * l + 1] | ((qh << (7 - 2 * l)) & 256)];
#pragma unroll
                for (int t = 0; t < 4; ++t) { o[8 * l + t] = db * (float) g1[t] * gq_sgn(sgp[l], t); o[8 * l + t + 4] = db * (float) g2[t] * gq_sgn(sgp[l], t + 4); }
            }
        } else if constexpr (TYPE == MI355Q_TYPE_IQ1_S) {            // :2359-2382 block: d qs[32] qh u16[8]
            const uint8_t * w = wrow + b * 50; const uint8_t * qs = w + 2 + 4 * j;
            const uint32_t qh = (uint32_t) w[34 + 2 * j] | ((uint32_t) w[35 + 2 * j] << 8);
            const float dl = gq_h(w) * (float) (2 * (int) ((qh >> 12) & 7) + 1), delta = (qh & 0x8000) ? -0.125f : 0.125f;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int8_t * g = c_grid_iq1s[qs[l] | (((qh >> (3 * l)) & 7) << 8)];
#pragma unroll
                for (int t = 0; t < 8; ++t) o[8 * l + t] = dl * ((float) g[t] + delta);
            }
        } else {                                                     // IQ1_M  :2384-2432 block: qs[32] qh[16] scales[8]
            static_assert(TYPE == MI355Q_TYPE_IQ1_M, "unhandled type");
            const uint8_t * w = wrow + b * 56; const uint8_t * qs = w + 4 * j; const uint8_t * qh = w + 32 + 2 * j;
            uint32_t sc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) sc[t] = (uint32_t) w[48 + 2 * t] | ((uint32_t) w[49 + 2 * t] << 8);
            const float d = h2f((sc[0] >> 12) | ((sc[1] >> 8) & 0x00f0u) | ((sc[2] >> 4) & 0x0f00u) | (sc[3] & 0xf000u));
            const float dl1 = d * (float) (2 * (int) ((sc[j / 2] >> (6 * (j % 2) + 0)) & 0x7) + 1), dl2 = d * (float) (2 * (int) ((sc[j / 2] >> (6 * (j % 2) + 3)) & 0x7) + 1);
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int8_t * g = c_grid_iq1s[qs[l] | (((uint32_t) qh[l / 2] << (8 - 4 * (l % 2))) & 0x700)];
                const float delta = (qh[l / 2] & (0x08 << (4 * (l % 2)))) ? -0.125f : 0.125f, dl = l < 2 ? dl1 : dl2;
#pragma unroll
                for (int t = 0; t < 8; ++t) o[8 * l + t] = dl * ((float) g[t] + delta);
            }
        }
    }
}

constexpr int GQ_BK = 128, GQ_BM = 64, GQ_BN = 64, GQ_STRIDE = (GQ_BK + 8) * 2;      // bytes per LDS row (as mmq_bf16.hip)

__device__ __forceinline__ uint32_t gq_pack(float a, float b) {
    const gq_f32x2 v = { a, b };
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, gq_bf16x2));       // v_cvt_pk_bf16_f32, round to nearest even
}

// 256 threads = 2 x 2 waves of 32 x 32 outputs; thread t stages unit (row t >> 2, 32-k quarter t & 3) of W and of X per 128-k step
template <int TYPE>
__global__ void __launch_bounds__(256, 2)
k_mmq_generic(const uint8_t * __restrict__ w, int64_t w_stride, const uint16_t * __restrict__ xb /* bf16 [n][k] */, float * __restrict__ y, int64_t y_stride,
              int m, int n, int k, const MoeTiles moe) {
    __shared__ __attribute__((aligned(16))) uint8_t Ws[GQ_BM * GQ_STRIDE];
    __shared__ __attribute__((aligned(16))) uint8_t Xs[GQ_BN * GQ_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int quarter = tid & 3, r = tid >> 2;
    const int m0 = blockIdx.x * GQ_BM, n0 = blockIdx.y * GQ_BN;
    if (moe.tile_expert) {                                    // grouped MUL_MAT_ID: this token tile's expert (uniform per workgroup), as the other tiers
        const int e = moe.tile_expert[n0 / moe.tile_tokens];
        if (e < 0) return;
        w += (int64_t) e * moe.expert_stride;
        n = moe.seg_end[e];
        if (n0 >= n) return;
    }
    const uint8_t * wrow = w + (int64_t) (m0 + r < m ? m0 + r : 0) * w_stride;
    const uint16_t * xrow = xb + (int64_t) (n0 + r < n ? n0 + r : 0) * k + 32 * quarter;
    gq_f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (gq_f32x4) { 0.f, 0.f, 0.f, 0.f };
    const int steps = k / GQ_BK;
    for (int ks = 0; ks < steps; ++ks) {
        float v[32];
        deq32<TYPE>(wrow, (int64_t) ks * 4 + quarter, v);
        uint32_t p[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) p[i] = gq_pack(v[2 * i], v[2 * i + 1]);
        const uint16_t * xp = xrow + (int64_t) ks * GQ_BK;
        const uint4 x0 = *(const uint4 *) xp, x1 = *(const uint4 *) (xp + 8), x2 = *(const uint4 *) (xp + 16), x3 = *(const uint4 *) (xp + 24);
        __syncthreads();                                        // the previous step's MFMAs have read the tiles
        uint8_t * wd = Ws + r * GQ_STRIDE + 64 * quarter;
        *(uint4 *) wd = make_uint4(p[0], p[1], p[2], p[3]);          *(uint4 *) (wd + 16) = make_uint4(p[4], p[5], p[6], p[7]);
        *(uint4 *) (wd + 32) = make_uint4(p[8], p[9], p[10], p[11]); *(uint4 *) (wd + 48) = make_uint4(p[12], p[13], p[14], p[15]);
        uint8_t * xd = Xs + r * GQ_STRIDE + 64 * quarter;
        *(uint4 *) xd = x0; *(uint4 *) (xd + 16) = x1; *(uint4 *) (xd + 32) = x2; *(uint4 *) (xd + 48) = x3;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GQ_BK / 32; ++kk) {
            const int koff = 2 * (32 * kk + 8 * (lane >> 4));
            gq_bf16x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *(const gq_bf16x8 *) (Ws + (32 * wm + 16 * i + (lane & 15)) * GQ_STRIDE + koff);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *(const gq_bf16x8 *) (Xs + (32 * wn + 16 * j + (lane & 15)) * GQ_STRIDE + koff);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    // C/D layout: col = lane & 15 (token), rows 4 (lane >> 4) + reg (weight row)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tok = n0 + 32 * wn + 16 * j + (lane & 15);
        if (tok >= n) continue;
        const int dr = moe.dst_row ? moe.dst_row[tok] : tok;                  // grouped MUL_MAT_ID: straight to the pair's row of the result
        if (dr < 0) continue;
        float * yr = (float *) ((char *) y + (int64_t) dr * y_stride);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mr = m0 + 32 * wm + 16 * i + 4 * (lane >> 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) if (mr + e < m) yr[mr + e] = acc[i][j][e];
        }
    }
}

// canonical rows of `type` with K a multiple of the 128-k step (and of the type's block)
bool mmq_generic_supported(int type, int64_t k) {
    switch (type) {
    case MI355Q_TYPE_Q4_0: case MI355Q_TYPE_Q4_1: case MI355Q_TYPE_Q5_0: case MI355Q_TYPE_Q5_1: case MI355Q_TYPE_Q8_0: case MI355Q_TYPE_IQ4_NL:
        return k % 128 == 0;
    case MI355Q_TYPE_Q2_K: case MI355Q_TYPE_Q3_K: case MI355Q_TYPE_IQ2_XXS: case MI355Q_TYPE_IQ2_XS: case MI355Q_TYPE_IQ2_S:
    case MI355Q_TYPE_IQ3_XXS: case MI355Q_TYPE_IQ3_S: case MI355Q_TYPE_IQ1_S: case MI355Q_TYPE_IQ1_M:
        return k % 256 == 0;
    default: return false;
    }
}
size_t mmq_generic_workspace(int64_t n, int64_t k) { return (size_t) (n * k * 2 + 255) & ~(size_t) 255; }

__global__ void __launch_bounds__(256) k_gq_x_to_bf16(const float * __restrict__ x, int64_t x_stride, uint32_t * __restrict__ out, int64_t n, int64_t k) {
    const int64_t pairs = k / 2;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n * pairs; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t r = i / pairs, c = i - r * pairs;
        const float * xr = (const float *) ((const char *) x + r * x_stride);
        out[i] = gq_pack(xr[2 * c], xr[2 * c + 1]);
    }
}

// w: CANONICAL device rows; x f32 [n][k] (row stride x_stride); workspace >= mmq_generic_workspace(n, k); y f32 [n][m]
int launch_mmq_generic(int type, const void * w, int64_t w_stride, const float * x, int64_t x_stride, float * y, int64_t y_stride,
                       int64_t m, int64_t n, int64_t k, void * workspace, size_t workspace_bytes, hipStream_t stream, const MoeTiles * moe_p = nullptr) {
    MoeTiles moe = {}; if (moe_p) moe = *moe_p;
    if (!mmq_generic_supported(type, k)) return MI355Q_ERR_UNSUPPORTED;
    if (m <= 0 || n <= 0) return MI355Q_OK;
    if (workspace_bytes < mmq_generic_workspace(n, k) || ((uintptr_t) workspace & 15)) return MI355Q_ERR_WORKSPACE;
    const int64_t pairs = n * k / 2;
    const int cgrid = (int) ((pairs + 255) / 256 < 8192 ? (pairs + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_gq_x_to_bf16, dim3(cgrid), dim3(256), 0, stream, x, x_stride, (uint32_t *) workspace, n, k);
    const dim3 grid((unsigned) ((m + GQ_BM - 1) / GQ_BM), (unsigned) ((n + GQ_BN - 1) / GQ_BN));
#define MI355Q_GQ_CASE(T) case T: hipLaunchKernelGGL((k_mmq_generic<T>), grid, dim3(256), 0, stream, (const uint8_t *) w, w_stride, (const uint16_t *) workspace, y, y_stride, (int) m, (int) n, (int) k, moe); break;
    switch (type) {
        MI355Q_GQ_CASE(MI355Q_TYPE_Q4_0) MI355Q_GQ_CASE(MI355Q_TYPE_Q4_1) MI355Q_GQ_CASE(MI355Q_TYPE_Q5_0) MI355Q_GQ_CASE(MI355Q_TYPE_Q5_1)
        MI355Q_GQ_CASE(MI355Q_TYPE_Q8_0) MI355Q_GQ_CASE(MI355Q_TYPE_IQ4_NL) MI355Q_GQ_CASE(MI355Q_TYPE_Q2_K) MI355Q_GQ_CASE(MI355Q_TYPE_Q3_K)
        MI355Q_GQ_CASE(MI355Q_TYPE_IQ2_XXS) MI355Q_GQ_CASE(MI355Q_TYPE_IQ2_XS) MI355Q_GQ_CASE(MI355Q_TYPE_IQ2_S) MI355Q_GQ_CASE(MI355Q_TYPE_IQ3_XXS)
        MI355Q_GQ_CASE(MI355Q_TYPE_IQ3_S) MI355Q_GQ_CASE(MI355Q_TYPE_IQ1_S) MI355Q_GQ_CASE(MI355Q_TYPE_IQ1_M)
    default: return MI355Q_ERR_UNSUPPORTED;
    }
#undef MI355Q_GQ_CASE
    return hipGetLastError() == hipSuccess ? MI355Q_OK : MI355Q_ERR_HIP;
}

} // namespace mi355q
