// gemv_generic.hip -- correctness tier: any supported type, any K (multiple of the block size), any
// row alignment.  Weights are in CANONICAL ggml block layout (ggml-common.h:167-418), activations
// are canonical block_q8_0 / block_q8_1 / block_q8_K rows produced by quantize_act.hip.
//
// One wave per weight row; lane l walks the 32-element sub-blocks l, l+64, ... of the row, decodes
// them with byte loads and forms the same integer sums as the CPU's scalar vec_dot
// (ggml-cpu-quants.c, scalar tails cited per type below).  Used for (type,K) pairs that have no
// planar fast kernel; it is NOT the roofline path (byte loads, no unrolling).
#include "mi355q_common.h"
#include "iq_tables.h"   // code books of the IQ2/IQ3/IQ1 formats, derived by probing the reference decoder (tools/gen_iq_tables.py)

namespace mi355q {

__device__ __constant__ int8_t c_iq4_codebook[16] = { -127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113 };

__device__ __forceinline__ float ld_h(const uint8_t * p) { return h2f((uint32_t) p[0] | ((uint32_t) p[1] << 8)); }
__device__ __forceinline__ float ld_f(const uint8_t * p) { float f; memcpy(&f, p, 4); return f; }
__device__ __forceinline__ int   ld_i16(const uint8_t * p) { return (int) (int16_t) ((uint32_t) p[0] | ((uint32_t) p[1] << 8)); }

__device__ __forceinline__ void k4_pair(const uint8_t * p, int j, int & sc, int & mn) {
    if (j < 4) { sc = p[j] & 63; mn = p[j + 4] & 63; }
    else       { sc = (p[j + 4] & 0x0F) | ((p[j - 4] >> 6) << 4); mn = (p[j + 4] >> 4) | ((p[j] >> 6) << 4); }
}

// Contribution of 32-element sub-block `s` (global index over the row) of one weight row against
// one activation row.  wrow / arow point at the start of the canonical rows.
template <int TYPE>
__device__ __forceinline__ float sub_dot(const uint8_t * wrow, const uint8_t * arow, int64_t s) {
    if constexpr (TYPE == MI355Q_TYPE_Q4_0) {                       // ggml-cpu-quants.c:2591-2605
        const uint8_t * w = wrow + s * 18; const uint8_t * a = arow + s * 34; const int8_t * y = (const int8_t *) a + 2;
        int s0 = 0, s1 = 0;
        for (int j = 0; j < 16; ++j) { s0 += ((w[2 + j] & 0x0F) - 8) * y[j]; s1 += ((w[2 + j] >> 4) - 8) * y[j + 16]; }
        return (float) (s0 + s1) * ld_h(w) * ld_h(a);
    } else if constexpr (TYPE == MI355Q_TYPE_Q4_1) {                // :2910-2924
        const uint8_t * w = wrow + s * 20; const uint8_t * a = arow + s * 36; const int8_t * y = (const int8_t *) a + 4;
        int s0 = 0, s1 = 0;
        for (int j = 0; j < 16; ++j) { s0 += (w[4 + j] & 0x0F) * y[j]; s1 += (w[4 + j] >> 4) * y[j + 16]; }
        return (ld_h(w) * ld_h(a)) * (float) (s0 + s1) + ld_h(w + 2) * ld_h(a + 2);
    } else if constexpr (TYPE == MI355Q_TYPE_Q5_0 || TYPE == MI355Q_TYPE_Q5_1) {   // :3228-3248, :3571-3591
        constexpr bool one = TYPE == MI355Q_TYPE_Q5_1;
        const uint8_t * w = wrow + s * (one ? 24 : 22); const uint8_t * a = arow + s * (one ? 36 : 34);
        const uint8_t * qhp = w + (one ? 4 : 2); const uint8_t * qs = qhp + 4;
        const int8_t * y = (const int8_t *) a + (one ? 4 : 2);
        const uint32_t qh = (uint32_t) qhp[0] | ((uint32_t) qhp[1] << 8) | ((uint32_t) qhp[2] << 16) | ((uint32_t) qhp[3] << 24);
        int s0 = 0, s1 = 0;
        for (int j = 0; j < 16; ++j) {
            const int x0 = (qs[j] & 0x0F) | (int) (((qh >> j) << 4) & 0x10);
            const int x1 = (qs[j] >> 4)   | (int) ((qh >> (j + 12)) & 0x10);
            s0 += (one ? x0 : x0 - 16) * y[j]; s1 += (one ? x1 : x1 - 16) * y[j + 16];
        }
        float r = (ld_h(w) * ld_h(a)) * (float) (s0 + s1);
        if constexpr (one) r += ld_h(w + 2) * ld_h(a + 2);
        return r;
    } else if constexpr (TYPE == MI355Q_TYPE_Q8_0) {                // :4004-4012
        const uint8_t * w = wrow + s * 34; const uint8_t * a = arow + s * 34;
        const int8_t * x = (const int8_t *) w + 2; const int8_t * y = (const int8_t *) a + 2;
        int sum = 0;
        for (int j = 0; j < 32; ++j) sum += x[j] * y[j];
        return (float) sum * (ld_h(w) * ld_h(a));
    } else if constexpr (TYPE == MI355Q_TYPE_IQ4_NL) {              // :12652-12660
        const uint8_t * w = wrow + s * 18; const uint8_t * a = arow + s * 34; const int8_t * y = (const int8_t *) a + 2;
        int s1 = 0, s2 = 0;
        for (int j = 0; j < 16; ++j) { s1 += y[j] * c_iq4_codebook[w[2 + j] & 0x0F]; s2 += y[j + 16] * c_iq4_codebook[w[2 + j] >> 4]; }
        return (ld_h(a) * ld_h(w)) * (float) (s1 + s2);
    } else {
        // ---- super-block types against Q8_K: block b, sub-block j (32 elements) ----
        const int64_t b = s >> 3; const int j = (int) (s & 7);
        const uint8_t * a  = arow + b * 292;
        const float     yd = ld_f(a);
        const int8_t *  y  = (const int8_t *) a + 4 + 32 * j;
        const int bs0 = ld_i16(a + 260 + 2 * (2 * j)), bs1 = ld_i16(a + 260 + 2 * (2 * j + 1));
        if constexpr (TYPE == MI355Q_TYPE_Q4_K || TYPE == MI355Q_TYPE_Q5_K) {   // :7535-7591, :8351-8412
            constexpr bool five = TYPE == MI355Q_TYPE_Q5_K;
            const uint8_t * w = wrow + b * (five ? 176 : 144);
            const uint8_t * qs = w + (five ? 48 : 16) + 32 * (j >> 1);
            int sc, mn; k4_pair(w + 4, j, sc, mn);
            int sum = 0;
            for (int l = 0; l < 32; ++l) {
                int q = (j & 1) ? (qs[l] >> 4) : (qs[l] & 0x0F);
                if constexpr (five) q += ((w[16 + l] >> j) & 1) ? 16 : 0;
                sum += q * y[l];
            }
            return (ld_h(w) * yd) * (float) (sc * sum) - (ld_h(w + 2) * yd) * (float) (mn * (bs0 + bs1));
        } else if constexpr (TYPE == MI355Q_TYPE_Q6_K) {            // :9423-9465
            const uint8_t * w = wrow + b * 210;
            const int h = j >> 2, c = j & 3;                         // element 128h + 32c + l
            const uint8_t * ql = w + 64 * h + 32 * (c & 1); const uint8_t * qh = w + 128 + 32 * h;
            const int8_t * scs = (const int8_t *) w + 192 + 8 * h + 2 * c;
            int sa = 0, sb = 0;
            for (int l = 0; l < 32; ++l) {
                const int lo = (c & 2) ? (ql[l] >> 4) : (ql[l] & 0x0F);
                const int q = (lo | (((qh[l] >> (2 * c)) & 3) << 4)) - 32;
                if (l < 16) sa += q * y[l]; else sb += q * y[l];
            }
            return (ld_h(w + 208) * yd) * (float) (scs[0] * sa + scs[1] * sb);
        } else if constexpr (TYPE == MI355Q_TYPE_Q3_K) {            // :6604-6661
            // blocks are 110 bytes: 2-byte aligned.  16-bit loads + v_dot4: q - 4 (1 - hbit) = (low2 | hbit << 2) - 4, and sum (v - 4) y = sum v y - 4 bsum
            const uint8_t * w = wrow + b * 110;
            const int h = j >> 2, c = j & 3;
            const uint16_t * qs = (const uint16_t *) (w + 32 + 32 * h); const uint16_t * hm = (const uint16_t *) w; const uint8_t * sp = w + 96;
            const uint32_t * y4 = (const uint32_t *) y;
            int sc[2];
            for (int t = 0; t < 2; ++t) {
                const int jj = 2 * j + t;
                const int lo = jj < 8 ? (sp[jj] & 0x0F) : (sp[jj - 8] >> 4);
                const int hi = (sp[8 + (jj & 3)] >> (2 * (jj >> 2))) & 3;
                sc[t] = (lo | (hi << 4)) - 32;
            }
            int sa = 0, sb = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t q4 = (uint32_t) qs[2 * i] | ((uint32_t) qs[2 * i + 1] << 16), h4 = (uint32_t) hm[2 * i] | ((uint32_t) hm[2 * i + 1] << 16);
                const uint32_t v = ((q4 >> (2 * c)) & 0x03030303u) | (((h4 >> (4 * h + c)) & 0x01010101u) << 2);
                if (i < 4) sa = __builtin_amdgcn_sdot4((int) v, (int) y4[i], sa, false); else sb = __builtin_amdgcn_sdot4((int) v, (int) y4[i], sb, false);
            }
            sa -= 4 * bs0; sb -= 4 * bs1;
            return (ld_h(w + 108) * yd) * (float) (sc[0] * sa + sc[1] * sb);
        } else if constexpr (TYPE == MI355Q_TYPE_Q2_K) {            // :5485-5523
            // blocks are 84 bytes (4-byte aligned rows): 32-bit loads + v_dot4 on the 2-bit fields spread to bytes
            const uint8_t * w = wrow + b * 84;
            const int h = j >> 2, c = j & 3;
            const uint32_t * qs4 = (const uint32_t *) (w + 16 + 32 * h);
            const uint32_t * y4 = (const uint32_t *) y;
            const int s0 = w[2 * j], s1 = w[2 * j + 1];
            int sa = 0, sb = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t v = (qs4[i] >> (2 * c)) & 0x03030303u;
                if (i < 4) sa = __builtin_amdgcn_sdot4((int) v, (int) y4[i], sa, false); else sb = __builtin_amdgcn_sdot4((int) v, (int) y4[i], sb, false);
            }
            const float dall = yd * ld_h(w + 80), dmin = yd * ld_h(w + 82);
            return dall * (float) ((s0 & 0x0F) * sa + (s1 & 0x0F) * sb) - dmin * (float) ((s0 >> 4) * bs0 + (s1 >> 4) * bs1);
        } else if constexpr (TYPE == MI355Q_TYPE_IQ2_XXS) {          // :9834-9862   block: d(2) qs u16[32]
            const uint8_t * w = wrow + b * 66; const uint8_t * q2 = w + 2 + 8 * j;
            const uint32_t aux1 = (uint32_t) q2[4] | ((uint32_t) q2[5] << 8) | ((uint32_t) q2[6] << 16) | ((uint32_t) q2[7] << 24);
            int sumi = 0;
            for (int l = 0; l < 4; ++l) {
                const uint8_t * g = c_grid_iq2xxs[q2[l]]; const uint32_t sg = c_signs_iq2[(aux1 >> (7 * l)) & 127];
                for (int t = 0; t < 8; ++t) sumi += (int) g[t] * y[8 * l + t] * (((sg >> t) & 1) ? -1 : 1);
            }
            return 0.125f * ((ld_h(w) * yd) * (float) (sumi * (2 * (int) (aux1 >> 28) + 1)));
        } else if constexpr (TYPE == MI355Q_TYPE_IQ2_XS) {           // :10411-10447 block: d(2) qs u16[32] scales[8]
            const uint8_t * w = wrow + b * 74; const uint8_t * q2 = w + 2 + 8 * j; const int sc = w[66 + j];
            int s1 = 0, s2 = 0;
            for (int l = 0; l < 4; ++l) {
                const uint32_t q = (uint32_t) q2[2 * l] | ((uint32_t) q2[2 * l + 1] << 8);
                const uint8_t * g = c_grid_iq2xs[q & 511]; const uint32_t sg = c_signs_iq2[q >> 9];
                int t0 = 0;
                for (int t = 0; t < 8; ++t) t0 += (int) g[t] * y[8 * l + t] * (((sg >> t) & 1) ? -1 : 1);
                if (l < 2) s1 += t0; else s2 += t0;
            }
            return 0.125f * ((ld_h(w) * yd) * (float) (s1 * (2 * (sc & 0xf) + 1) + s2 * (2 * (sc >> 4) + 1)));
        } else if constexpr (TYPE == MI355Q_TYPE_IQ2_S) {            // :10885-10921 block: d(2) qs[32] signs[32] qh[8] scales[8]
            const uint8_t * w = wrow + b * 82; const uint8_t * qs = w + 2 + 4 * j; const uint8_t * sg = w + 34 + 4 * j;
            const int qh = w[66 + j], sc = w[74 + j];
            int s1 = 0, s2 = 0;
            for (int l = 0; l < 4; ++l) {
                const uint8_t * g = c_grid_iq2s[qs[l] | ((qh << (8 - 2 * l)) & 0x300)];
                int t0 = 0;
                for (int t = 0; t < 8; ++t) t0 += y[8 * l + t] * (int) g[t] * (((sg[l] >> t) & 1) ? -1 : 1);
                if (l < 2) s1 += t0; else s2 += t0;
            }
            return 0.125f * ((ld_h(w) * yd) * (float) ((1 + 2 * (sc & 0xf)) * s1 + (1 + 2 * (sc >> 4)) * s2));
        } else if constexpr (TYPE == MI355Q_TYPE_IQ3_XXS) {          // :11218-11247 block: d(2) qs[64] scales_and_signs[32]
            const uint8_t * w = wrow + b * 98; const uint8_t * q3 = w + 2 + 8 * j; const uint8_t * gas = w + 66 + 4 * j;
            const uint32_t aux = (uint32_t) gas[0] | ((uint32_t) gas[1] << 8) | ((uint32_t) gas[2] << 16) | ((uint32_t) gas[3] << 24);
            int sumi = 0;
            for (int l = 0; l < 4; ++l) {
                const uint8_t * g1 = c_grid_iq3xxs[q3[2 * l]]; const uint8_t * g2 = c_grid_iq3xxs[q3[2 * l + 1]];
                const uint32_t sg = c_signs_iq2[(aux >> (7 * l)) & 127];
                for (int t = 0; t < 4; ++t) {
                    sumi += (int) g1[t] * y[8 * l + t]     * (((sg >> t) & 1) ? -1 : 1);
                    sumi += (int) g2[t] * y[8 * l + t + 4] * (((sg >> (t + 4)) & 1) ? -1 : 1);
                }
            }
            return 0.25f * ((ld_h(w) * yd) * (float) (sumi * (2 * (int) (aux >> 28) + 1)));
        } else if constexpr (TYPE == MI355Q_TYPE_IQ3_S) {            // :11733-11776 block: d(2) qs[64] qh[8] signs[32] scales[4]
            const uint8_t * w = wrow + b * 110; const uint8_t * qs = w + 2 + 8 * j; const int qh = w[66 + j]; const uint8_t * sg = w + 74 + 4 * j;
            const int nib = (j & 1) ? (w[106 + (j >> 1)] >> 4) : (w[106 + (j >> 1)] & 0xf);
            int sumi = 0;
            for (int l = 0; l < 4; ++l) {
                const uint8_t * g1 = c_grid_iq3s[qs[2 * l] | ((qh << (8 - 2 * l)) & 256)]; const uint8_t * g2 = c_grid_iq3s[qs[2 * l + 1] | ((qh << (7 - 2 * l)) & 256)];
                for (int t = 0; t < 4; ++t) {
                    sumi += (int) g1[t] * y[8 * l + t]     * (((sg[l] >> t) & 1) ? -1 : 1);
                    sumi += (int) g2[t] * y[8 * l + t + 4] * (((sg[l] >> (t + 4)) & 1) ? -1 : 1);
                }
            }
            return (ld_h(w) * yd) * (float) (sumi * (2 * nib + 1));
        } else if constexpr (TYPE == MI355Q_TYPE_IQ1_S) {            // :12099-12128 block: d(2) qs[32] qh u16[8]
            const uint8_t * w = wrow + b * 50; const uint8_t * qs = w + 2 + 4 * j;
            const uint32_t qh = (uint32_t) w[34 + 2 * j] | ((uint32_t) w[35 + 2 * j] << 8);
            const int ls = 2 * (int) ((qh >> 12) & 7) + 1, delta = (qh & 0x8000) ? -1 : 1;
            int lsum = 0;
            for (int l = 0; l < 4; ++l) {
                const int8_t * g = c_grid_iq1s[qs[l] | (((qh >> (3 * l)) & 7) << 8)];
                for (int t = 0; t < 8; ++t) lsum += y[8 * l + t] * (int) g[t];
            }
            return ld_h(w) * yd * ((float) (ls * lsum) + 0.125f * (float) (ls * delta * (bs0 + bs1)));
        } else if constexpr (TYPE == MI355Q_TYPE_IQ1_M) {            // :12401-12446 block: qs[32] qh[16] scales[8]
            const uint8_t * w = wrow + b * 56; const uint8_t * qs = w + 4 * j; const uint8_t * qh = w + 32 + 2 * j;
            uint32_t sc[4];
            for (int t = 0; t < 4; ++t) sc[t] = (uint32_t) w[48 + 2 * t] | ((uint32_t) w[49 + 2 * t] << 8);
            const uint32_t du = (sc[0] >> 12) | ((sc[1] >> 8) & 0x00f0u) | ((sc[2] >> 4) & 0x0f00u) | (sc[3] & 0xf000u);
            const int delta[4] = { (qh[0] & 0x08) ? -1 : 1, (qh[0] & 0x80) ? -1 : 1, (qh[1] & 0x08) ? -1 : 1, (qh[1] & 0x80) ? -1 : 1 };
            int sum1[2] = { 0, 0 }, sum2[2] = { 0, 0 };
            for (int l = 0; l < 4; ++l) {
                const int8_t * g = c_grid_iq1s[qs[l] | (((uint32_t) qh[l / 2] << (8 - 4 * (l % 2))) & 0x700)];
                int l1 = 0, l2 = 0;
                for (int t = 0; t < 8; ++t) { l1 += y[8 * l + t] * (int) g[t]; l2 += y[8 * l + t]; }
                sum1[l / 2] += l1; sum2[l / 2] += l2 * delta[l];
            }
            const int ls1 = 2 * (int) ((sc[j / 2] >> (6 * (j % 2) + 0)) & 0x7) + 1, ls2 = 2 * (int) ((sc[j / 2] >> (6 * (j % 2) + 3)) & 0x7) + 1;
            return h2f(du) * yd * ((float) (sum1[0] * ls1 + sum1[1] * ls2) + 0.125f * (float) (sum2[0] * ls1 + sum2[1] * ls2));
        } else {                                                     // IQ4_XS  :12981-13012
            const uint8_t * w = wrow + b * 136;
            const uint32_t sh = (uint32_t) w[2] | ((uint32_t) w[3] << 8);
            const int ls = ((w[4 + (j >> 1)] >> (4 * (j & 1))) & 0x0F) | (int) (((sh >> (2 * j)) & 3) << 4);
            const uint8_t * qs = w + 8 + 16 * j;
            int s1 = 0, s2 = 0;
            for (int l = 0; l < 16; ++l) { s1 += y[l] * c_iq4_codebook[qs[l] & 0x0F]; s2 += y[l + 16] * c_iq4_codebook[qs[l] >> 4]; }
            return ((ld_h(w) * yd) * (float) (ls - 32)) * (float) (s1 + s2);
        }
    }
}

struct GenericMoe {        // MUL_MAT_ID mode when ids != nullptr: blockIdx.y = (token t, slot u) pair
    const int32_t * ids; int64_t ids_stride; int64_t expert_stride; int n_used; int x_ne1; int n_expert; int pad;
};

// grid.x = rows (capped, grid-stride), grid.y = activation rows; block = 256 (4 waves, one row each)
template <int TYPE>
__global__ void __launch_bounds__(256)
k_gemv_generic(const uint8_t * __restrict__ w, int64_t w_stride, const uint8_t * __restrict__ act, int64_t act_stride,
               float * __restrict__ y, int64_t y_stride, int64_t m, int64_t k, const GenericMoe moe) {
    const int     lane = lane_id();
    int64_t       n    = blockIdx.y;
    const int64_t nsub = k / 32;
    float * yrow = (float *) ((char *) y + n * y_stride);
    if (moe.ids) {
        const int pair = (int) blockIdx.y, t = pair / moe.n_used, u = pair - t * moe.n_used;
        const int e = *(const int32_t *) ((const char *) moe.ids + (int64_t) t * moe.ids_stride + 4 * u);
        if (e < 0 || e >= moe.n_expert) {                      // the reference asserts on such an id; on every device path the pair's row becomes NaN
            for (int64_t row = (int64_t) blockIdx.x * 256 + threadIdx.x; row < m; row += (int64_t) gridDim.x * 256) yrow[row] = __int_as_float(0x7FC00000);
            return;
        }
        w += (int64_t) e * moe.expert_stride;
        n = (int64_t) t * moe.x_ne1 + (u % moe.x_ne1);         // activation row in the quantized workspace
    }
    const uint8_t * arow = act + n * act_stride;
    for (int64_t row = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6); row < m; row += (int64_t) gridDim.x * 4) {
        const uint8_t * wrow = w + row * w_stride;
        float acc = 0.0f;
        for (int64_t s = lane; s < nsub; s += 64) acc += sub_dot<TYPE>(wrow, arow, s);
        acc = wave_sum(acc);
        if (lane == 0) yrow[row] = acc;
    }
}

#define MI355Q_GENERIC_CASE(T) \
    case T: hipLaunchKernelGGL((k_gemv_generic<T>), grid, block, 0, stream, (const uint8_t *) w, w_stride, \
                               (const uint8_t *) act, act_stride, y, y_stride, m, k, gm); break;

int launch_gemv_generic(int type, const void * w, int64_t w_stride, const void * act, int64_t act_stride,
                        float * y, int64_t y_stride, int64_t m, int64_t n, int64_t k, hipStream_t stream,
                        const GenericMoe * moe = nullptr) {
    if (m <= 0 || n <= 0) return MI355Q_OK;
    if (n > 65535) return MI355Q_ERR_UNSUPPORTED;
    GenericMoe gm = {};
    if (moe) gm = *moe;
    const dim3 grid((unsigned) ((m + 3) / 4 < 8192 ? (m + 3) / 4 : 8192), (unsigned) n), block(256);
    switch (type) {
        MI355Q_GENERIC_CASE(MI355Q_TYPE_Q4_0) MI355Q_GENERIC_CASE(MI355Q_TYPE_Q4_1)
        MI355Q_GENERIC_CASE(MI355Q_TYPE_Q5_0) MI355Q_GENERIC_CASE(MI355Q_TYPE_Q5_1)
        MI355Q_GENERIC_CASE(MI355Q_TYPE_Q8_0) MI355Q_GENERIC_CASE(MI355Q_TYPE_Q2_K)
        MI355Q_GENERIC_CASE(MI355Q_TYPE_Q3_K) MI355Q_GENERIC_CASE(MI355Q_TYPE_Q4_K)
        MI355Q_GENERIC_CASE(MI355Q_TYPE_Q5_K) MI355Q_GENERIC_CASE(MI355Q_TYPE_Q6_K)
        MI355Q_GENERIC_CASE(MI355Q_TYPE_IQ4_NL) MI355Q_GENERIC_CASE(MI355Q_TYPE_IQ4_XS)
        MI355Q_GENERIC_CASE(MI355Q_TYPE_IQ2_XXS) MI355Q_GENERIC_CASE(MI355Q_TYPE_IQ2_XS) MI355Q_GENERIC_CASE(MI355Q_TYPE_IQ2_S)
        MI355Q_GENERIC_CASE(MI355Q_TYPE_IQ3_XXS) MI355Q_GENERIC_CASE(MI355Q_TYPE_IQ3_S)
        MI355Q_GENERIC_CASE(MI355Q_TYPE_IQ1_S) MI355Q_GENERIC_CASE(MI355Q_TYPE_IQ1_M)
    default: return MI355Q_ERR_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? MI355Q_OK : MI355Q_ERR_HIP;
}

} // namespace mi355q
