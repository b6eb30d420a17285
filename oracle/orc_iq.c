/* oracle/orc_iq.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The code-book ("IQ") formats: block decode (dequantize_row_iq*, ggml/src/ggml-quants.c:2197-2432) and the
 * scalar spec of ggml_vec_dot_iq*_q8_K (ggml/src/ggml-cpu/ggml-cpu-quants.c, scalar tails cited per function),
 * with the reference's f32 operand order.  The code books in orc_iq_tables.h were reconstructed by probing the
 * reference decoder (tools/gen_iq_tables.py); block layouts are ggml-common.h:339-396.
 */
#include "oracle.h"
#include "orc_formats.h"
#include "orc_iq_tables.h"

#include <string.h>

#pragma pack(push, 1)
typedef struct { uint16_t d; uint16_t qs[32]; }                                          orc_iq2_xxs; /* 66 B  :339-343 */
typedef struct { uint16_t d; uint16_t qs[32]; uint8_t scales[8]; }                       orc_iq2_xs;  /* 74 B  :346-351 */
typedef struct { uint16_t d; uint8_t qs[64]; uint8_t qh[8]; uint8_t scales[8]; }          orc_iq2_s;   /* 82 B  :354-360 */
typedef struct { uint16_t d; uint8_t qs[96]; }                                           orc_iq3_xxs; /* 98 B  :365-369 */
typedef struct { uint16_t d; uint8_t qs[64]; uint8_t qh[8]; uint8_t signs[32]; uint8_t scales[4]; } orc_iq3_s; /* 110 B :373-380 */
typedef struct { uint16_t d; uint8_t qs[32]; uint16_t qh[8]; }                           orc_iq1_s;   /* 50 B  :383-388 */
typedef struct { uint8_t qs[32]; uint8_t qh[16]; uint8_t scales[8]; }                    orc_iq1_m;   /* 56 B  :391-396 */
#pragma pack(pop)
_Static_assert(sizeof(orc_iq2_xxs) == 66 && sizeof(orc_iq2_xs) == 74 && sizeof(orc_iq2_s) == 82 && sizeof(orc_iq3_xxs) == 98 &&
               sizeof(orc_iq3_s) == 110 && sizeof(orc_iq1_s) == 50 && sizeof(orc_iq1_m) == 56, "iq blocks");

#define H2F(h) orc_f16_to_f32(h)
#define IQ1_DELTA 0.125f                                         /* IQ1S_DELTA == IQ1M_DELTA, ggml-common.h:1078-1079 */

static inline int sgn(uint8_t signs, int j) { return (signs >> j) & 1 ? -1 : 1; }

/* the f16 super-scale of IQ1_M is scattered over the top nibbles of the four 16-bit scale words (ggml-quants.c:2395-2397) */
static float iq1m_d(const orc_iq1_m *x) {
    uint16_t sc[4]; memcpy(sc, x->scales, 8);
    const uint16_t u = (uint16_t)((sc[0] >> 12) | ((sc[1] >> 8) & 0x00f0) | ((sc[2] >> 4) & 0x0f00) | (sc[3] & 0xf000));
    return H2F(u);
}

int orc_iq_dequantize_row(int type, const void *src, float *y, int64_t nb) {
    switch (type) {
    case ORC_IQ2_XXS: { const orc_iq2_xxs *x = src;                                          /* :2197-2222 */
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(x[i].d);
            for (int ib = 0; ib < 8; ++ib) { uint32_t aux[2]; memcpy(aux, x[i].qs + 4 * ib, 8); const uint8_t *a8 = (const uint8_t *) aux;
                const float db = d * (0.5f + (float)(aux[1] >> 28)) * 0.25f;
                for (int l = 0; l < 4; ++l) { const uint8_t *g = orc_grid_iq2xxs[a8[l]]; const uint8_t s = orc_signs_iq2[(aux[1] >> (7 * l)) & 127];
                    for (int j = 0; j < 8; ++j) *y++ = db * (float) g[j] * (float) sgn(s, j); } } }
        return 0; }
    case ORC_IQ2_XS: { const orc_iq2_xs *x = src;                                            /* :2225-2248 */
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(x[i].d);
            for (int ib = 0; ib < 8; ++ib) { float db[2];
                db[0] = d * (0.5f + (float)(x[i].scales[ib] & 0xf)) * 0.25f; db[1] = d * (0.5f + (float)(x[i].scales[ib] >> 4)) * 0.25f;
                for (int l = 0; l < 4; ++l) { const uint16_t q = x[i].qs[4 * ib + l]; const uint8_t *g = orc_grid_iq2xs[q & 511]; const uint8_t s = orc_signs_iq2[q >> 9];
                    for (int j = 0; j < 8; ++j) *y++ = db[l / 2] * (float) g[j] * (float) sgn(s, j); } } }
        return 0; }
    case ORC_IQ2_S: { const orc_iq2_s *x = src;                                              /* :2252-2280 */
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(x[i].d); const uint8_t *qs = x[i].qs, *signs = x[i].qs + 32;
            for (int ib = 0; ib < 8; ++ib) { float db[2];
                db[0] = d * (0.5f + (float)(x[i].scales[ib] & 0xf)) * 0.25f; db[1] = d * (0.5f + (float)(x[i].scales[ib] >> 4)) * 0.25f;
                for (int l = 0; l < 4; ++l) { const uint8_t *g = orc_grid_iq2s[qs[l] | ((x[i].qh[ib] << (8 - 2 * l)) & 0x300)];
                    for (int j = 0; j < 8; ++j) *y++ = db[l / 2] * (float) g[j] * (float) sgn(signs[l], j); }
                qs += 4; signs += 4; } }
        return 0; }
    case ORC_IQ3_XXS: { const orc_iq3_xxs *x = src;                                          /* :2284-2312 */
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(x[i].d); const uint8_t *qs = x[i].qs, *sas = x[i].qs + 64;
            for (int ib = 0; ib < 8; ++ib) { uint32_t aux; memcpy(&aux, sas + 4 * ib, 4);
                const float db = d * (0.5f + (float)(aux >> 28)) * 0.5f;
                for (int l = 0; l < 4; ++l) { const uint8_t s = orc_signs_iq2[(aux >> (7 * l)) & 127];
                    const uint8_t *g1 = orc_grid_iq3xxs[qs[2 * l]], *g2 = orc_grid_iq3xxs[qs[2 * l + 1]];
                    for (int j = 0; j < 4; ++j) { y[j] = db * (float) g1[j] * (float) sgn(s, j); y[j + 4] = db * (float) g2[j] * (float) sgn(s, j + 4); }
                    y += 8; }
                qs += 8; } }
        return 0; }
    case ORC_IQ3_S: { const orc_iq3_s *x = src;                                              /* :2316-2357 */
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(x[i].d); const uint8_t *qs = x[i].qs, *signs = x[i].signs;
            for (int ib = 0; ib < 8; ++ib) {
                const int nib = ib & 1 ? x[i].scales[ib / 2] >> 4 : x[i].scales[ib / 2] & 0xf;
                const float db = d * (float)(1 + 2 * nib); const uint8_t qh = x[i].qh[ib];
                for (int l = 0; l < 4; ++l) {
                    const uint8_t *g1 = orc_grid_iq3s[qs[2 * l] | ((qh << (8 - 2 * l)) & 256)], *g2 = orc_grid_iq3s[qs[2 * l + 1] | ((qh << (7 - 2 * l)) & 256)];
                    for (int j = 0; j < 4; ++j) { y[j] = db * (float) g1[j] * (float) sgn(signs[l], j); y[j + 4] = db * (float) g2[j] * (float) sgn(signs[l], j + 4); }
                    y += 8; }
                qs += 8; signs += 4; } }
        return 0; }
    case ORC_IQ1_S: { const orc_iq1_s *x = src;                                              /* :2359-2382 */
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(x[i].d); const uint8_t *qs = x[i].qs;
            for (int ib = 0; ib < 8; ++ib) { const uint16_t qh = x[i].qh[ib];
                const float dl = d * (float)(2 * ((qh >> 12) & 7) + 1); const float delta = qh & 0x8000 ? -IQ1_DELTA : IQ1_DELTA;
                for (int l = 0; l < 4; ++l) { const int8_t *g = orc_grid_iq1s[qs[l] | (((qh >> (3 * l)) & 7) << 8)];
                    for (int j = 0; j < 8; ++j) *y++ = dl * ((float) g[j] + delta); }
                qs += 4; } }
        return 0; }
    case ORC_IQ1_M: { const orc_iq1_m *x = src;                                              /* :2384-2432 */
        for (int64_t i = 0; i < nb; ++i) { const float d = iq1m_d(&x[i]); uint16_t sc[4]; memcpy(sc, x[i].scales, 8);
            const uint8_t *qs = x[i].qs, *qh = x[i].qh;
            for (int ib = 0; ib < 8; ++ib) {
                const float dl1 = d * (float)(2 * ((sc[ib / 2] >> (6 * (ib % 2) + 0)) & 0x7) + 1);
                const float dl2 = d * (float)(2 * ((sc[ib / 2] >> (6 * (ib % 2) + 3)) & 0x7) + 1);
                const int idx[4] = { qs[0] | ((qh[0] << 8) & 0x700), qs[1] | ((qh[0] << 4) & 0x700), qs[2] | ((qh[1] << 8) & 0x700), qs[3] | ((qh[1] << 4) & 0x700) };
                const float delta[4] = { qh[0] & 0x08 ? -IQ1_DELTA : IQ1_DELTA, qh[0] & 0x80 ? -IQ1_DELTA : IQ1_DELTA,
                                         qh[1] & 0x08 ? -IQ1_DELTA : IQ1_DELTA, qh[1] & 0x80 ? -IQ1_DELTA : IQ1_DELTA };
                for (int l = 0; l < 4; ++l) { const int8_t *g = orc_grid_iq1s[idx[l]]; const float dl = l < 2 ? dl1 : dl2;
                    for (int j = 0; j < 8; ++j) *y++ = dl * ((float) g[j] + delta[l]); }
                qs += 4; qh += 2; } }
        return 0; }
    default: return 2;
    }
}

int orc_iq_vec_dot(int type, int64_t nb, float *out, const void *vw, const void *va) {
    const orc_q8_K *y = va;
    float sumf = 0.0f;
    switch (type) {
    case ORC_IQ2_XXS: { const orc_iq2_xxs *x = vw;                                          /* ggml-cpu-quants.c:9834-9862 */
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(x[i].d) * y[i].d; const int8_t *q8 = y[i].qs; int32_t bsum = 0;
            for (int ib = 0; ib < 8; ++ib) { uint32_t aux[2]; memcpy(aux, x[i].qs + 4 * ib, 8); const uint8_t *a8 = (const uint8_t *) aux;
                const int ls = 2 * (int)(aux[1] >> 28) + 1; int32_t sumi = 0;
                for (int l = 0; l < 4; ++l) { const uint8_t *g = orc_grid_iq2xxs[a8[l]]; const uint8_t s = orc_signs_iq2[(aux[1] >> (7 * l)) & 127];
                    for (int j = 0; j < 8; ++j) sumi += g[j] * q8[j] * sgn(s, j);
                    q8 += 8; }
                bsum += sumi * ls; }
            sumf += d * (float) bsum; }
        *out = 0.125f * sumf; return 0; }
    case ORC_IQ2_XS: { const orc_iq2_xs *x = vw;                                            /* :10411-10447 */
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(x[i].d) * y[i].d; const int8_t *q8 = y[i].qs; int32_t bsum = 0;
            for (int ib = 0; ib < 8; ++ib) { const int ls1 = 2 * (x[i].scales[ib] & 0xf) + 1, ls2 = 2 * (x[i].scales[ib] >> 4) + 1;
                for (int l = 0; l < 4; ++l) { const uint16_t q = x[i].qs[4 * ib + l]; const uint8_t *g = orc_grid_iq2xs[q & 511]; const uint8_t s = orc_signs_iq2[q >> 9];
                    int32_t sumi = 0;
                    for (int j = 0; j < 8; ++j) sumi += g[j] * q8[j] * sgn(s, j);
                    q8 += 8; bsum += sumi * (l < 2 ? ls1 : ls2); } }
            sumf += d * (float) bsum; }
        *out = 0.125f * sumf; return 0; }
    case ORC_IQ2_S: { const orc_iq2_s *x = vw;                                              /* :10885-10921 */
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(x[i].d) * y[i].d; const int8_t *q8 = y[i].qs; const uint8_t *qs = x[i].qs, *signs = x[i].qs + 32; int bsum = 0;
            for (int ib = 0; ib < 8; ++ib) { const int ls1 = 1 + 2 * (x[i].scales[ib] & 0xf), ls2 = 1 + 2 * (x[i].scales[ib] >> 4);
                for (int l = 0; l < 4; ++l) { const uint8_t *g = orc_grid_iq2s[qs[l] | ((x[i].qh[ib] << (8 - 2 * l)) & 0x300)]; int sumi = 0;
                    for (int j = 0; j < 8; ++j) sumi += q8[j] * g[j] * sgn(signs[l], j);
                    q8 += 8; bsum += (l < 2 ? ls1 : ls2) * sumi; }
                qs += 4; signs += 4; }
            sumf += d * (float) bsum; }
        *out = 0.125f * sumf; return 0; }
    case ORC_IQ3_XXS: { const orc_iq3_xxs *x = vw;                                          /* :11218-11247 */
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(x[i].d) * y[i].d; const int8_t *q8 = y[i].qs; const uint8_t *q3 = x[i].qs, *gas = x[i].qs + 64; int32_t bsum = 0;
            for (int ib = 0; ib < 8; ++ib) { uint32_t aux; memcpy(&aux, gas + 4 * ib, 4); const int ls = 2 * (int)(aux >> 28) + 1; int32_t sumi = 0;
                for (int l = 0; l < 4; ++l) { const uint8_t *g1 = orc_grid_iq3xxs[q3[2 * l]], *g2 = orc_grid_iq3xxs[q3[2 * l + 1]]; const uint8_t s = orc_signs_iq2[(aux >> (7 * l)) & 127];
                    for (int j = 0; j < 4; ++j) { sumi += g1[j] * q8[j] * sgn(s, j); sumi += g2[j] * q8[j + 4] * sgn(s, j + 4); }
                    q8 += 8; }
                q3 += 8; bsum += sumi * ls; }
            sumf += d * (float) bsum; }
        *out = 0.25f * sumf; return 0; }
    case ORC_IQ3_S: { const orc_iq3_s *x = vw;                                              /* :11733-11776 */
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(x[i].d) * y[i].d; const int8_t *q8 = y[i].qs; const uint8_t *qs = x[i].qs, *signs = x[i].signs; int32_t bsum = 0;
            for (int ib = 0; ib < 8; ++ib) {
                const int nib = ib & 1 ? x[i].scales[ib / 2] >> 4 : x[i].scales[ib / 2] & 0xf; const int ls = 2 * nib + 1; const uint8_t qh = x[i].qh[ib]; int32_t sumi = 0;
                for (int l = 0; l < 4; ++l) {
                    const uint8_t *g1 = orc_grid_iq3s[qs[2 * l] | ((qh << (8 - 2 * l)) & 256)], *g2 = orc_grid_iq3s[qs[2 * l + 1] | ((qh << (7 - 2 * l)) & 256)];
                    for (int j = 0; j < 4; ++j) { sumi += g1[j] * q8[j] * sgn(signs[l], j); sumi += g2[j] * q8[j + 4] * sgn(signs[l], j + 4); }
                    q8 += 8; }
                qs += 8; signs += 4; bsum += sumi * ls; }
            sumf += d * (float) bsum; }
        *out = sumf; return 0; }
    case ORC_IQ1_S: { const orc_iq1_s *x = vw;                                              /* :12099-12128 */
        for (int64_t i = 0; i < nb; ++i) { const int8_t *q8 = y[i].qs; const uint8_t *qs = x[i].qs; int sumi = 0, sumi1 = 0;
            for (int ib = 0; ib < 8; ++ib) { const uint16_t qh = x[i].qh[ib]; const int ls = 2 * ((qh >> 12) & 7) + 1; const int delta = qh & 0x8000 ? -1 : 1; int lsum = 0;
                for (int l = 0; l < 4; ++l) { const int8_t *g = orc_grid_iq1s[qs[l] | (((qh >> (3 * l)) & 7) << 8)];
                    for (int j = 0; j < 8; ++j) lsum += q8[j] * g[j];
                    q8 += 8; }
                sumi += ls * lsum; sumi1 += ls * delta * (y[i].bsums[2 * ib] + y[i].bsums[2 * ib + 1]); qs += 4; }
            sumf += H2F(x[i].d) * y[i].d * ((float) sumi + IQ1_DELTA * (float) sumi1); }
        *out = sumf; return 0; }
    case ORC_IQ1_M: { const orc_iq1_m *x = vw;                                              /* :12401-12446 */
        for (int64_t i = 0; i < nb; ++i) { const int8_t *q8 = y[i].qs; const uint8_t *qs = x[i].qs, *qh = x[i].qh; uint16_t sc[4]; memcpy(sc, x[i].scales, 8);
            int sumi1 = 0, sumi2 = 0;
            for (int ib = 0; ib < 8; ++ib) {
                const int delta[4] = { qh[0] & 0x08 ? -1 : 1, qh[0] & 0x80 ? -1 : 1, qh[1] & 0x08 ? -1 : 1, qh[1] & 0x80 ? -1 : 1 };
                int sum1[2] = {0, 0}, sum2[2] = {0, 0};
                for (int l = 0; l < 4; ++l) { const int8_t *g = orc_grid_iq1s[qs[l] | (((uint16_t) qh[l / 2] << (8 - 4 * (l % 2))) & 0x700)]; int l1 = 0, l2 = 0;
                    for (int j = 0; j < 8; ++j) { l1 += q8[j] * g[j]; l2 += q8[j]; }
                    q8 += 8; sum1[l / 2] += l1; sum2[l / 2] += l2 * delta[l]; }
                const int ls1 = 2 * ((sc[ib / 2] >> (6 * (ib % 2) + 0)) & 0x7) + 1, ls2 = 2 * ((sc[ib / 2] >> (6 * (ib % 2) + 3)) & 0x7) + 1;
                sumi1 += sum1[0] * ls1 + sum1[1] * ls2; sumi2 += sum2[0] * ls1 + sum2[1] * ls2; qs += 4; qh += 2; }
            sumf += iq1m_d(&x[i]) * y[i].d * ((float) sumi1 + IQ1_DELTA * (float) sumi2); }
        *out = sumf; return 0; }
    default: return 2;
    }
}
