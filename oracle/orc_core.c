/* oracle/orc_core.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Plain-C restatement of the reference CPU path for quantized MUL_MAT / MUL_MAT_ID.
 * Integer arithmetic is exact by construction; every f32 expression below keeps the
 * operand order of the reference's scalar code so the results are bit-identical to
 * oracle/_ref/scalar (compile with -ffp-contract=off, no fast-math).
 */
#include "oracle.h"
#include "orc_formats.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* geometry                                                                               */
/* ------------------------------------------------------------------------------------ */
typedef struct { int type; int blck; int bytes; int act; } orc_geom;

static const orc_geom k_geom[] = {
    /* type        blck  bytes  activation format used by the CPU (ggml-cpu.c:211-376) */
    { ORC_Q4_0,    32,   18,    ORC_Q8_0 },
    { ORC_Q4_1,    32,   20,    ORC_Q8_1 },
    { ORC_Q5_0,    32,   22,    ORC_Q8_0 },
    { ORC_Q5_1,    32,   24,    ORC_Q8_1 },
    { ORC_Q8_0,    32,   34,    ORC_Q8_0 },
    { ORC_Q2_K,    256,  84,    ORC_Q8_K },
    { ORC_Q3_K,    256,  110,   ORC_Q8_K },
    { ORC_Q4_K,    256,  144,   ORC_Q8_K },
    { ORC_Q5_K,    256,  176,   ORC_Q8_K },
    { ORC_Q6_K,    256,  210,   ORC_Q8_K },
    { ORC_IQ4_NL,  32,   18,    ORC_Q8_0 },
    { ORC_IQ4_XS,  256,  136,   ORC_Q8_K },
    { ORC_IQ2_XXS, 256,  66,    ORC_Q8_K },
    { ORC_IQ2_XS,  256,  74,    ORC_Q8_K },
    { ORC_IQ2_S,   256,  82,    ORC_Q8_K },
    { ORC_IQ3_XXS, 256,  98,    ORC_Q8_K },
    { ORC_IQ3_S,   256,  110,   ORC_Q8_K },
    { ORC_IQ1_S,   256,  50,    ORC_Q8_K },
    { ORC_IQ1_M,   256,  56,    ORC_Q8_K },
    /* activation-only formats */
    { ORC_Q8_1,    32,   36,    -1 },
    { ORC_Q8_K,    256,  292,   -1 },
};

static const orc_geom *geom(int type) {
    for (size_t i = 0; i < sizeof(k_geom) / sizeof(k_geom[0]); ++i)
        if (k_geom[i].type == type) return &k_geom[i];
    return NULL;
}

int orc_supported(int type) { const orc_geom *g = geom(type); return g && g->act >= 0; }
int64_t orc_blck_size(int type) { const orc_geom *g = geom(type); return g ? g->blck : 0; }
int64_t orc_type_size(int type) { const orc_geom *g = geom(type); return g ? g->bytes : 0; }
int64_t orc_row_size(int type, int64_t k) { const orc_geom *g = geom(type); return g ? k / g->blck * g->bytes : 0; }
int orc_vec_dot_type(int type) { const orc_geom *g = geom(type); return g ? g->act : -1; }

/* ------------------------------------------------------------------------------------ */
/* f16 <-> f32, IEEE binary16, round-to-nearest-even, subnormals kept                     */
/* ------------------------------------------------------------------------------------ */
float orc_f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1Fu;
    const uint32_t m = h & 0x3FFu;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) {
            bits = sign;
        } else {                                   /* subnormal: value = m * 2^-24 */
            float v = (float) m * (1.0f / 16777216.0f);
            memcpy(&bits, &v, 4);
            bits |= sign;
        }
    } else if (e == 31) {
        bits = sign | 0x7F800000u | (m << 13);
    } else {
        bits = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f; memcpy(&f, &bits, 4);
    return f;
}

uint16_t orc_f32_to_f16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    const uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) {                       /* inf / nan */
        return (uint16_t)(sign | 0x7C00u | (ax > 0x7F800000u ? 0x0200u : 0));
    }
    if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);   /* rounds to >= 65520 -> inf */
    if (ax < 0x33000001u) return sign;                           /* <= 2^-25 -> 0 (ties to even) */
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t man = (ax & 0x7FFFFFu) | 0x800000u;                 /* 24-bit significand */
    int shift;
    uint32_t hexp;
    if (e < -14) { shift = 13 + (-14 - e); hexp = 0; }           /* subnormal half */
    else         { shift = 13;             hexp = (uint32_t)(e + 15); }
    uint32_t q = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1u);
    const uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    uint32_t out;
    if (hexp == 0) out = q;                                      /* may carry into exp=1: fine */
    else           out = ((hexp - 1) << 10) + q;                 /* q has the implicit bit at 0x400 */
    return (uint16_t)(sign | out);
}

#define H2F(h) orc_f16_to_f32(h)

/* code-book formats live in orc_iq.c */
int orc_iq_dequantize_row(int type, const void *src, float *y, int64_t nb);
int orc_iq_vec_dot(int type, int64_t nb, float *out, const void *vw, const void *va);

/* ------------------------------------------------------------------------------------ */
/* K-quant 6-bit (scale,min) pair j of the 12-byte field  (get_scale_min_k4,              */
/* ggml-quants.c:631-638)                                                                 */
/* ------------------------------------------------------------------------------------ */
static void k4_scale_min(int j, const uint8_t *p, int *sc, int *mn) {
    if (j < 4) { *sc = p[j] & 63;                            *mn = p[j + 4] & 63; }
    else       { *sc = (p[j + 4] & 0x0F) | ((p[j - 4] >> 6) << 4);
                 *mn = (p[j + 4] >> 4)   | ((p[j]     >> 6) << 4); }
}

/* Q3_K: 16 signed 6-bit scales out of 12 bytes (ggml-quants.c:1073-1079) */
static void q3k_scales(const uint8_t *p, int out[16]) {
    for (int j = 0; j < 16; ++j) {
        const int lo = (j < 8) ? (p[j] & 0x0F) : (p[j - 8] >> 4);
        const int hi = (p[8 + (j & 3)] >> (2 * (j >> 2))) & 3;
        out[j] = (lo | (hi << 4)) - 32;
    }
}

/* ------------------------------------------------------------------------------------ */
/* integer decode of one block into q[] (the "a"/aux8 arrays of the scalar vec_dots)      */
/* ------------------------------------------------------------------------------------ */
static void q4k_unpack(const uint8_t *qs, int8_t *q) {           /* ggml-cpu-quants.c:7548-7554 */
    for (int g = 0; g < 4; ++g)
        for (int l = 0; l < 32; ++l) {
            q[64 * g + l]      = (int8_t)(qs[32 * g + l] & 0x0F);
            q[64 * g + 32 + l] = (int8_t)(qs[32 * g + l] >> 4);
        }
}
static void q5k_unpack(const uint8_t *qs, const uint8_t *qh, int8_t *q) { /* :8367-8381 */
    for (int g = 0; g < 4; ++g)
        for (int l = 0; l < 32; ++l) {
            q[64 * g + l]      = (int8_t)((qs[32 * g + l] & 0x0F) + ((qh[l] >> (2 * g)     & 1) ? 16 : 0));
            q[64 * g + 32 + l] = (int8_t)((qs[32 * g + l] >> 4)   + ((qh[l] >> (2 * g + 1) & 1) ? 16 : 0));
        }
}
static void q6k_unpack(const uint8_t *ql, const uint8_t *qh, int8_t *q) { /* :9438-9449 */
    for (int h = 0; h < 2; ++h)
        for (int l = 0; l < 32; ++l) {
            const uint8_t hb = qh[32 * h + l];
            q[128 * h + l]      = (int8_t)(((ql[64 * h + l]      & 0x0F) | (((hb >> 0) & 3) << 4)) - 32);
            q[128 * h + 32 + l] = (int8_t)(((ql[64 * h + 32 + l] & 0x0F) | (((hb >> 2) & 3) << 4)) - 32);
            q[128 * h + 64 + l] = (int8_t)(((ql[64 * h + l]      >> 4)   | (((hb >> 4) & 3) << 4)) - 32);
            q[128 * h + 96 + l] = (int8_t)(((ql[64 * h + 32 + l] >> 4)   | (((hb >> 6) & 3) << 4)) - 32);
        }
}
static void q3k_unpack(const uint8_t *qs, const uint8_t *hm, int8_t *q) { /* :6620-6639 */
    for (int h = 0; h < 2; ++h)
        for (int s = 0; s < 4; ++s)
            for (int l = 0; l < 32; ++l) {
                const int lo = (qs[32 * h + l] >> (2 * s)) & 3;
                const int bit = (hm[l] >> (4 * h + s)) & 1;
                q[128 * h + 32 * s + l] = (int8_t)(lo - (bit ? 0 : 4));
            }
}
static void q2k_unpack(const uint8_t *qs, int8_t *q) {           /* :5504-5519 */
    for (int h = 0; h < 2; ++h)
        for (int s = 0; s < 4; ++s)
            for (int l = 0; l < 32; ++l)
                q[128 * h + 32 * s + l] = (int8_t)((qs[32 * h + l] >> (2 * s)) & 3);
}
static int q5_small(const uint8_t *qs, uint32_t qh, int j, int hi) {  /* 5-bit value, unsigned */
    if (!hi) return (qs[j] & 0x0F) | (int)(((qh >> j) << 4) & 0x10);
    return (qs[j] >> 4) | (int)((qh >> (j + 12)) & 0x10);
}

/* ------------------------------------------------------------------------------------ */
/* dequantize_row_<type>  (ggml-quants.c:255-363, 712-745, 1056-1100, 1280-1302,          */
/*                         1482-1507, 1690-1719, 2436-2475)                               */
/* ------------------------------------------------------------------------------------ */
int orc_dequantize_row(int type, const void *src, float *y, int64_t k) {
    const orc_geom *g = geom(type);
    if (!g || k % g->blck) return 1;
    const int64_t nb = k / g->blck;
    switch (type) {
    case ORC_Q4_0: { const orc_q4_0 *x = src;
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(x[i].d);
            for (int j = 0; j < 16; ++j) {
                y[32 * i + j]      = (float)((x[i].qs[j] & 0x0F) - 8) * d;
                y[32 * i + j + 16] = (float)((x[i].qs[j] >> 4) - 8) * d; } }
        return 0; }
    case ORC_Q4_1: { const orc_q4_1 *x = src;
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(x[i].d), m = H2F(x[i].m);
            for (int j = 0; j < 16; ++j) {
                y[32 * i + j]      = (float)(x[i].qs[j] & 0x0F) * d + m;
                y[32 * i + j + 16] = (float)(x[i].qs[j] >> 4) * d + m; } }
        return 0; }
    case ORC_Q5_0: { const orc_q5_0 *x = src;
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(x[i].d); uint32_t qh; memcpy(&qh, x[i].qh, 4);
            for (int j = 0; j < 16; ++j) {
                y[32 * i + j]      = (float)(q5_small(x[i].qs, qh, j, 0) - 16) * d;
                y[32 * i + j + 16] = (float)(q5_small(x[i].qs, qh, j, 1) - 16) * d; } }
        return 0; }
    case ORC_Q5_1: { const orc_q5_1 *x = src;
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(x[i].d), m = H2F(x[i].m); uint32_t qh; memcpy(&qh, x[i].qh, 4);
            for (int j = 0; j < 16; ++j) {
                y[32 * i + j]      = (float) q5_small(x[i].qs, qh, j, 0) * d + m;
                y[32 * i + j + 16] = (float) q5_small(x[i].qs, qh, j, 1) * d + m; } }
        return 0; }
    case ORC_Q8_0: { const orc_q8_0 *x = src;
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(x[i].d);
            for (int j = 0; j < 32; ++j) y[32 * i + j] = (float) x[i].qs[j] * d; }
        return 0; }
    case ORC_IQ4_NL: { const orc_iq4_nl *x = src;
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(x[i].d);
            for (int j = 0; j < 16; ++j) {
                y[32 * i + j]      = d * (float) orc_iq4_codebook[x[i].qs[j] & 0x0F];
                y[32 * i + j + 16] = d * (float) orc_iq4_codebook[x[i].qs[j] >> 4]; } }
        return 0; }
    case ORC_Q2_K: { const orc_q2_K *x = src; int8_t q[256];
        for (int64_t i = 0; i < nb; ++i) {
            const float d = H2F(x[i].d), mn = H2F(x[i].dmin);
            q2k_unpack(x[i].qs, q);
            /* element e lives in 16-group: within half h, shift s, l<16 -> scale 8h+2s, l>=16 -> 8h+2s+1 */
            for (int e = 0; e < 256; ++e) {
                const int sc = x[i].scales[e / 16];
                const float dl = d * (float)(sc & 0x0F), ml = mn * (float)(sc >> 4);
                y[256 * i + e] = dl * (float) q[e] - ml;
            } }
        return 0; }
    case ORC_Q3_K: { const orc_q3_K *x = src; int8_t q[256]; int sc[16];
        for (int64_t i = 0; i < nb; ++i) {
            const float d = H2F(x[i].d);
            q3k_unpack(x[i].qs, x[i].hmask, q); q3k_scales(x[i].scales, sc);
            for (int e = 0; e < 256; ++e) { const float dl = d * (float) sc[e / 16]; y[256 * i + e] = dl * (float) q[e]; } }
        return 0; }
    case ORC_Q4_K: { const orc_q4_K *x = src; int8_t q[256];
        for (int64_t i = 0; i < nb; ++i) {
            const float d = H2F(x[i].d), mn = H2F(x[i].dmin);
            q4k_unpack(x[i].qs, q);
            for (int j = 0; j < 8; ++j) { int sc, m; k4_scale_min(j, x[i].scales, &sc, &m);
                const float d1 = d * (float) sc, m1 = mn * (float) m;
                for (int l = 0; l < 32; ++l) y[256 * i + 32 * j + l] = d1 * (float) q[32 * j + l] - m1; } }
        return 0; }
    case ORC_Q5_K: { const orc_q5_K *x = src; int8_t q[256];
        for (int64_t i = 0; i < nb; ++i) {
            const float d = H2F(x[i].d), mn = H2F(x[i].dmin);
            q5k_unpack(x[i].qs, x[i].qh, q);
            for (int j = 0; j < 8; ++j) { int sc, m; k4_scale_min(j, x[i].scales, &sc, &m);
                const float d1 = d * (float) sc, m1 = mn * (float) m;
                for (int l = 0; l < 32; ++l) y[256 * i + 32 * j + l] = d1 * (float) q[32 * j + l] - m1; } }
        return 0; }
    case ORC_Q6_K: { const orc_q6_K *x = src; int8_t q[256];
        for (int64_t i = 0; i < nb; ++i) {
            const float d = H2F(x[i].d);
            q6k_unpack(x[i].ql, x[i].qh, q);
            for (int e = 0; e < 256; ++e) y[256 * i + e] = d * (float) x[i].scales[e / 16] * (float) q[e]; }
        return 0; }
    case ORC_IQ4_XS: { const orc_iq4_xs *x = src;
        for (int64_t i = 0; i < nb; ++i) {
            const float d = H2F(x[i].d);
            for (int ib = 0; ib < 8; ++ib) {
                const int ls = ((x[i].scales_l[ib / 2] >> (4 * (ib % 2))) & 0x0F) | (((x[i].scales_h >> (2 * ib)) & 3) << 4);
                const float dl = d * (float)(ls - 32);
                for (int j = 0; j < 16; ++j) {
                    y[256 * i + 32 * ib + j]      = dl * (float) orc_iq4_codebook[x[i].qs[16 * ib + j] & 0x0F];
                    y[256 * i + 32 * ib + j + 16] = dl * (float) orc_iq4_codebook[x[i].qs[16 * ib + j] >> 4]; } } }
        return 0; }
    default: return orc_iq_dequantize_row(type, src, y, nb);
    }
}

/* ------------------------------------------------------------------------------------ */
/* activation quantizers                                                                  */
/* ------------------------------------------------------------------------------------ */
static int8_t round_q(float v, int mode) {
    return (int8_t)(mode == ORC_ROUND_EVEN ? nearbyintf(v) : roundf(v));
}

void orc_quantize_row_q8_0(const float *x, void *vy, int64_t k, int mode) {
    orc_q8_0 *y = vy;
    for (int64_t i = 0; i < k / 32; ++i) {
        float amax = 0.0f;
        for (int j = 0; j < 32; ++j) { const float a = fabsf(x[32 * i + j]); if (a > amax) amax = a; }
        const float d = amax / 127.0f;
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d = orc_f32_to_f16(d);
        for (int j = 0; j < 32; ++j) y[i].qs[j] = round_q(x[32 * i + j] * id, mode);
    }
}

void orc_quantize_row_q8_1(const float *x, void *vy, int64_t k, int mode) {
    orc_q8_1 *y = vy;
    for (int64_t i = 0; i < k / 32; ++i) {
        float amax = 0.0f;
        for (int j = 0; j < 32; ++j) { const float a = fabsf(x[32 * i + j]); if (a > amax) amax = a; }
        const float d = amax / 127.0f;
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d = orc_f32_to_f16(d);
        int sum = 0;
        for (int j = 0; j < 32; ++j) { y[i].qs[j] = round_q(x[32 * i + j] * id, mode); sum += y[i].qs[j]; }
        y[i].s = orc_f32_to_f16((float) sum * d);
    }
}

/* nearest_int(): round-half-even via the 1.5*2^23 magic constant (ggml-quants.c:372-377) */
static int nearest_int_magic(float v) {
    float t = v + 12582912.0f; int32_t i; memcpy(&i, &t, 4);
    return (i & 0x007FFFFF) - 0x00400000;
}

void orc_quantize_row_q8_K(const float *x, void *vy, int64_t k) {
    orc_q8_K *y = vy;
    for (int64_t i = 0; i < k / 256; ++i, x += 256) {
        float amax = 0.0f, vmax = 0.0f;               /* signed value of the FIRST largest-|x| element */
        for (int j = 0; j < 256; ++j) { const float a = fabsf(x[j]); if (a > amax) { amax = a; vmax = x[j]; } }
        if (!amax) {                                   /* all-zero block: bsums left as they are */
            y[i].d = 0.0f; memset(y[i].qs, 0, 256); continue;
        }
        const float iscale = -127.0f / vmax;
        for (int j = 0; j < 256; ++j) { int v = nearest_int_magic(iscale * x[j]); y[i].qs[j] = (int8_t)(v > 127 ? 127 : v); }
        for (int j = 0; j < 16; ++j) { int s = 0; for (int l = 0; l < 16; ++l) s += y[i].qs[16 * j + l]; y[i].bsums[j] = (int16_t) s; }
        y[i].d = 1.0f / iscale;
    }
}

int orc_quantize_row_act(int act_type, const float *x, void *y, int64_t k, int mode) {
    switch (act_type) {
    case ORC_Q8_0: orc_quantize_row_q8_0(x, y, k, mode); return 0;
    case ORC_Q8_1: orc_quantize_row_q8_1(x, y, k, mode); return 0;
    case ORC_Q8_K: orc_quantize_row_q8_K(x, y, k);       return 0;
    default: return 1;
    }
}

/* ------------------------------------------------------------------------------------ */
/* vec_dot: scalar spec of ggml_vec_dot_<type>_<act> (ggml-cpu-quants.c)                  */
/* ------------------------------------------------------------------------------------ */

/* the shared K-quant accumulation skeleton of :7535-7591, :8351-8412, :9423-9465, :6604-6661:
 * 8 int32 lanes (element index mod 8) per block, scaled by the 16- or 32-wide sub-scale,
 * then 8 float lanes sums[l] += d*aux32[l], folded at the very end. */
static void lanes_accumulate(const int8_t *q, const int8_t *q8, const int *sub_scale, int sub_len, int32_t aux32[8]) {
    for (int l = 0; l < 8; ++l) aux32[l] = 0;
    for (int e = 0; e < 256; ++e) aux32[e & 7] += sub_scale[e / sub_len] * ((int) q8[e] * (int) q[e]);
}

int orc_vec_dot(int type, int64_t k, float *out, const void *vw, const void *va) {
    const orc_geom *g = geom(type);
    if (!g || g->act < 0 || k % g->blck) return 1;
    const int64_t nb = k / g->blck;
    float sumf = 0.0f;
    switch (type) {
    case ORC_Q4_0: { const orc_q4_0 *x = vw; const orc_q8_0 *y = va;            /* :2591-2605 */
        for (int64_t i = 0; i < nb; ++i) { int s0 = 0, s1 = 0;
            for (int j = 0; j < 16; ++j) { s0 += ((x[i].qs[j] & 0x0F) - 8) * y[i].qs[j]; s1 += ((x[i].qs[j] >> 4) - 8) * y[i].qs[j + 16]; }
            sumf += (float)(s0 + s1) * H2F(x[i].d) * H2F(y[i].d); }
        break; }
    case ORC_Q4_1: { const orc_q4_1 *x = vw; const orc_q8_1 *y = va;            /* :2910-2924 */
        for (int64_t i = 0; i < nb; ++i) { int s0 = 0, s1 = 0;
            for (int j = 0; j < 16; ++j) { s0 += (x[i].qs[j] & 0x0F) * y[i].qs[j]; s1 += (x[i].qs[j] >> 4) * y[i].qs[j + 16]; }
            sumf += (H2F(x[i].d) * H2F(y[i].d)) * (float)(s0 + s1) + H2F(x[i].m) * H2F(y[i].s); }
        break; }
    case ORC_Q5_0: { const orc_q5_0 *x = vw; const orc_q8_0 *y = va;            /* :3228-3248 */
        for (int64_t i = 0; i < nb; ++i) { uint32_t qh; memcpy(&qh, x[i].qh, 4); int s0 = 0, s1 = 0;
            for (int j = 0; j < 16; ++j) { s0 += (q5_small(x[i].qs, qh, j, 0) - 16) * y[i].qs[j]; s1 += (q5_small(x[i].qs, qh, j, 1) - 16) * y[i].qs[j + 16]; }
            sumf += (H2F(x[i].d) * H2F(y[i].d)) * (float)(s0 + s1); }
        break; }
    case ORC_Q5_1: { const orc_q5_1 *x = vw; const orc_q8_1 *y = va;            /* :3571-3591 */
        for (int64_t i = 0; i < nb; ++i) { uint32_t qh; memcpy(&qh, x[i].qh, 4); int s0 = 0, s1 = 0;
            for (int j = 0; j < 16; ++j) { s0 += q5_small(x[i].qs, qh, j, 0) * y[i].qs[j]; s1 += q5_small(x[i].qs, qh, j, 1) * y[i].qs[j + 16]; }
            sumf += (H2F(x[i].d) * H2F(y[i].d)) * (float)(s0 + s1) + H2F(x[i].m) * H2F(y[i].s); }
        break; }
    case ORC_Q8_0: { const orc_q8_0 *x = vw; const orc_q8_0 *y = va;            /* :4004-4012 */
        for (int64_t i = 0; i < nb; ++i) { int s = 0;
            for (int j = 0; j < 32; ++j) s += x[i].qs[j] * y[i].qs[j];
            sumf += (float) s * (H2F(x[i].d) * H2F(y[i].d)); }
        break; }
    case ORC_IQ4_NL: { const orc_iq4_nl *x = vw; const orc_q8_0 *y = va;        /* :12652-12660 */
        for (int64_t i = 0; i < nb; ++i) { const float d = H2F(y[i].d) * H2F(x[i].d); int s1 = 0, s2 = 0;
            for (int j = 0; j < 16; ++j) { s1 += y[i].qs[j] * orc_iq4_codebook[x[i].qs[j] & 0x0F]; s2 += y[i].qs[j + 16] * orc_iq4_codebook[x[i].qs[j] >> 4]; }
            sumf += d * (float)(s1 + s2); }
        break; }
    case ORC_Q2_K: { const orc_q2_K *x = vw; const orc_q8_K *y = va; int8_t q[256];  /* :5485-5523 */
        for (int64_t i = 0; i < nb; ++i) {
            int summs = 0;
            for (int j = 0; j < 16; ++j) summs += y[i].bsums[j] * (x[i].scales[j] >> 4);
            const float dall = y[i].d * H2F(x[i].d);
            const float dmin = y[i].d * H2F(x[i].dmin);
            q2k_unpack(x[i].qs, q);
            int isum = 0;
            for (int j = 0; j < 16; ++j) { int t = 0; for (int l = 0; l < 16; ++l) t += y[i].qs[16 * j + l] * q[16 * j + l]; isum += (x[i].scales[j] & 0x0F) * t; }
            sumf += dall * (float) isum - dmin * (float) summs; }
        break; }
    case ORC_Q3_K: { const orc_q3_K *x = vw; const orc_q8_K *y = va; int8_t q[256]; int sc[16]; int32_t aux32[8]; /* :6604-6661 */
        float sums[8] = {0};
        for (int64_t i = 0; i < nb; ++i) {
            q3k_unpack(x[i].qs, x[i].hmask, q); q3k_scales(x[i].scales, sc);
            lanes_accumulate(q, y[i].qs, sc, 16, aux32);
            const float d = H2F(x[i].d) * y[i].d;
            for (int l = 0; l < 8; ++l) sums[l] += d * (float) aux32[l]; }
        for (int l = 0; l < 8; ++l) sumf += sums[l];
        break; }
    case ORC_Q4_K: case ORC_Q5_K: { const orc_q8_K *y = va; int8_t q[256]; int sc[8], mn[8]; int32_t aux32[8]; /* :7535-7591, :8351-8412 */
        float sums[8] = {0};
        for (int64_t i = 0; i < nb; ++i) {
            const uint8_t *scales; uint16_t hd, hdmin;
            if (type == ORC_Q4_K) { const orc_q4_K *x = (const orc_q4_K *) vw + i; q4k_unpack(x->qs, q); scales = x->scales; hd = x->d; hdmin = x->dmin; }
            else                  { const orc_q5_K *x = (const orc_q5_K *) vw + i; q5k_unpack(x->qs, x->qh, q); scales = x->scales; hd = x->d; hdmin = x->dmin; }
            for (int j = 0; j < 8; ++j) k4_scale_min(j, scales, &sc[j], &mn[j]);
            int sumi = 0;
            for (int j = 0; j < 16; ++j) sumi += y[i].bsums[j] * mn[j / 2];
            lanes_accumulate(q, y[i].qs, sc, 32, aux32);
            const float d = H2F(hd) * y[i].d;
            for (int l = 0; l < 8; ++l) sums[l] += d * (float) aux32[l];
            const float dmin = H2F(hdmin) * y[i].d;
            sumf -= dmin * (float) sumi; }
        for (int l = 0; l < 8; ++l) sumf += sums[l];
        break; }
    case ORC_Q6_K: { const orc_q6_K *x = vw; const orc_q8_K *y = va; int8_t q[256]; int sc[16]; int32_t aux32[8]; /* :9423-9465 */
        float sums[8] = {0};
        for (int64_t i = 0; i < nb; ++i) {
            q6k_unpack(x[i].ql, x[i].qh, q);
            for (int j = 0; j < 16; ++j) sc[j] = x[i].scales[j];
            lanes_accumulate(q, y[i].qs, sc, 16, aux32);
            const float d = H2F(x[i].d) * y[i].d;
            for (int l = 0; l < 8; ++l) sums[l] += d * (float) aux32[l]; }
        for (int l = 0; l < 8; ++l) sumf += sums[l];
        break; }
    case ORC_IQ4_XS: { const orc_iq4_xs *x = vw; const orc_q8_K *y = va;        /* :12981-13012 */
        for (int64_t i = 0; i < nb; ++i) {
            const float d4d8 = H2F(x[i].d) * y[i].d;
            for (int ib = 0; ib < 8; ++ib) {
                const int ls = ((x[i].scales_l[ib / 2] >> (4 * (ib % 2))) & 0x0F) | (((x[i].scales_h >> (2 * ib)) & 3) << 4);
                const float dl = d4d8 * (float)(ls - 32);
                int s1 = 0, s2 = 0;
                for (int j = 0; j < 16; ++j) {
                    s1 += y[i].qs[32 * ib + j]      * orc_iq4_codebook[x[i].qs[16 * ib + j] & 0x0F];
                    s2 += y[i].qs[32 * ib + j + 16] * orc_iq4_codebook[x[i].qs[16 * ib + j] >> 4]; }
                sumf += dl * (float)(s1 + s2); } }
        break; }
    default: return orc_iq_vec_dot(type, nb, out, vw, va);
    }
    *out = sumf;
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* MUL_MAT / MUL_MAT_ID                                                                   */
/* ------------------------------------------------------------------------------------ */
int orc_mul_mat(int type, const void *src0, const float *src1, float *dst,
                int64_t M, int64_t N, int64_t K,
                int64_t ne02, int64_t ne03, int64_t ne12, int64_t ne13, int mode) {
    const orc_geom *g = geom(type);
    if (!g || g->act < 0 || K % g->blck) return 1;
    if (ne02 <= 0 || ne03 <= 0 || ne12 % ne02 || ne13 % ne03) return 3;
    const int64_t wrow = orc_row_size(type, K);
    const int64_t arow = orc_row_size(g->act, K);
    const int64_t r2 = ne12 / ne02, r3 = ne13 / ne03;
    uint8_t *act = malloc((size_t)(arow > 0 ? arow : 1));
    if (!act) return 4;
    int rc = 0;
    for (int64_t i13 = 0; i13 < ne13 && !rc; ++i13)
    for (int64_t i12 = 0; i12 < ne12 && !rc; ++i12) {
        const uint8_t *w = (const uint8_t *) src0 + ((i13 / r3) * ne02 + (i12 / r2)) * M * wrow;
        for (int64_t n = 0; n < N && !rc; ++n) {
            const float *xcol = src1 + ((i13 * ne12 + i12) * N + n) * K;
            float *ycol = dst + ((i13 * ne12 + i12) * N + n) * M;
            if (g->act == ORC_Q8_K) memset(act, 0, (size_t) arow);   /* defined bsums for zero blocks */
            orc_quantize_row_act(g->act, xcol, act, K, mode);
            for (int64_t m = 0; m < M; ++m) rc |= orc_vec_dot(type, K, &ycol[m], w + m * wrow, act);
        }
    }
    free(act);
    return rc;
}

int orc_mul_mat_id(int type, const void *as, const float *b, const int32_t *ids, float *dst,
                   int64_t M, int64_t K, int64_t n_expert, int64_t n_used, int64_t n_tok,
                   int64_t b_ne1, int mode) {
    const orc_geom *g = geom(type);
    if (!g || g->act < 0 || K % g->blck) return 1;
    const int64_t wrow = orc_row_size(type, K);
    const int64_t arow = orc_row_size(g->act, K);
    uint8_t *act = malloc((size_t)(arow > 0 ? arow : 1));
    if (!act) return 4;
    int rc = 0;
    for (int64_t it = 0; it < n_tok; ++it)
        for (int64_t iu = 0; iu < n_used; ++iu) {
            const int32_t e = ids[it * n_used + iu];
            if (e < 0 || e >= n_expert) { free(act); return 5; }
            const float *xcol = b + (it * b_ne1 + (iu % b_ne1)) * K;
            float *ycol = dst + (it * n_used + iu) * M;
            if (g->act == ORC_Q8_K) memset(act, 0, (size_t) arow);
            orc_quantize_row_act(g->act, xcol, act, K, mode);
            const uint8_t *w = (const uint8_t *) as + (int64_t) e * M * wrow;
            for (int64_t m = 0; m < M; ++m) rc |= orc_vec_dot(type, K, &ycol[m], w + m * wrow, act);
        }
    free(act);
    return rc;
}
