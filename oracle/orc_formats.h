/* oracle/orc_formats.h -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Byte layouts of the packed blocks, restated from ggml/src/ggml-common.h:167-418.
 * All structs are byte-packed; f16 fields are kept as raw uint16_t. */
#ifndef ORC_FORMATS_H
#define ORC_FORMATS_H
#include <stdint.h>

#define ORC_QK   32     /* elements per "small" block  (QK4_0 .. QK8_1, QK4_NL) */
#define ORC_QKK  256    /* elements per super-block    (QK_K)                  */

#pragma pack(push, 1)
typedef struct { uint16_t d;               uint8_t qs[16]; }             orc_q4_0;   /* 18 B  :167-172 */
typedef struct { uint16_t d, m;            uint8_t qs[16]; }             orc_q4_1;   /* 20 B  :174-185 */
typedef struct { uint16_t d;  uint8_t qh[4]; uint8_t qs[16]; }           orc_q5_0;   /* 22 B  :187-193 */
typedef struct { uint16_t d, m; uint8_t qh[4]; uint8_t qs[16]; }         orc_q5_1;   /* 24 B  :195-207 */
typedef struct { uint16_t d;               int8_t  qs[32]; }             orc_q8_0;   /* 34 B  :209-214 */
typedef struct { uint16_t d, s;            int8_t  qs[32]; }             orc_q8_1;   /* 36 B  :216-227 */
typedef struct { uint8_t scales[16]; uint8_t qs[64]; uint16_t d, dmin; } orc_q2_K;   /* 84 B  :257-268 */
typedef struct { uint8_t hmask[32]; uint8_t qs[64]; uint8_t scales[12]; uint16_t d; } orc_q3_K; /* 110 B :274-280 */
typedef struct { uint16_t d, dmin; uint8_t scales[12]; uint8_t qs[128]; }            orc_q4_K;  /* 144 B :285-296 */
typedef struct { uint16_t d, dmin; uint8_t scales[12]; uint8_t qh[32]; uint8_t qs[128]; } orc_q5_K; /* 176 B :302-314 */
typedef struct { uint8_t ql[128]; uint8_t qh[64]; int8_t scales[16]; uint16_t d; }   orc_q6_K;  /* 210 B :320-326 */
typedef struct { float d; int8_t qs[256]; int16_t bsums[16]; }                        orc_q8_K;  /* 292 B :329-334 */
typedef struct { uint16_t d; uint8_t qs[16]; }                                        orc_iq4_nl; /* 18 B  :405-409 */
typedef struct { uint16_t d; uint16_t scales_h; uint8_t scales_l[4]; uint8_t qs[128]; } orc_iq4_xs; /* 136 B :411-417 */
#pragma pack(pop)

_Static_assert(sizeof(orc_q4_0) == 18 && sizeof(orc_q4_1) == 20 && sizeof(orc_q5_0) == 22 &&
               sizeof(orc_q5_1) == 24 && sizeof(orc_q8_0) == 34 && sizeof(orc_q8_1) == 36, "small blocks");
_Static_assert(sizeof(orc_q2_K) == 84 && sizeof(orc_q3_K) == 110 && sizeof(orc_q4_K) == 144 &&
               sizeof(orc_q5_K) == 176 && sizeof(orc_q6_K) == 210 && sizeof(orc_q8_K) == 292, "k blocks");
_Static_assert(sizeof(orc_iq4_nl) == 18 && sizeof(orc_iq4_xs) == 136, "iq4 blocks");

/* the 16-entry non-linear code book of IQ4_NL / IQ4_XS (ggml-quants.c:2434) */
static const int8_t orc_iq4_codebook[16] = {
    -127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113
};

#endif
