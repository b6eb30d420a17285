/* oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, no ggml) of the reference's quantized mul_mat arithmetic:
 * the block decoders, the activation quantizers and the per-type integer dot products of
 * the ggml CPU backend, in their ISA-independent ("scalar spec") form, so that results are
 * bit-identical to the reference CPU backend built without SIMD (oracle/_ref/scalar).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (llama.cpp.dsp_amd/) never links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py checks every function below
 * bit-for-bit against oracle/_ref/scalar (the real reference compiled from
 * /root/reference by oracle/Makefile) and the committed fixtures in tests/golden/ were
 * produced by that reference (tests/golden/make_golden.py).
 *
 * Type ids are ggml's (ggml/include/ggml.h enum ggml_type).
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum orc_type {
    ORC_F32 = 0, ORC_F16 = 1,
    ORC_Q4_0 = 2, ORC_Q4_1 = 3, ORC_Q5_0 = 6, ORC_Q5_1 = 7, ORC_Q8_0 = 8, ORC_Q8_1 = 9,
    ORC_Q2_K = 10, ORC_Q3_K = 11, ORC_Q4_K = 12, ORC_Q5_K = 13, ORC_Q6_K = 14, ORC_Q8_K = 15,
    ORC_IQ2_XXS = 16, ORC_IQ2_XS = 17, ORC_IQ3_XXS = 18, ORC_IQ1_S = 19, ORC_IQ4_NL = 20,
    ORC_IQ3_S = 21, ORC_IQ2_S = 22, ORC_IQ4_XS = 23, ORC_IQ1_M = 29,
};

/* rounding of q = round(x/d) in the 32-block activation quantizers:
 *   ORC_ROUND_AWAY : roundf(), quantize_row_q8_0_ref  (ggml-quants.c:194-217)  -- scalar spec
 *   ORC_ROUND_EVEN : nearbyint(), what the AVX2/NEON paths do (ggml-cpu-quants.c:842-845)   */
enum orc_round { ORC_ROUND_AWAY = 0, ORC_ROUND_EVEN = 1 };

/* ---- format geometry (ggml-common.h:167-418) ------------------------------------------ */
int     orc_supported(int type);              /* 1 if this oracle implements `type` as src0 */
int64_t orc_blck_size(int type);              /* elements per block                          */
int64_t orc_type_size(int type);              /* bytes per block                             */
int64_t orc_row_size(int type, int64_t k);    /* bytes of a row of k elements                */
int     orc_vec_dot_type(int type);           /* activation format the CPU pairs with `type`
                                                 (ggml-cpu.c:211-376)                         */

/* ---- f16 <-> f32 (ggml-impl.h:441-489 semantics: IEEE, round-to-nearest-even) --------- */
float    orc_f16_to_f32(uint16_t h);
uint16_t orc_f32_to_f16(float f);

/* ---- weight block decode: dequantize_row_<type> (ggml-quants.c) ----------------------- */
int orc_dequantize_row(int type, const void *src, float *dst, int64_t k);

/* ---- activation quantizers ------------------------------------------------------------- */
void orc_quantize_row_q8_0(const float *x, void *y, int64_t k, int round_mode); /* ggml-quants.c:194-217  */
void orc_quantize_row_q8_1(const float *x, void *y, int64_t k, int round_mode); /* ggml-quants.c:220-253  */
void orc_quantize_row_q8_K(const float *x, void *y, int64_t k);                 /* ggml-quants.c:2479-2516 */
int  orc_quantize_row_act(int act_type, const float *x, void *y, int64_t k, int round_mode);

/* ---- one weight row . one quantized activation row (scalar tails of ggml_vec_dot_*,
 *      ggml-cpu-quants.c; exact float operation order) -------------------------------- */
int orc_vec_dot(int type, int64_t k, float *out, const void *w_row, const void *act_row);

/* ---- whole op: ggml_compute_forward_mul_mat (ggml-cpu.c:1266-1458) -------------------
 * src0: quantized [ne00=K, ne01=M, ne02, ne03], contiguous.
 * src1: f32       [K, N, ne12, ne13], contiguous.    dst: f32 [M, N, ne12, ne13].
 * Broadcast: i02 = i12 / (ne12/ne02), i03 = i13 / (ne13/ne03)  (ggml-cpu.c:1197-1198).   */
int orc_mul_mat(int type, const void *src0, const float *src1, float *dst,
                int64_t M, int64_t N, int64_t K,
                int64_t ne02, int64_t ne03, int64_t ne12, int64_t ne13, int round_mode);

/* ---- ggml_compute_forward_mul_mat_id (ggml-cpu.c:1540-1718) --------------------------
 * as : quantized [K, M, n_expert];  b: f32 [K, b_ne1, n_tok] (b_ne1 == n_used or 1);
 * ids: i32 [n_used, n_tok];        dst: f32 [M, n_used, n_tok]:
 *   dst[:, iu, it] = as[:, :, ids[iu,it]] . b[:, iu % b_ne1, it]                          */
int orc_mul_mat_id(int type, const void *as, const float *b, const int32_t *ids, float *dst,
                   int64_t M, int64_t K, int64_t n_expert, int64_t n_used, int64_t n_tok,
                   int64_t b_ne1, int round_mode);

#ifdef __cplusplus
}
#endif
#endif
