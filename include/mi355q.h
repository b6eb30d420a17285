/* mi355q.h -- C-ABI of libmi355q.so: MI355X (gfx950 / CDNA4) kernels for the quantized
 * dequant + mat-vec / mat-mat multiply that dominates llama.cpp token generation.
 *
 * This is the drop-in boundary for the hot path.  Plain pointers and sizes only; no ggml, torch
 * or HIP types appear in any signature (streams are passed as void* = hipStream_t).
 *
 * What each entry point replaces in the reference (paths relative to /root/reference):
 *
 *   mi355q_mul_mat        GGML_OP_MUL_MAT with quantized src0, f32 src1, f32 dst
 *                           CPU : ggml_compute_forward_mul_mat      ggml/src/ggml-cpu/ggml-cpu.c:1266-1458
 *                           GPU : ggml_cuda_mul_mat -> mul_mat_vec_q / mul_mat_q / dequant+GEMM
 *                                 ggml/src/ggml-cuda/ggml-cuda.cu:1877-1938, mmvq.cu:130-288, mmq.cuh:2595-2674
 *   mi355q_mul_mat_id     GGML_OP_MUL_MAT_ID (MoE expert-indirect matmul)
 *                           CPU : ggml_compute_forward_mul_mat_id   ggml/src/ggml-cpu/ggml-cpu.c:1540-1718
 *                           GPU : ggml_cuda_mul_mat_id              ggml/src/ggml-cuda/ggml-cuda.cu:1994-2130
 *   mi355q_quantize_act   activation quantizer run before every quantized matmul
 *                           CPU : from_float of the vec_dot_type    ggml/src/ggml-cpu/ggml-cpu.c:1328-1363,
 *                                 quantize_row_q8_0/q8_1/q8_K       ggml/src/ggml-quants.c:194-253, 2479-2516
 *                           GPU : quantize_q8_1                     ggml/src/ggml-cuda/quantize.cu:4-38
 *   mi355q_weights_upload / _download
 *                         ggml_backend_buffer_i.set_tensor / get_tensor for quantized weights
 *                           ggml/src/ggml-backend-impl.h:41-66 (CUDA: ggml-cuda.cu:568-600); the device keeps a
 *                           coalescing-friendly per-row plane layout, invisible above this ABI
 *   mi355q_row_size / mi355q_type_supported
 *                         ggml_row_size / ggml_backend_device_i.supports_op
 *                           ggml/src/ggml.c (ggml_row_size), ggml/src/ggml-backend-impl.h:137-185
 *
 * The ggml backend plugin (libggml-mi355.so: ggml_backend_init / ggml_backend_score, the four vtables of
 * ggml/src/ggml-backend-impl.h) is a thin C++ host layer over this ABI; see INTEGRATION.md.
 *
 * Numerics contract: activations are quantized exactly as the reference CPU backend does for the weight
 * type (Q8_0 / Q8_1 for the 32-element formats, Q8_K for K-quants and IQ4_XS), every per-block integer dot
 * product is exact, and only the order of the final f32 additions differs from the CPU.
 *
 * All functions return 0 on success, a negative MI355Q_ERR_* otherwise (never throw, never abort).
 * A missing / non-gfx950 device is an error, not a fallback: there is no CPU path in this library.
 */
#ifndef MI355Q_H
#define MI355Q_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355Q_API_VERSION 3

/* error codes */
#define MI355Q_OK                0
#define MI355Q_ERR_UNSUPPORTED  -1   /* type / shape combination not implemented            */
#define MI355Q_ERR_SHAPE        -2   /* inconsistent sizes, K not a multiple of the block   */
#define MI355Q_ERR_ALIGN        -3   /* pointer / stride alignment contract violated        */
#define MI355Q_ERR_HIP          -4   /* a HIP runtime call failed (see mi355q_last_error)   */
#define MI355Q_ERR_WORKSPACE    -5   /* workspace too small                                 */
#define MI355Q_ERR_NO_DEVICE    -6   /* no usable gfx950 device                             */

/* weight / activation type ids are ggml's enum ggml_type values (ggml/include/ggml.h) */
#define MI355Q_TYPE_F32    0
#define MI355Q_TYPE_Q4_0   2
#define MI355Q_TYPE_Q4_1   3
#define MI355Q_TYPE_Q5_0   6
#define MI355Q_TYPE_Q5_1   7
#define MI355Q_TYPE_Q8_0   8
#define MI355Q_TYPE_Q8_1   9
#define MI355Q_TYPE_Q2_K   10
#define MI355Q_TYPE_Q3_K   11
#define MI355Q_TYPE_Q4_K   12
#define MI355Q_TYPE_Q5_K   13
#define MI355Q_TYPE_Q6_K   14
#define MI355Q_TYPE_Q8_K   15
#define MI355Q_TYPE_IQ2_XXS 16
#define MI355Q_TYPE_IQ2_XS  17
#define MI355Q_TYPE_IQ3_XXS 18
#define MI355Q_TYPE_IQ1_S   19
#define MI355Q_TYPE_IQ4_NL  20
#define MI355Q_TYPE_IQ3_S   21
#define MI355Q_TYPE_IQ2_S   22
#define MI355Q_TYPE_IQ4_XS  23
#define MI355Q_TYPE_IQ1_M   29

/* flags for mi355q_mul_mat* / mi355q_quantize_act */
#define MI355Q_FLAG_ROUND_AWAY   0x0  /* Q8_0/Q8_1 activations: q = roundf(x/d)   (quantize_row_q8_0_ref; default) */
#define MI355Q_FLAG_ROUND_EVEN   0x1  /*                        q = rint(x/d)     (what the AVX2 / NEON CPU paths do) */
#define MI355Q_FLAG_FORCE_GENERIC 0x2 /* debugging: use the generic (canonical-layout-style) kernel tier            */
#define MI355Q_FLAG_FORCE_GEMV   0x4  /* use the GEMV tier even for large N                                         */
#define MI355Q_FLAG_FORCE_MMQ    0x8  /* use the tiled MFMA tier even for small N                                   */

/* ---- library / device ------------------------------------------------------------------------ */
int          mi355q_api_version(void);
int          mi355q_device_count(void);                 /* number of visible gfx950 devices (0 if none)   */
int          mi355q_set_device(int device);
int          mi355q_device_info(int device, char *name, size_t name_len, size_t *free_bytes, size_t *total_bytes,
                                int *compute_units);
const char * mi355q_last_error(void);                   /* thread-local description of the last failure   */

/* ---- geometry ------------------------------------------------------------------------------------ */
int     mi355q_type_supported(int type);                /* 1 if `type` is accepted as src0 of mul_mat      */
int64_t mi355q_blck_size(int type);
int64_t mi355q_type_size(int type);
int64_t mi355q_row_size(int type, int64_t k);           /* == ggml_row_size(type, k); device rows use the same size */
int     mi355q_act_type(int type);                      /* activation format paired with `type` (CPU's vec_dot_type) */
int     mi355q_weights_are_planar(int type, int64_t k); /* 1 if device rows of (type,k) use the plane layout */

/* ---- thin device-memory helpers (so hosts without a HIP binding can drive the library) ----------- */
int mi355q_malloc(void **dev_ptr, size_t bytes);
int mi355q_free(void *dev_ptr);
int mi355q_memset(void *dev_ptr, int value, size_t bytes, void *stream);
int mi355q_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, void *stream);   /* stream NULL: synchronous */
int mi355q_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, void *stream);
int mi355q_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes, void *stream);
/* pinned (page-locked) host memory: staging for asynchronous set/get_tensor, ggml's host buffer type (ggml-backend-impl.h:137-185 get_host_buffer_type) */
int mi355q_host_malloc(void **host_ptr, size_t bytes);
int mi355q_host_free(void *host_ptr);
/* copy between two devices of this process (hipMemcpyPeerAsync: xGMI where the devices are linked), ordered on `stream` of the CURRENT device */
int mi355q_memcpy_peer(void *dst, int dst_device, const void *src, int src_device, size_t bytes, void *stream);
/* events (ggml_backend_event_*: ggml-backend-impl.h:87-124 event_record / event_wait, :137-185 event_new / event_free / event_synchronize) */
int mi355q_event_create(void **event);
int mi355q_event_destroy(void *event);
int mi355q_event_record(void *event, void *stream);
int mi355q_event_wait(void *stream, void *event);           /* makes `stream` (of any device) wait for the event */
int mi355q_event_synchronize(void *event);
int mi355q_stream_create(void **stream);
int mi355q_stream_destroy(void *stream);
int mi355q_stream_synchronize(void *stream);
int mi355q_device_synchronize(void);

/* ---- weights: canonical ggml rows <-> device rows -------------------------------------------------
 * `nrows` whole rows of `k` elements each.  Device rows have the same byte size and row stride as
 * canonical rows (so row-granular views and MoE expert offsets stay valid); only the order of bytes
 * INSIDE a row changes when mi355q_weights_are_planar(type,k).  dst/src device pointers must be
 * 16-byte aligned.  `stream` may be NULL (default stream); the call returns after the copy is complete. */
int mi355q_weights_upload(int type, void *dst_dev, const void *src_host, int64_t nrows, int64_t k, void *stream);
int mi355q_weights_download(int type, void *dst_host, const void *src_dev, int64_t nrows, int64_t k, void *stream);
/* device-to-device variants (canonical bytes already on the device <-> device rows) */
int mi355q_weights_pack_d2d(int type, void *dst_dev, const void *src_canonical_dev, int64_t nrows, int64_t k, void *stream);
int mi355q_weights_unpack_d2d(int type, void *dst_canonical_dev, const void *src_dev, int64_t nrows, int64_t k, void *stream);

/* ---- activation quantizer --------------------------------------------------------------------------
 * x: f32 device, `n` rows of `k` elements, row stride x_stride_bytes (multiple of 4).
 * out: device, n * mi355q_row_size(act_type,k) bytes, CANONICAL ggml block structs
 *      (block_q8_0 / block_q8_1 / block_q8_K, ggml-common.h:209-227, 329-334), bit-identical to the CPU. */
int mi355q_quantize_act(int act_type, const float *x, int64_t x_stride_bytes, void *out,
                        int64_t n, int64_t k, int flags, void *stream);

/* ---- MUL_MAT -----------------------------------------------------------------------------------------
 * y[n][m] = sum_k W[m][k] * x[n][k]       W: `m` device rows of type `type` (uploaded with
 * mi355q_weights_upload), row stride w_stride_bytes (>= row_size, multiple of 16);
 * x: f32 [n rows of k], row stride x_stride_bytes;  y: f32 [n rows of m], row stride y_stride_bytes.
 * workspace: device scratch of at least mi355q_mul_mat_workspace(type,m,n,k) bytes (may be NULL if 0). */
size_t mi355q_mul_mat_workspace(int type, int64_t m, int64_t n, int64_t k);
int    mi355q_mul_mat(int type, const void *w, int64_t w_stride_bytes,
                      const float *x, int64_t x_stride_bytes,
                      float *y, int64_t y_stride_bytes,
                      int64_t m, int64_t n, int64_t k,
                      void *workspace, size_t workspace_bytes, int flags, void *stream);

/* Several weight matrices against the SAME activations in one launch (wq/wk/wv, ffn_gate/ffn_up):
 * y_i[n][m_i] = W_i . x.  All W_i share type-independent k and n; types may differ per matrix.  */
typedef struct mi355q_mat {
    int         type;
    const void *w;
    int64_t     w_stride;   /* bytes between weight rows (>= row_size, multiple of 16 for planar rows) */
    float      *y;
    int64_t     y_stride;   /* bytes between output rows (one per activation row)                      */
    int64_t     m;
} mi355q_mat;
int    mi355q_mul_mat_multi(const mi355q_mat *mats, int n_mats,
                            const float *x, int64_t x_stride_bytes, int64_t n, int64_t k,
                            void *workspace, size_t workspace_bytes, int flags, void *stream);

/* ---- MUL_MAT_ID ---------------------------------------------------------------------------------------
 * y[t][u][:] = W[ids[t][u]] . x[t][u % x_ne1][:]
 * w: n_expert consecutive matrices of m device rows each (expert stride = expert_stride_bytes);
 * x: f32 [n_tok][x_ne1][k] contiguous rows with strides x_stride1_bytes (slot) and x_stride2_bytes (token);
 * ids: i32 device [n_tok][n_used] with row stride ids_stride_bytes;  y: f32 [n_tok][n_used][m] contiguous.
 * The expert ids are read ON THE DEVICE at every size (no host round trip: the call stays capturable).  Up to 16 (token, slot) pairs every
 * pair is one GEMV column; from 17 pairs on (prefill) the pairs are counting-sorted by expert on the device into tile-aligned segments,
 * gathered, multiplied by ONE launch of the matrix-core tier over all experts and scattered back (the reference groups on the host behind a
 * stream synchronize, ggml-cuda.cu:2008-2011); size the workspace with mi355q_mul_mat_id_workspace.  A pair whose expert id is out of range
 * [0, n_expert) -- the reference asserts on it -- gets an output row of NaN on every path. */
size_t mi355q_mul_mat_id_workspace(int type, int64_t m, int64_t k, int64_t n_used, int64_t n_tok, int64_t x_ne1, int64_t n_expert);
int    mi355q_mul_mat_id(int type, const void *w, int64_t w_stride_bytes, int64_t expert_stride_bytes, int64_t n_expert,
                         const float *x, int64_t x_ne1, int64_t x_stride1_bytes, int64_t x_stride2_bytes,
                         const int32_t *ids, int64_t ids_stride_bytes,
                         float *y, int64_t m, int64_t k, int64_t n_used, int64_t n_tok,
                         void *workspace, size_t workspace_bytes, int flags, void *stream);

/* ---- decode plan: the token's dependent chain as ONE persistent launch ----------------------------------------------
 * Replaces the reference's per-node launches + CUDA-graph replay of the token-generation graph
 * (ggml-cuda.cu:2470-2781, evaluate_and_capture_cuda_graph / ggml_backend_cuda_graph_compute): the N = 1 nodes of a
 * decode step -- the quantized MUL_MATs AND the glue between them -- run as an ordered list of STAGES inside one
 * cooperative launch of one workgroup per CU.
 *
 * GEMV stage: 1..4 weight matrices (planar device rows, types may differ) against one activation vector x[k];
 *   y_i[m_i] = W_i . x exactly as mi355q_mul_mat computes it (bit-identical for the same x).  x is formed in the
 *   stage's prologue from f32 vectors of k elements (every workgroup does it for itself, into LDS):
 *     MI355Q_X_PLAIN      x = x0
 *     MI355Q_X_NORM       t = x0 (+ x1);  [sum_out = t;]  x = rms_norm(t, eps) (* norm_w)     ADD -> RMS_NORM -> MUL of build_norm
 *     MI355Q_X_UNARY_MUL  x = unary(x0) * x1      (x_unary: MI355Q_UNARY_SILU / _RELU / _SIGMOID)   build_ffn LLM_FFN_SILU + LLM_FFN_PAR
 *   with the f32 operations of the CPU ops (same expressions as mi355q_op_add_rms_norm_mul / mi355q_op_unary_mul).
 * ATTN stage: rope of q and k, the K / V cache stores of this token and causal attention of ONE token over an f16 KV cache
 *   (ROPE, CPY, MUL_MAT(k,q), SOFT_MAX, MUL_MAT(v,kq) / FLASH_ATTN_EXT of build_attn_mha, src/llama-graph.cpp:1140-1260): see mi355q_attn.
 *
 * Dependencies are DATA-DRIVEN: an operand (x0, x1, q, k, v) whose address range lies inside an output (mats[i].y, sum_out,
 * attn.out) of an EARLIER stage of the plan is a value produced during the run: the producers publish every element as an
 * 8-byte {value, tag} granule in a plan-private buffer and the consumers poll the granules they need -- there is no grid-wide
 * barrier and no flag.  Any other operand must be complete before the launch.  Weights stream across stage boundaries.
 * Plain f32 copies of the outputs are stored at mats[i].y / sum_out / attn.out unless MI355Q_STAGE_NO_PLAIN is set (set it for
 * intermediates nobody outside the plan reads: a caller whose allocator reuses their memory must).
 * Only weight types with a planar layout at this k (mi355q_weights_are_planar) are accepted.
 * run(): asynchronous on `stream`.  status(): synchronizes; 1 = a poll timed out (plan unusable).                         */
#define MI355Q_STAGE_DEPENDS  0x1    /* (API version 1; accepted and ignored: dependencies follow from the operand addresses) */
#define MI355Q_STAGE_NO_PLAIN 0x2    /* do not store plain f32 copies of this stage's outputs                               */
#define MI355Q_STAGE_GEMV 0
#define MI355Q_STAGE_ATTN 1
#define MI355Q_X_PLAIN     0
#define MI355Q_X_NORM      1
#define MI355Q_X_UNARY_MUL 2
typedef struct mi355q_rope_params {      /* the op_params of GGML_OP_ROPE (ggml.c ggml_rope_impl) */
    int   n_dims, mode, n_ctx_orig;
    float freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow;
} mi355q_rope_params;
/* One token's attention.  q [n_head*head_dim], k, v [n_head_kv*head_dim]: f32 outputs of earlier stages (or complete before the launch).
 * Cache element (position j, dim d, kv head g) is an f16 at  k_cache + j*k_nb_pos + g*k_nb_head + 2*d   and
 * v_cache + j*v_nb_pos + d*v_nb_dim + g*v_nb_head  (v_nb_dim == 2: rows per position, the -fa layout; v_nb_pos == 2: the transposed
 * V cache of the non-flash graph).  This token's roped K row (f16, n_head_kv*head_dim contiguous) is stored at *k_dst and element
 * e of its V row at *v_dst + e*v_dst_nb; k_dst / v_dst are DEVICE slots holding the destination pointers, so that the caller moves the
 * store position every token without rebuilding the plan (cf. mi355q_op_cpy_indirect); the token attends to positions [0, n_kv) with
 * the additive mask row `mask` (f32 or f16, n_kv entries, may be NULL = no masking), as soft_max_ext(kq*scale + mask) does; the row
 * being stored is read from the registers, not from the cache.  out [n_head*head_dim] f32.  rope.mode 0 (normal) or 2 (neox).           */
typedef struct mi355q_attn {
    const float   *q, *k, *v;
    const int32_t *pos;              /* device: the token's position (rope angle)                */
    mi355q_rope_params rope;
    const float   *freq_factors;     /* optional rope frequency factors [n_dims/2]               */
    const void    *k_cache, *v_cache;
    int64_t        k_nb_pos, k_nb_head, v_nb_pos, v_nb_dim, v_nb_head;
    void * const  *k_dst, * const *v_dst;
    int64_t        v_dst_nb;
    const void    *mask;
    int            mask_f16;
    int            n_head, n_head_kv, head_dim, n_kv;
    float          scale;
    float         *out;
    const int32_t *n_kv_dev;         /* optional, device: the window length of THIS run (<= n_kv, which then is the maximum the plan is sized for) */
    int64_t        q_id, k_id, v_id, out_id;   /* value ids (see mi355q_stage); 0 = identify by address */
} mi355q_attn;
typedef struct mi355q_stage {
    mi355q_mat   mats[4];    /* GEMV: y_stride is unused (one activation row) */
    int          n_mats;
    int          flags;
    const float *x;          /* GEMV: x0, device, k floats */
    int64_t      k;
    /* --- API version 2 --- */
    int          kind;       /* MI355Q_STAGE_GEMV (0) / MI355Q_STAGE_ATTN */
    int          x_kind;     /* MI355Q_X_* */
    int          x_unary;
    float        eps;
    const float *x1;         /* second operand (X_NORM: optional addend; X_UNARY_MUL: the multiplier) */
    const float *norm_w;     /* X_NORM: optional weight vector [k] (complete before the launch)        */
    float       *sum_out;    /* X_NORM with x1: where t = x0 + x1 is stored (the next residual's operand), optional */
    const mi355q_attn *attn; /* MI355Q_STAGE_ATTN */
    /* VALUE IDS (optional, 0 = unused).  By default an operand is matched to the earlier output whose ADDRESS range contains it.  A caller
     * whose allocator hands the same memory to several intermediates of one graph (ggml_gallocr does: K's projection reuses the dead pre-rope
     * Q's block, V's reuses K's) cannot be matched that way once the plan reorders consumers behind later producers.  Such a caller labels
     * every output with a unique non-zero id (y_id[i] for mats[i].y, sum_id for sum_out, attn->out_id) and every operand with the id of the
     * value it means (x_id, x1_id, attn->q_id / k_id / v_id; 0 = a plain vector complete before the launch): ids then decide, addresses
     * only give the offset of a sub-vector inside the labelled output.                                                                       */
    int64_t      y_id[4], sum_id, x_id, x1_id;
    /* OUTPUT FORM.  MI355Q_Y_ROWS (0): y_i = W_i . x, one vector per matrix.  MI355Q_Y_UNARY_MUL: n_mats == 2, same type and m; the stage
     * publishes ONE vector  y = unary(W_0 . x) * (W_1 . x)  of m elements at mats[0].y (id y_id[0]; mats[1].y is unused): UNARY -> MUL of
     * build_ffn (LLM_FFN_SILU + LLM_FFN_PAR) folded into the producer, so that ffn_down gathers one vector instead of two.  Every workgroup
     * computes matching rows of both matrices; same f32 expressions as mi355q_op_unary_mul.                                                   */
    int          y_kind, y_unary;
    /* --- API version 3 --- */
    float       *x_out;      /* X_NORM / X_UNARY_MUL: optional; the activation vector the prologue forms (rms_norm(.) * w, unary(.) * x1) is ALSO
                              * stored here as plain f32 -- for a caller whose graph hands that vector to somebody besides this stage's matrices
                              * (llama.cpp: result_norm is read back as the embeddings; a LoRA branch multiplies it again).  One workgroup stores it. */
} mi355q_stage;
#define MI355Q_Y_ROWS      0
#define MI355Q_Y_UNARY_MUL 1
typedef struct mi355q_plan mi355q_plan;
int     mi355q_plan_create(mi355q_plan **out, const mi355q_stage *stages, int n_stages, int flags);
int     mi355q_plan_run(mi355q_plan *plan, void *stream);
int     mi355q_plan_status(mi355q_plan *plan);
/* test hook: set the plan's run counter (the source of the granule tags' epoch); a value near the 32-bit wrap exercises the granule reset */
int     mi355q_plan_debug_set_runs(mi355q_plan *plan, unsigned long long runs);
/* diagnostics of an aborted plan: its 32 sync words (0: abort flag; 1..7: kind of the wait that gave up, stage, workgroup, wave, two operands, landed pages) */
int     mi355q_plan_debug_words(mi355q_plan *plan, unsigned *out32);
/* asynchronous form: enqueues a 4-byte device-to-host copy of the plan's abort word into *host_flag (pinned memory recommended) on `stream`;
 * after the stream has been synchronized, *host_flag != 0 means a poll of an earlier run timed out.                                       */
int     mi355q_plan_status_async(mi355q_plan *plan, unsigned *host_flag, void *stream);
int64_t mi355q_plan_weight_bytes(const mi355q_plan *plan);
int     mi355q_plan_launch_stages(const mi355q_plan *plan);
int     mi355q_plan_destroy(mi355q_plan *plan);

/* ---- residency ops: the small f32/f16 graph ops between the quantized matmuls of a decode graph ------
 * Tensors are described the ggml way: ne[] elements per dimension, nb[] strides in BYTES (views, permutes and broadcasts
 * are expressed through them), data = device pointer.  type: MI355Q_T_F32 / MI355Q_T_F16.  Each function is the device
 * counterpart of one ggml CPU op and follows its arithmetic (cited in csrc/ops_glue.hip):
 *   op_bin_bcast  GGML_OP_ADD/SUB/MUL/DIV   dst = a (op) b, b broadcast over a (ne_a % ne_b == 0), dst has a's shape
 *   op_unary      GGML_OP_UNARY             SILU, RELU, SIGMOID, TANH, NEG, ABS
 *   op_rms_norm   GGML_OP_RMS_NORM          per row: x / sqrt(mean(x^2) + eps), f32, rows contiguous
 *   op_cpy        GGML_OP_CPY/CONT/DUP      logical element order, any strides, f32 <-> f16
 *   op_soft_max   GGML_OP_SOFT_MAX          softmax(a*scale + slope*mask) per row; mask f32/f16 [ne00, >= ne01] or NULL
 *   op_rope       GGML_OP_ROPE              normal (mode 0) and neox (mode 2) rotary embedding, YaRN parameters, optional
 *                                           frequency factors (src2); pos = i32 device [ne2]
 *   op_mul_mat_f  GGML_OP_MUL_MAT           with an f16 or f32 src0 (attention KQ / KQV): dst[m,n] = sum_k a[k,m] b[k,n],
 *                                           dims 2/3 of src0 broadcast; an f16 src0 rounds src1 to f16 as the CPU does          */
#define MI355Q_T_F32 0
#define MI355Q_T_F16 1
typedef struct mi355q_tensor {
    void   *data;
    int     type;
    int64_t ne[4];
    int64_t nb[4];
} mi355q_tensor;
#define MI355Q_OP_ADD 1
#define MI355Q_OP_SUB 2
#define MI355Q_OP_MUL 3
#define MI355Q_OP_DIV 4
#define MI355Q_UNARY_SILU    1
#define MI355Q_UNARY_RELU    2
#define MI355Q_UNARY_SIGMOID 3
#define MI355Q_UNARY_TANH    4
#define MI355Q_UNARY_NEG     5
#define MI355Q_UNARY_ABS     6
int mi355q_op_bin_bcast(int op, const mi355q_tensor *a, const mi355q_tensor *b, const mi355q_tensor *dst, void *stream);
int mi355q_op_unary(int uop, const mi355q_tensor *a, const mi355q_tensor *dst, void *stream);
int mi355q_op_rms_norm(const mi355q_tensor *a, const mi355q_tensor *dst, float eps, void *stream);
/* Fusions around the path (the reference's graph emits them as separate nodes: build_norm = RMS_NORM then MUL,
 * src/llama-graph.cpp; build_ffn LLM_FFN_SILU + LLM_FFN_PAR = UNARY then MUL).  Same f32 operations in the same order as the
 * separate ops, so results are bit-identical to issuing the nodes one by one.
 *   add_rms_norm_mul: x = b ? a + b : a;  if (sum) *sum = x;  dst = rms_norm(x, eps) [* weight[ne0]]     (b, sum, weight may be NULL)
 *   unary_mul:        dst = unary(a) * b          (SiLU / ReLU / sigmoid; contiguous f32 tensors of one shape)                  */
int mi355q_op_add_rms_norm_mul(const mi355q_tensor *a, const mi355q_tensor *b, const mi355q_tensor *sum, const float *weight,
                               const mi355q_tensor *dst, float eps, void *stream);
int mi355q_op_unary_mul(int uop, const mi355q_tensor *a, const mi355q_tensor *b, const mi355q_tensor *dst, void *stream);
/* GGML_OP_FLASH_ATTN_EXT (ggml.c ggml_flash_attn_ext; CPU ggml-cpu/ops.cpp:6690-6905) for an f16 KV cache:
 *   q f32 [DK, N, H, B], k f16 [DK, n_kv, Hk, Bk], v f16 [DV, n_kv, Hv, Bv] (NOT transposed), mask f16 [>= n_kv, >= N] or NULL,
 *   dst f32 [DV, H, N, B];  dst = softmax(softcap(scale k.q) + slope_h mask) v,  head sizes <= 256, n_kv <= 36864.
 * workspace (optional, mi355q_op_flash_attn_ext_workspace bytes): with it, a call with few query rows (decode) splits the KV range over
 * several workgroups per row and merges the pieces in a second launch, and a call with >= 16 query rows (prefill) computes scores,
 * softmax and P V on the matrix-core kernels with the scores of the batch in the workspace; without it every row is one workgroup.   */
size_t mi355q_op_flash_attn_ext_workspace(int64_t dv, int64_t n_q, int64_t n_head, int64_t n_batch, int64_t n_kv);
int mi355q_op_flash_attn_ext(const mi355q_tensor *q, const mi355q_tensor *k, const mi355q_tensor *v, const mi355q_tensor *mask,
                             const mi355q_tensor *dst, float scale, float max_bias, float logit_softcap,
                             void *workspace, size_t workspace_bytes, void *stream);
int mi355q_op_cpy(const mi355q_tensor *a, const mi355q_tensor *dst, void *stream);
int mi355q_op_soft_max(const mi355q_tensor *a, const mi355q_tensor *mask, const mi355q_tensor *dst, float scale, float max_bias, void *stream);
int mi355q_op_rope(const mi355q_tensor *a, const int32_t *pos, const float *freq_factors, const mi355q_tensor *dst,
                   const mi355q_rope_params *p, void *stream);
int mi355q_op_mul_mat_f(const mi355q_tensor *a, const mi355q_tensor *b, const mi355q_tensor *dst, void *stream);
/* GGML_OP_GET_ROWS: dst[:, i10, i11, i12] = a[:, ids[i10, i11, i12], i11, i12]; a f32/f16, ids i32 (tensor with type field ignored), dst f32.
 * GGML_OP_SCALE:    dst = a * scale (f32). */
/* op_cpy whose destination base pointer is read ON THE DEVICE from dest_table[index] (dst->data is ignored): lets a captured
 * launch graph be replayed while the KV-cache store position moves every token (cf. ggml-cuda.cu cpy_dest_ptrs). */
int mi355q_op_cpy_indirect(const mi355q_tensor *a, const mi355q_tensor *dst, void *const *dest_table, int index, void *stream);
int mi355q_op_get_rows(const mi355q_tensor *a, const mi355q_tensor *ids, const mi355q_tensor *dst, void *stream);
/* The MoE router ops of build_moe_ffn (src/llama-graph.cpp:824-965): GGML_OP_ARGSORT (ggml_top_k = argsort descending + a view of the first k):
 * dst i32, same shape as a (f32), dst[row][r] = index of the element of rank r of the row (ties keep index order; ggml-cuda/argsort.cu:11, CPU
 * ops.cpp ggml_compute_forward_argsort_f32); GGML_OP_SUM_ROWS: dst[0, i1, i2, i3] = sum_i0 a[i0, i1, i2, i3], accumulated in f64 as the CPU does. */
int mi355q_op_argsort(const mi355q_tensor *a, const mi355q_tensor *dst, int descending, void *stream);
int mi355q_op_sum_rows(const mi355q_tensor *a, const mi355q_tensor *dst, void *stream);
int mi355q_op_scale(const mi355q_tensor *a, const mi355q_tensor *dst, float scale, void *stream);

/* ---- launch graphs: capture everything enqueued on `stream` between begin and end, replay it with one call ---------------
 * (hipStreamBeginCapture / hipStreamEndCapture / hipGraphInstantiate / hipGraphLaunch; the reference's counterpart is the CUDA
 * graph path of ggml-cuda.cu:2470-2781).  No allocation or synchronization may happen between begin and end.             */
typedef struct mi355q_graph mi355q_graph;
int mi355q_graph_capture_begin(void *stream);
int mi355q_graph_capture_end(void *stream, mi355q_graph **out);     /* on failure the stream leaves capture mode, *out = NULL */
int mi355q_graph_launch(mi355q_graph *graph, void *stream);
int mi355q_graph_destroy(mi355q_graph *graph);

#ifdef __cplusplus
}
#endif
#endif /* MI355Q_H */
