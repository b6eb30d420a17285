// plan.hip -- a token-generation step as ONE persistent launch (MI355X-first replacement for the reference's "one kernel
// per node + CUDA graph" decode loop, ggml-cuda.cu:2470-2781): the N = 1 quantized matmuls AND the glue between them.
//
// A PLAN is an ordered list of STAGES, walked by every workgroup of one launch of #CU workgroups x 16 waves (one per CU, all resident):
//
//   GEMV    up to 4 planar weight matrices of one type against one activation vector x.  The prologue forms x in LDS --
//           x0 | rms_norm(x0 + x1) * w | unary(x0) * x1, quantized exactly as the CPU does (act_quant.cuh) -- then the
//           workgroup streams its rows; arithmetic per row is gemv_fast.hip's (gemv_stream.cuh is shared): bit-identical.
//   ATTN    rope(q), rope(k), this token's K / V cache stores and causal attention of the one token over the f16 cache,
//           one workgroup per (head, KV split); COMBINE merges the splits of a head (log-sum-exp).
//
//   * DATAFLOW, not barriers.  A value produced during the run is published element by element as an 8-byte
//     {f32 value, u32 tag} GRANULE (one naturally aligned agent-scope store; tag = launch epoch + producing stage) in a
//     plan-private buffer; a consumer polls exactly the granules it needs with agent-scope (sc1) loads until every tag
//     matches.  No counter, no flag, no release/acquire fence, no store acknowledgement: the hand-off is the data
//     (MI355X_MICROARCH.md, hand-off price list: a data-tagged granule is the cheapest cross-CU edge; a flag/barrier
//     protocol costs 1.7-2.5x).  Round 1's two-level grid barrier + separate activation fetch cost 8-9 us per dependent
//     step; see DESIGN.md section 5.6 for the measured chain now.
//   * the weight stream does not stop at a dependency: a wave requests the first chunks of its rows of stage s+1 as soon
//     as it has finished stage s, BEFORE it polls for the activations.
//   * no re-arming: tags grow monotonically over launches (the host re-zeroes the granules before the 32-bit epoch wraps).
//   * every poll has a wall-clock bound (s_memrealtime): on timeout the workgroup raises the plan's sticky abort flag and
//     returns, so the grid always drains (one workgroup per CU: all are resident unless another persistent kernel holds CUs).
//
// Bound: HBM read of W.  Algorithmic bytes per launch = sum over stages, matrices of m * row_size(type, k) (+ the KV cache
// window of ATTN stages: 2 * n_kv * n_head_kv * head_dim * 2 bytes).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <vector>
#include <cstring>
#include <chrono>
#include <mutex>

#include "gemv_stream.cuh"

namespace mi355q {

constexpr int PLAN_D_MAX = 8;                     // ring depth (1-KiB steps in flight per wave)
__host__ __device__ constexpr int plan_depth(int type) {   // Q5_K / Q6_K slots carry qh too (1.5 KiB per step): 6 steps are the bytes of 8 Q4_K steps
    return (type == MI355Q_TYPE_Q5_K || type == MI355Q_TYPE_Q6_K) ? 6 : PLAN_D_MAX;
}
enum { PLAN_F_NEW_X = 2, PLAN_F_PLAIN_Y = 4, PLAN_F_SUM = 8, PLAN_F_SUM_PLAIN = 16, PLAN_F_PAIRED = 32, PLAN_F_DIRECT = 64, PLAN_F_ATTN_LDS = 128 };
enum { PLAN_K_GEMV = 0, PLAN_K_ATTN = 1, PLAN_K_COMBINE = 2 };
enum { PLAN_SYNC_ABORT = 0, PLAN_SYNC_WORDS = 32 };

typedef unsigned long long Granule;               // low dword: f32 value bits, high dword: tag

// an operand vector: plain f32 (complete before the launch) or granules published by an earlier stage (tag_off = its index + 1)
struct VecSrc { const float * plain; const Granule * gran; unsigned tag_off, pad; };

struct AttnStage {
    VecSrc q, k, v;
    const int32_t * pos, * n_kv_dev;
    const float * freq_factors;
    const char * k_cache, * v_cache;
    int64_t k_nb_pos, k_nb_head, v_nb_pos, v_nb_dim, v_nb_head, v_dst_nb;
    char * const * k_dst, * const * v_dst;
    const char * mask;
    Granule * part;                                // [n_head][n_split][head_dim + 2]  (o, m, l) of every split
    Granule * out_gran; float * out_plain;
    int mask_f16, n_head, n_head_kv, hd, n_kv, n_split, per, plain, p_f16;
    int kv_lds;                                    // the split's K / V window fits the LDS: plan_attn_lds runs the stage
    int fa_seq;                                    // rows-per-position V cache, window in one workgroup: the CPU's sequential f16 accumulation (plan_attn_lds)
    int entry_barrier;                             // the previous stage is not a GEMV: its last LDS reads must be fenced off
    float scale;
    // rope (ggml_rope_cache_init; see ops_glue.hip k_rope)
    int n_dims, neox; float freq_scale, ext_factor, attn_factor, theta_scale, corr0, corr1;
};

struct alignas(64) PlanStage {
    // -- what the streamers need per row, contiguous (arrives with a few scalar loads issued together) --
    const uint8_t * w[GEMV_MAX_MATS];
    int64_t         w_stride[GEMV_MAX_MATS];
    float *         y[GEMV_MAX_MATS];
    Granule *       yg;                           // granules of the stage's outputs, indexed by CONCATENATED row
    int             row_begin[GEMV_MAX_MATS];     // first concatenated row of each matrix (unused entries: INT_MAX)
    int             total_rows, rows_per_wg, k, n_mats;
    // -- the rest --
    int             type, flags, prime, kind;
    int             wg_base, wg_count, group, pub_wg;     // pub_wg: the workgroup of the range that publishes the stage's residual sum / x_out. the workgroups [wg_base, wg_base + wg_count) run this stage; group > 1: this and the next group - 1 descriptors run CONCURRENTLY on disjoint workgroups
    VecSrc          x0, x1;
    const float *   norm_w;
    Granule *       sum_gran; float * sum_plain; float * x_out;
    const AttnStage * attn;
    const AttnStage * next_attn;                  // the attention descriptor of the NEXT stage (prefetched with it), or null
    float           eps; int x_kind, x_unary; unsigned tag_off;
};
typedef const __attribute__((address_space(4))) PlanStage * StageC;   // descriptors are read with scalar loads

// type sets a kernel instantiation can stream (register allocation is the max over the set)
constexpr unsigned tbit(int t) { return 1u << t; }
constexpr unsigned SET_K46  = tbit(MI355Q_TYPE_Q4_K) | tbit(MI355Q_TYPE_Q6_K);
constexpr unsigned SET_K456 = SET_K46 | tbit(MI355Q_TYPE_Q5_K);
constexpr unsigned SET_80   = tbit(MI355Q_TYPE_Q8_0) | tbit(MI355Q_TYPE_Q4_0);
constexpr unsigned SET_ALL  = SET_K456 | SET_80;
// IQ4_XS / IQ4_NL models (llama-quant.cpp: IQ4_XS or IQ4_NL for most tensors, Q5_K / Q6_K for attn_v, some ffn_down and the output matrix)
constexpr unsigned SET_IQ4  = tbit(MI355Q_TYPE_IQ4_XS) | tbit(MI355Q_TYPE_IQ4_NL) | tbit(MI355Q_TYPE_Q5_K) | tbit(MI355Q_TYPE_Q6_K);
constexpr unsigned SET_ANY  = SET_ALL | SET_IQ4;

struct PlanCursor { int gr, s; const uint8_t * row; };

#ifdef MI355Q_STAMPS
// Diagnostic build only (libmi355q_dbg.so): waves 0 and 15 of every workgroup record 100 MHz wall-clock stamps per stage:
// g_plan_stamps[((stage*grid + wg)*2 + (wave==15))*8 + i].  The product library contains none of this.
__device__ unsigned long long * g_plan_stamps = nullptr;
__device__ int g_plan_stamp_stages = 0;
#define PLAN_STAMP(i) do { if (g_plan_stamps && lane == 0 && (wave == 0 || wave == GEMV_WAVES - 1) && c.stage < g_plan_stamp_stages) \
    g_plan_stamps[(((size_t) c.stage * c.grid + blockIdx.x) * 2 + (wave ? 1 : 0)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PLAN_STAMP(i) do { } while (0)
#endif

// A wave's own copy of the per-matrix fields of a stage descriptor, in SGPRs: read once per stage with independent
// scalar loads (one round trip).  Reading them per row instead (matrix index -> base -> stride: a dependent chain of
// scalar loads) costs microseconds whenever the descriptors miss the scalar cache.
struct StageW {
    const uint8_t * w0, * w1, * w2, * w3; int64_t ws0, ws1, ws2, ws3; float * y0, * y1, * y2, * y3; Granule * g; int rb1, rb2, rb3; int paired;
    __device__ __forceinline__ void load(StageC st) {          // what the loader needs (live through the prologue)
        paired = 0;
        w0 = st->w[0]; w1 = st->w[1]; w2 = st->w[2]; w3 = st->w[3];
        ws0 = st->w_stride[0]; ws1 = st->w_stride[1]; ws2 = st->w_stride[2]; ws3 = st->w_stride[3];
        rb1 = st->row_begin[1]; rb2 = st->row_begin[2]; rb3 = st->row_begin[3];
    }
    // PAIRED stage (y = unary(W0 x) * (W1 x)): this workgroup's LOCAL row 2 q is row p0 + q of matrix 0, local row 2 q + 1 the same row of matrix 1.
    // A wave takes pairs q = wave, wave + 16, ... and streams the two rows of a pair back to back (plan_next_row), so it holds both dot products
    // itself and publishes unary(.) * (.) without a trip through LDS.
    __device__ __forceinline__ void load_paired(StageC st, int p0, int np) {
        paired = 1;
        ws0 = st->w_stride[0]; ws1 = st->w_stride[1]; ws2 = ws3 = 0;
        w0 = st->w[0] + (int64_t) p0 * ws0; w1 = st->w[1] + (int64_t) p0 * ws1; w2 = w3 = nullptr;
        rb1 = np; rb2 = rb3 = 0x7FFFFFFF;
    }
    __device__ __forceinline__ void load_out(StageC st) {      // what the consumer needs (read after the prologue)
        y0 = st->y[0]; y1 = st->y[1]; y2 = st->y[2]; y3 = st->y[3]; g = st->yg;
    }
    // all fields are read BEFORE the selects (a select between field addresses would keep the struct in scratch memory)
    __device__ __forceinline__ const uint8_t * row_ptr(int r) const {
        const uint64_t a0 = (uint64_t) w0, a1 = (uint64_t) w1, a2 = (uint64_t) w2, a3 = (uint64_t) w3;
        const int64_t s0 = ws0, s1 = ws1, s2 = ws2, s3 = ws3;
        const int b1 = rb1, b2 = rb2, b3 = rb3;
        if (paired) return (const uint8_t *) (((r & 1) ? a1 : a0) + (uint64_t) ((int64_t) (r >> 1) * ((r & 1) ? s1 : s0)));
        const int mi = (r >= b1) + (r >= b2) + (r >= b3);      // row_begin is ascending
        const uint64_t w = mi == 0 ? a0 : mi == 1 ? a1 : mi == 2 ? a2 : a3;
        const int64_t ws = mi == 0 ? s0 : mi == 1 ? s1 : mi == 2 ? s2 : s3;
        const int rb = mi == 0 ? 0 : mi == 1 ? b1 : mi == 2 ? b2 : b3;
        return (const uint8_t *) (w + (uint64_t) ((int64_t) (r - rb) * ws));
    }
    __device__ __forceinline__ float * y_ptr(int r) const {
        const uint64_t a0 = (uint64_t) y0, a1 = (uint64_t) y1, a2 = (uint64_t) y2, a3 = (uint64_t) y3;
        const int b1 = rb1, b2 = rb2, b3 = rb3;
        const int mi = (r >= b1) + (r >= b2) + (r >= b3);
        const uint64_t y = mi == 0 ? a0 : mi == 1 ? a1 : mi == 2 ? a2 : a3;
        const int rb = mi == 0 ? 0 : mi == 1 ? b1 : mi == 2 ? b2 : b3;
        return (float *) y + (r - rb);
    }
};

struct StageGeom { int r_hi, nb, nchunks, steps; };
template <int T> __device__ __forceinline__ StageGeom plan_geom(int k, int r_hi) {
    StageGeom g;
    g.r_hi = r_hi;
    g.nb = (T == MI355Q_TYPE_Q8_0 || T == MI355Q_TYPE_Q4_0 || T == MI355Q_TYPE_IQ4_NL) ? k >> 5 : k >> 8;
    g.nchunks = row_chunks(T, k); g.steps = (g.nchunks + 63) >> 6;
    return g;
}

// the row a wave takes after local row r: rows are dealt round robin; a PAIRED stage deals PAIRS (rows 2q, 2q+1 back to back, then 16 pairs on)
__device__ __forceinline__ int plan_next_row(int r, int paired) { return r + (paired ? ((r & 1) ? 2 * GEMV_WAVES - 1 : 1) : GEMV_WAVES); }

template <int T>
__device__ __forceinline__ void plan_issue(Chunk & slot, PlanCursor & ld, const StageW & st, const StageGeom & g, int lane) {
    if (ld.gr < g.r_hi) {                                     // wave-uniform
        if (64 * ld.s + lane < g.nchunks) chunk_load<T>(slot, ld.row, g.nb, ld.s, lane);
        if (++ld.s == g.steps) { ld.s = 0; ld.gr = plan_next_row(ld.gr, st.paired); if (ld.gr < g.r_hi) ld.row = st.row_ptr(ld.gr); }
    }
}

// request slots [from, to) of the ring for stage st (its loader cursor continues where it stands)
template <int T, int D>
__device__ __forceinline__ void plan_fill(Chunk (&ring)[D], int from, int to, PlanCursor & ld, const StageW & st, const StageGeom & g, int lane) {
#pragma unroll
    for (int d = 0; d < D; ++d) if (d >= from && d < to) plan_issue<T>(ring[d], ld, st, g, lane);
}

__device__ __forceinline__ void publish(Granule * gp, float v, unsigned tag) {
    __hip_atomic_store(gp, ((Granule) tag << 32) | (Granule) __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ float unary_f(int uop, float x);

// consume this wave's rows of the stage; slot d holds item d, d+D, ... ; refills keep D items in flight
template <int T, int D>
__device__ __forceinline__ void plan_run(Chunk (&ring)[D], PlanCursor & ld, const StageW & st, const StageGeom & g, int r_lo, int wave, int lane,
                                         const ActView * av, unsigned tag, bool plain, int pair_p0, int pair_unary) {
    // the consumers' lane-invariant state (LDS offsets, shifts) is derived from an opaque copy of the lane id HERE, so
    // that it cannot be computed (and kept live, and spilled) before the prologue
    int lane_c = lane; asm volatile("" : "+v"(lane_c));
    int cs_gr = r_lo + (st.paired ? 2 * wave : wave), cs_s = 0;
    float acc[1] = { 0.0f };
    float first_of_pair = 0.0f;                              // PAIRED: the dot product of the pair's matrix-0 row, until the matrix-1 row is done
    while (cs_gr < g.r_hi) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (cs_gr < g.r_hi) {                             // wave-uniform
                if (64 * cs_s + lane_c < g.nchunks) Consume<T, 1>::run(ring[d], cs_s, lane_c, av, acc);
                if (++cs_s == g.steps) {                      // row finished: reduce, publish (one granule), next row
                    const float t = wave_sum(acc[0]);
                    if (st.paired) {                           // (wave-uniform)
                        if (!(cs_gr & 1)) first_of_pair = t;
                        else if (lane_c == 0) {
                            const float r = __fmul_rn(unary_f(pair_unary, first_of_pair), t);
                            publish(st.g + pair_p0 + (cs_gr >> 1), r, tag);
                            if (plain) st.y0[pair_p0 + (cs_gr >> 1)] = r;
                        }
                    } else if (lane_c == 0) { publish(st.g + cs_gr, t, tag); if (plain) *st.y_ptr(cs_gr) = t; }
                    acc[0] = 0.0f; cs_s = 0; cs_gr = plan_next_row(cs_gr, st.paired);
                }
                plan_issue<T>(ring[d], ld, st, g, lane);      // refill the slot just consumed
            }
        }
    }
}

// LDS visibility + workgroup barrier WITHOUT draining vmcnt: __syncthreads() fences every address space and so waits
// for the weight loads in flight; these fences name the LDS only (lgkmcnt), and -- unlike a bare s_barrier, which is
// IntrNoMem for the compiler -- they also keep the LDS accesses on their side of the barrier at compile time.
__device__ __forceinline__ void plan_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- operand gather ------------------------------------------------------------------------------------------------
// A CHUNK is 128 consecutive elements of a vector; lane l owns elements 128c + 2l, +1.  Granules: one 16-byte agent-scope
// (sc1) buffer load per lane = {v0, tag0, v1, tag1}; plain vectors: one 8-byte load.  Lanes beyond the vector read zeros
// (the buffer resource bounds the access) and are excluded from the tag check.
typedef unsigned int plan_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int plan_u2 __attribute__((ext_vector_type(2)));
struct SrcView { __amdgpu_buffer_rsrc_t rs; unsigned expect; int tagged; };
__device__ __forceinline__ SrcView src_view(const VecSrc & v, int first, int n, unsigned epoch) {
    SrcView s;
    s.tagged = v.gran != nullptr;
    s.expect = epoch + v.tag_off;
    if (s.tagged) s.rs = __builtin_amdgcn_make_buffer_rsrc((void *) (v.gran + first), 0, n * 8, 0x00020000);
    else          s.rs = __builtin_amdgcn_make_buffer_rsrc((void *) (v.plain + first), 0, n * 4, 0x00020000);
    return s;
}
// one attempt: returns the two values and whether they are valid (tags match)
__device__ __forceinline__ bool src_try(const SrcView & s, int chunk, int lane, float & a0, float & a1) {
    if (s.tagged) {
        const plan_u4 v = __builtin_amdgcn_raw_buffer_load_b128(s.rs, chunk * 1024 + 16 * lane, 0, 16 /* sc1 */);
        a0 = __uint_as_float(v.x); a1 = __uint_as_float(v.z);
        return v.y == s.expect && v.w == s.expect;
    }
    const plan_u2 v = __builtin_amdgcn_raw_buffer_load_b64(s.rs, chunk * 512 + 8 * lane, 0, 0);
    a0 = __uint_as_float(v.x); a1 = __uint_as_float(v.y);
    return true;
}

struct PollCtx { unsigned * sync; unsigned long long timeout, t0; };
// false = give up (timeout or the plan's abort flag is up)
__device__ __forceinline__ bool poll_backoff(const PollCtx & pc, unsigned & spins, int lane) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 63u) == 0u) {
        unsigned ab = 0;
        if (lane == 0) ab = __hip_atomic_load(pc.sync + PLAN_SYNC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ab = (unsigned) __builtin_amdgcn_readfirstlane((int) ab);
        if (ab != 0u) return false;
        if (__builtin_amdgcn_s_memrealtime() - pc.t0 > pc.timeout) {
            if (lane == 0) __hip_atomic_store(pc.sync + PLAN_SYNC_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

// f64 sum over the 64 lanes on DPP (no LDS crossbar), result uniform
__device__ __forceinline__ double dpp_d(double v, int which) {
    const long long b = __double_as_longlong(v);
    int lo = (int) b, hi = (int) (b >> 32);
    switch (which) {
    case 0: lo = dpp_i<0xB1>(lo);  hi = dpp_i<0xB1>(hi);  break;
    case 1: lo = dpp_i<0x4E>(lo);  hi = dpp_i<0x4E>(hi);  break;
    case 2: lo = dpp_i<0x141>(lo); hi = dpp_i<0x141>(hi); break;
    default: lo = dpp_i<0x140>(lo); hi = dpp_i<0x140>(hi); break;
    }
    return __longlong_as_double(((long long) hi << 32) | (unsigned int) lo);
}
__device__ __forceinline__ double readlane_d(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int) b, l), hi = __builtin_amdgcn_readlane((int) (b >> 32), l);
    return __longlong_as_double(((long long) hi << 32) | (unsigned int) lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_d(v, 0); v += dpp_d(v, 1); v += dpp_d(v, 2); v += dpp_d(v, 3);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
__device__ __forceinline__ float wave_max_f(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}

__device__ __forceinline__ float unary_f(int uop, float x) {          // ops_glue.hip k_unary_mul, same expressions
    if (uop == MI355Q_UNARY_SILU) return __fdiv_rn(x, 1.0f + expf(-x));
    if (uop == MI355Q_UNARY_RELU) return x > 0.0f ? x : 0.0f;
    return __fdiv_rn(1.0f, 1.0f + expf(-x));
}

// A pointer / value that is wave-uniform by construction but arrives in VGPRs (arguments of a non-inlined function): moved to SGPRs so that
// everything read through it becomes scalar loads instead of per-lane global loads with their ~1 us dependent latencies.
template <typename P> __device__ __forceinline__ P uniform_ptr(P p) {
    const unsigned long long v = (unsigned long long) (uintptr_t) p;
    const unsigned lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) v), hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (v >> 32));
    return (P) (uintptr_t) (((unsigned long long) hi << 32) | lo);
}

struct StageCtx {
    uint8_t * lds; float * stg; int * ctl; double * part; unsigned * sync; unsigned long long timeout;
    unsigned grid, epoch; int even, stage, image;     // image: bytes of the activation image at the start of the LDS allocation
    const uint8_t * next_desc, * next_attn, * next_desc2; unsigned pre_lds;   // the next stage's descriptors (global; next_desc2: the one behind it -- the second member when the next stage heads a group) and a 1-KiB LDS scratch area their lines are DMA-ed into
};

// Warm this XCD's L2 with the NEXT stage's descriptor while the current stage runs: descriptors are cold after every launch boundary and a
// scalar load that misses to HBM costs ~1-2 us at the head of each of the ~200 stages of a token.  16 bytes per lane, global -> LDS with no VGPR
// destination (the data is never read from there); untracked by the compiler's vmcnt bookkeeping, which can only make its waits longer.
__device__ __forceinline__ void plan_prefetch_desc(const StageCtx & c, int wave, int lane) {
    static_assert(sizeof(PlanStage) <= 32 * 16 && sizeof(AttnStage) <= 32 * 16, "descriptor prefetch covers 512 bytes each");
    if (wave == GEMV_WAVES - 1 && c.next_desc != nullptr) {
        const bool second = lane >= 32;                        // lanes 0..31: the stage descriptor; lanes 32..63: its attention descriptor, if any, else the descriptor behind it
        const bool attn2 = c.next_attn != nullptr;
        const uint8_t * base = second ? (attn2 ? c.next_attn : c.next_desc2) : c.next_desc;
        const int n16 = second && attn2 ? (int) ((sizeof(AttnStage) + 15) / 16) : (int) ((sizeof(PlanStage) + 15) / 16);
        if (base != nullptr && (lane & 31) < n16) {
            const uint8_t * g = base + 16 * (lane & 31);
            const unsigned dst = (unsigned) __builtin_amdgcn_readfirstlane((int) c.pre_lds);     // (wave-uniform by construction; the asm needs an SGPR)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(dst) : "memory");
        }
    }
}
enum { CTL_OK = 0 };

// Gather the stage's activation vector t[k] (f32) into the LDS staging area: PLAIN t = x0, NORM t = x0 (+ x1), UNARY_MUL
// t = unary(x0) * x1.  Chunks are dealt round-robin to the waves, two in flight per wave.  NORM: returns this wave's partial
// sum of squares in `ssq`; the workgroup `sum_wg` also publishes t (the next residual's operand).  false = poll gave up.
__device__ __forceinline__ bool plan_gather(StageC st, const StageCtx & c, int k, int wave, int lane, double & ssq) {
    const int x_kind = st->x_kind, uop = st->x_unary & 0xFF;
    const bool two = st->x1.plain != nullptr || st->x1.gran != nullptr;
    VecSrc v0, v1;
    v0.plain = st->x0.plain; v0.gran = st->x0.gran; v0.tag_off = st->x0.tag_off; v0.pad = 0;
    v1.plain = st->x1.plain; v1.gran = st->x1.gran; v1.tag_off = st->x1.tag_off; v1.pad = 0;
    const SrcView s0 = src_view(v0, 0, k, c.epoch);
    const SrcView s1 = src_view(two ? v1 : v0, 0, k, c.epoch);
    const bool pub = (st->flags & PLAN_F_SUM) && (int) blockIdx.x == st->pub_wg;
    const unsigned tag = c.epoch + st->tag_off;
    const int chunks = (k + 127) >> 7;
    PollCtx pc; pc.sync = c.sync; pc.timeout = c.timeout; pc.t0 = __builtin_amdgcn_s_memrealtime();
    unsigned spins = 0;
    ssq = 0.0;
    // four 16-byte loads per lane are in flight per poll: two chunks of both operands, or four chunks of a single operand
    // (a chunk past the end reads zeros through the bounded buffer resource and is not checked)
    auto emit = [&](int cc, float p0, float p1, float q0, float q1) {
        const int e = 128 * cc + 2 * lane;
        if (e < k) {                                           // (k is even)
            float t0, t1;
            if (x_kind == MI355Q_X_UNARY_MUL) { t0 = __fmul_rn(unary_f(uop, p0), q0); t1 = __fmul_rn(unary_f(uop, p1), q1); }
            else if (two)                     { t0 = __fadd_rn(p0, q0); t1 = __fadd_rn(p1, q1); }
            else                              { t0 = p0; t1 = p1; }
            if (x_kind == MI355Q_X_NORM) {
                ssq += (double) __fmul_rn(t0, t0); ssq += (double) __fmul_rn(t1, t1);      // (ggml_float)(x*x): the square is rounded to f32 first
                if (pub) {
                    publish(st->sum_gran + e, t0, tag); publish(st->sum_gran + e + 1, t1, tag);
                    if (st->flags & PLAN_F_SUM_PLAIN) { st->sum_plain[e] = t0; st->sum_plain[e + 1] = t1; }
                }
            }
            *(float2 *) (c.stg + e) = make_float2(t0, t1);
        }
    };
    const int per_it = two ? 2 : 4;
#pragma unroll 1
    for (int ch = wave; ch < chunks; ch += per_it * GEMV_WAVES) {
        float a0, a1, b0, b1, e0, e1, f0, f1;
        const int c1 = ch + GEMV_WAVES, c2 = ch + 2 * GEMV_WAVES, c3 = ch + 3 * GEMV_WAVES;
        for (;;) {
            bool ok;
            if (two) {
                ok =       src_try(s0, ch, lane, a0, a1) || 128 * ch + 2 * lane >= k;
                ok = (src_try(s1, ch, lane, b0, b1) || 128 * ch + 2 * lane >= k) && ok;
                ok = (src_try(s0, c1, lane, e0, e1) || 128 * c1 + 2 * lane >= k) && ok;
                ok = (src_try(s1, c1, lane, f0, f1) || 128 * c1 + 2 * lane >= k) && ok;
            } else {
                ok =       src_try(s0, ch, lane, a0, a1) || 128 * ch + 2 * lane >= k;
                ok = (src_try(s0, c1, lane, b0, b1) || 128 * c1 + 2 * lane >= k) && ok;
                ok = (src_try(s0, c2, lane, e0, e1) || 128 * c2 + 2 * lane >= k) && ok;
                ok = (src_try(s0, c3, lane, f0, f1) || 128 * c3 + 2 * lane >= k) && ok;
            }
            if (__ballot(!ok) == 0ull) break;
            if (!poll_backoff(pc, spins, lane)) return false;
        }
        if (two) { emit(ch, a0, a1, b0, b1); emit(c1, e0, e1, f0, f1); }
        else     { emit(ch, a0, a1, 0.f, 0.f); emit(c1, b0, b1, 0.f, 0.f); emit(c2, e0, e1, 0.f, 0.f); emit(c3, f0, f1, 0.f, 0.f); }
    }
    return true;
}

template <int FAM>
__device__ __forceinline__ void plan_quantize_span(const float4 v, int span, uint8_t * lds, int k, bool even, int lane) {
    if constexpr (FAM == FAM_Q8K) quantize_span_to_lds<FAM_Q8K, false>(v, span, lds, k, lane);
    else if (even)                quantize_span_to_lds<FAM_Q80, true>(v, span, lds, k, lane);
    else                          quantize_span_to_lds<FAM_Q80, false>(v, span, lds, k, lane);
}

// A stage whose activation vector is ONE vector taken as it is (X_PLAIN, no second operand: wo, ffn_down): a wave fetches whole 256-element
// spans -- lane l the four elements 4l .. 4l+3, i.e. two 16-byte granule loads -- and quantizes each span into the LDS image straight from its
// registers.  No staging copy and no workgroup barrier between gathering and quantizing (plan_gather needs one: its chunks are dealt 16 apart,
// so a 256-block's two halves sit in different waves).  Two spans (four loads) are in flight per poll.  false = the poll gave up.
template <int FAM>
__device__ __forceinline__ bool plan_gather_direct(StageC st, const StageCtx & c, int k, int wave, int lane) {
    VecSrc v0;
    v0.plain = st->x0.plain; v0.gran = st->x0.gran; v0.tag_off = st->x0.tag_off; v0.pad = 0;
    const SrcView s0 = src_view(v0, 0, k, c.epoch);
    const int spans = k >> 8;                                  // (k % 256 == 0, checked at plan creation)
    PollCtx pc; pc.sync = c.sync; pc.timeout = c.timeout; pc.t0 = __builtin_amdgcn_s_memrealtime();
    unsigned spins = 0;
#pragma unroll 1
    for (int sp = wave; sp < spans; sp += 2 * GEMV_WAVES) {
        const int sp2 = sp + GEMV_WAVES;
        float4 va, vb;
        for (;;) {
            bool ok = true;
            if (s0.tagged) {
                const plan_u4 a0 = __builtin_amdgcn_raw_buffer_load_b128(s0.rs, sp * 2048 + 32 * lane, 0, 16 /* sc1 */);
                const plan_u4 a1 = __builtin_amdgcn_raw_buffer_load_b128(s0.rs, sp * 2048 + 32 * lane + 16, 0, 16);
                const plan_u4 b0 = __builtin_amdgcn_raw_buffer_load_b128(s0.rs, sp2 * 2048 + 32 * lane, 0, 16);      // (past the vector: zeros, not checked)
                const plan_u4 b1 = __builtin_amdgcn_raw_buffer_load_b128(s0.rs, sp2 * 2048 + 32 * lane + 16, 0, 16);
                va = make_float4(__uint_as_float(a0.x), __uint_as_float(a0.z), __uint_as_float(a1.x), __uint_as_float(a1.z));
                vb = make_float4(__uint_as_float(b0.x), __uint_as_float(b0.z), __uint_as_float(b1.x), __uint_as_float(b1.z));
                ok = a0.y == s0.expect && a0.w == s0.expect && a1.y == s0.expect && a1.w == s0.expect;
                if (sp2 < spans) ok = ok && b0.y == s0.expect && b0.w == s0.expect && b1.y == s0.expect && b1.w == s0.expect;
            } else {
                const plan_u4 a = __builtin_amdgcn_raw_buffer_load_b128(s0.rs, sp * 1024 + 16 * lane, 0, 0);
                const plan_u4 b = __builtin_amdgcn_raw_buffer_load_b128(s0.rs, sp2 * 1024 + 16 * lane, 0, 0);
                va = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
                vb = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w));
            }
            if (__ballot(!ok) == 0ull) break;
            if (!poll_backoff(pc, spins, lane)) return false;
        }
        plan_quantize_span<FAM>(va, sp, c.lds, k, c.even != 0, lane);
        if (sp2 < spans) plan_quantize_span<FAM>(vb, sp2, c.lds, k, c.even != 0, lane);
    }
    return true;
}

// One GEMV stage, start to finish, for weight type T:
//   prime (weights in flight) -> [gather x (polls its producers) -> glue -> quantize x -> LDS] -> stream rows, publishing every y
template <int T>
static __device__ __forceinline__ bool plan_stage(StageC st, const StageCtx & c) {
    // an opaque copy of the lane id per stage: everything derived from it is recomputed here (a few VALU ops) instead
    // of being hoisted out of the stage loop for BOTH types' loaders, quantizers and pollers and kept live (and spilled)
    int lane = lane_id(); asm volatile("" : "+v"(lane));
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    PLAN_STAMP(0);
    const int k = st->k, flags = st->flags;
    int r_lo = ((int) blockIdx.x - st->wg_base) * st->rows_per_wg, r_hi = r_lo + st->rows_per_wg;
    if (r_hi > st->total_rows) r_hi = st->total_rows;
    StageW sw;
    int pair_p0 = 0, pair_np = 0;
    if (flags & PLAN_F_PAIRED) {                              // rows_per_wg = pairs per workgroup, total_rows = m: local rows [0, 2 np)
        pair_p0 = r_lo; pair_np = max(0, r_hi - r_lo);
        sw.load_paired(st, pair_p0, pair_np);
        r_lo = 0; r_hi = 2 * pair_np;
    } else sw.load(st);
    const StageGeom g = plan_geom<T>(k, r_hi);

    constexpr int PLAN_D = plan_depth(T);
    Chunk ring[PLAN_D];
    PlanCursor ld;
    ld.gr = r_lo + (sw.paired ? 2 * wave : wave); ld.s = 0; ld.row = nullptr;
    if (ld.gr < r_hi) ld.row = sw.row_ptr(ld.gr);
#ifdef MI355Q_STAMPS
    { unsigned long long probe = (unsigned long long) (uintptr_t) ld.row + (unsigned) k; asm volatile("" :: "s"(probe)); }   // (forces the descriptor's scalar loads to have returned)
    PLAN_STAMP(5);
#endif
    plan_fill<T, PLAN_D>(ring, 0, st->prime, ld, sw, g, lane);                 // weights start flowing before anything else
    plan_prefetch_desc(c, wave, lane);
    PLAN_STAMP(1);

    if (flags & PLAN_F_NEW_X) {
        constexpr int FAM = (T == MI355Q_TYPE_Q8_0 || T == MI355Q_TYPE_Q4_0 || T == MI355Q_TYPE_IQ4_NL) ? FAM_Q80 : FAM_Q8K;
        plan_lds_barrier();                                   // all waves are done with the previous LDS image and staging area
        if (flags & PLAN_F_DIRECT) {                          // one plain vector: gathered and quantized span by span, one barrier
            const bool okd = plan_gather_direct<FAM>(st, c, k, wave, lane);
            if (!okd && lane == 0) c.ctl[CTL_OK] = 0;
            PLAN_STAMP(2);
            plan_lds_barrier();
            if (!c.ctl[CTL_OK]) return false;
        } else {
        const int x_kind = st->x_kind;
        const float * nw = x_kind == MI355Q_X_NORM ? st->norm_w : nullptr;
        float4 w_first = make_float4(1.f, 1.f, 1.f, 1.f);      // the norm weights of this wave's first span: fetched before the producers are polled
        if (nw && wave * 256 + 4 * lane < k) w_first = *(const float4 *) (nw + wave * 256 + 4 * lane);
        double ssq;
        const bool ok = plan_gather(st, c, k, wave, lane, ssq);
        if (!ok && lane == 0) c.ctl[CTL_OK] = 0;
        PLAN_STAMP(2);
        if (x_kind == MI355Q_X_NORM) {
            ssq = wave_sum_f64(ssq);
            if (lane == 0) c.part[wave] = ssq;
        }
        plan_lds_barrier();
        if (!c.ctl[CTL_OK]) return false;
        PLAN_STAMP(6);
        float scale = 1.0f;
        if (x_kind == MI355Q_X_NORM) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < GEMV_WAVES; i += 4) s += (c.part[i] + c.part[i + 1]) + (c.part[i + 2] + c.part[i + 3]);
            const float mean = (float) (s / (double) k);
            const float root = (float) sqrt((double) __fadd_rn(mean, st->eps));      // both roundings of the CPU (ops_glue.hip k_add_rms_norm_mul)
            scale = (float) (1.0 / (double) root);
        }
        const int spans = (k + 255) >> 8;
        float * x_out = st->x_out;
        const bool pub_x = x_out != nullptr && (int) blockIdx.x == st->pub_wg;
#pragma unroll 1
        for (int span = wave; span < spans; span += GEMV_WAVES) {
            const int e = span * 256 + 4 * lane;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < k) {
                v = *(const float4 *) (c.stg + e);
                if (x_kind == MI355Q_X_NORM) {
                    v.x = __fmul_rn(v.x, scale); v.y = __fmul_rn(v.y, scale); v.z = __fmul_rn(v.z, scale); v.w = __fmul_rn(v.w, scale);
                    if (nw) { const float4 ww = span == wave ? w_first : *(const float4 *) (nw + e); v.x = __fmul_rn(v.x, ww.x); v.y = __fmul_rn(v.y, ww.y); v.z = __fmul_rn(v.z, ww.z); v.w = __fmul_rn(v.w, ww.w); }
                }
                if (pub_x) *(float4 *) (x_out + e) = v;         // the formed vector itself is a graph value somebody else reads (result_norm / embeddings, a LoRA branch)
            }
            plan_quantize_span<FAM>(v, span, c.lds, k, c.even != 0, lane);
        }
        PLAN_STAMP(7);
        plan_lds_barrier();
        }
    }
    plan_fill<T, PLAN_D>(ring, st->prime, PLAN_D, ld, sw, g, lane);                // top the ring up (a no-op when prime == depth)
    sw.load_out(st);
    PLAN_STAMP(3);
    ActView av[1];
    av[0].base = c.lds; av[0].k = k;
    // (a PAIRED stage publishes unary(row of matrix 0) * (row of matrix 1) per pair from the wave that streamed both: the output unary is kept in
    // the high byte of x_unary)
    plan_run<T, PLAN_D>(ring, ld, sw, g, r_lo, wave, lane, av, c.epoch + st->tag_off, (flags & PLAN_F_PLAIN_Y) != 0, pair_p0, st->x_unary >> 8);
    PLAN_STAMP(4);
    return true;
}

// ---- attention of one token ------------------------------------------------------------------------------------------
// Workgroup b = (head h, KV split sp).  LDS (inside the staging area): sq / sk / sv f32 [hd], kh / vh f16 [hd] (this token's
// cache row, rounded as stored), sc f32 [per] (scores, then probabilities), red f32 [16][hd] (P V partials), maxs / sums [16].
typedef const __attribute__((address_space(4))) AttnStage * AttnC;          // the descriptor is read with scalar loads

static __device__ __noinline__ bool plan_attn(AttnC a_in, const StageCtx & c_in, unsigned tag_in) {
    const AttnC a = (AttnC) (uintptr_t) uniform_ptr((const void *) (uintptr_t) a_in);      // (see uniform_ptr)
    // LDS pointers are re-derived from the kernel's LDS symbol: taken from the caller's struct they would be generic pointers (flat loads)
    extern __shared__ __attribute__((aligned(16))) uint8_t plan_lds_attn[];
    StageCtx c;
    c.image = __builtin_amdgcn_readfirstlane(c_in.image);
    c.lds = plan_lds_attn; c.ctl = (int *) (plan_lds_attn + c.image); c.part = (double *) (plan_lds_attn + c.image + 64);
    c.stg = (float *) (plan_lds_attn + c.image + 64 + 8 * GEMV_WAVES + 1024);
    c.sync = uniform_ptr(c_in.sync); c.timeout = c_in.timeout;
    c.grid = (unsigned) __builtin_amdgcn_readfirstlane((int) c_in.grid); c.epoch = (unsigned) __builtin_amdgcn_readfirstlane((int) c_in.epoch);
    c.even = c_in.even; c.stage = __builtin_amdgcn_readfirstlane(c_in.stage);
    c.next_desc = uniform_ptr(c_in.next_desc); c.next_attn = uniform_ptr(c_in.next_attn); c.next_desc2 = uniform_ptr(c_in.next_desc2); c.pre_lds = c_in.pre_lds;
    const unsigned tag = (unsigned) __builtin_amdgcn_readfirstlane((int) tag_in);
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int tid = (int) threadIdx.x;
    const int hd = a->hd, n_split = a->n_split;
    if ((int) blockIdx.x >= a->n_head * n_split) return true;                    // (uniform: the whole workgroup has nothing to do)
    const int h = (int) blockIdx.x / n_split, sp = (int) blockIdx.x % n_split, gq = a->n_head / a->n_head_kv, g = h / gq;
    float * sq = c.stg, * sk = sq + hd, * sv = sk + hd;
    __half * kh = (__half *) (sv + hd), * vh = kh + hd;
    float * maxs = (float *) (vh + hd), * sums = maxs + GEMV_WAVES;
    float * red = sums + GEMV_WAVES;                                          // [16][hd]
    float * sc = red + GEMV_WAVES * hd;                                        // [per]
    // loads that depend on nothing this launch computes are issued first: the token's position, its cache slots, the window, the mask
    const int32_t * pos_p = a->pos, * nkv_p = a->n_kv_dev;
    char * const * kdst_p = a->k_dst, * const * vdst_p = a->v_dst;
    const int pos = pos_p[0];
    char * const kdst = *kdst_p, * const vdst = *vdst_p;
    int n_kv = a->n_kv;                                                        // the window of THIS run (the plan is sized for a->n_kv)
    if (nkv_p) n_kv = min(n_kv, max(1, nkv_p[0]));
    const int per = (n_kv + n_split - 1) / n_split;
    const int j0 = sp * per, j1 = min(n_kv, j0 + per), cnt = max(0, j1 - j0);
    const char * maskp = a->mask; const int mask_f16 = a->mask_f16;
    plan_lds_barrier();                                                        // the staging area is free (previous stage's quantizer is done)
    plan_prefetch_desc(c, wave, lane);
    PLAN_STAMP(0);
    for (int jj = tid; jj < cnt; jj += GEMV_THREADS)                           // the additive mask of this split's positions (0 without a mask)
        sc[jj] = maskp ? (mask_f16 ? __half2float(((const __half *) maskp)[j0 + jj]) : ((const float *) maskp)[j0 + jj]) : 0.0f;

    // 1. q head h, k / v head g  (hd <= 256: at most two 128-element chunks each)
    {
        PollCtx pc; pc.sync = c.sync; pc.timeout = c.timeout; pc.t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        const int nch = (hd + 127) >> 7;
        bool ok_all = true;
        for (int item = wave; item < 3 * nch; item += GEMV_WAVES) {
            const int which = item / nch, ch = item % nch;
            VecSrc vs;
            if (which == 0)      { vs.plain = a->q.plain; vs.gran = a->q.gran; vs.tag_off = a->q.tag_off; }
            else if (which == 1) { vs.plain = a->k.plain; vs.gran = a->k.gran; vs.tag_off = a->k.tag_off; }
            else                 { vs.plain = a->v.plain; vs.gran = a->v.gran; vs.tag_off = a->v.tag_off; }
            vs.pad = 0;
            const SrcView s = src_view(vs, (which == 0 ? h : g) * hd, hd, c.epoch);
            float * dst = which == 0 ? sq : which == 1 ? sk : sv;
            float v0, v1;
            for (;;) {
                const bool ok = src_try(s, ch, lane, v0, v1) || 128 * ch + 2 * lane >= hd;
                if (__ballot(!ok) == 0ull) break;
                if (!poll_backoff(pc, spins, lane)) { ok_all = false; break; }
            }
            if (!ok_all) break;
            const int e = 128 * ch + 2 * lane;
            if (e < hd) { dst[e] = v0; dst[e + 1] = v1; }
        }
        if (!ok_all && lane == 0) c.ctl[CTL_OK] = 0;
    }
    plan_lds_barrier();
    if (!c.ctl[CTL_OK]) return false;
    PLAN_STAMP(1);

    // 2. rope (ops_glue.hip k_rope / ggml-cpu/ops.cpp:5088-5270); q is then rounded to f16 as the CPU's f16 vec_dot does with src1,
    //    k and v to f16 as the cache stores them.  One rotated pair per thread: pairs of q first, then of k.
    const int half = hd >> 1;
    if (tid < 2 * half) {
        float * x = tid < half ? sq : sk;
        const int ip = tid < half ? tid : tid - half;
        const int i0 = 2 * ip;
        const int n_dims = a->n_dims, neox = a->neox;
        float r0, r1; int e0, e1;
        if (i0 < n_dims) {
            const float * ffp = a->freq_factors;
            const float ff = ffp ? ffp[ip] : 1.0f;
            const float tscale = a->theta_scale;
            float th = (float) pos;
            for (int j = 0; j < ip; ++j) th = __fmul_rn(th, tscale);             // repeated f32 multiplication, as ggml_rope_cache_init does
            const float theta_extrap = __fdiv_rn(th, ff);
            const float theta_interp = a->freq_scale * theta_extrap;
            float theta = theta_interp, mscale = a->attn_factor;
            if (a->ext_factor != 0.0f) {
                const float y = ((float) ip - a->corr0) / fmaxf(0.001f, a->corr1 - a->corr0);
                const float ramp_mix = (1.0f - fminf(1.0f, fmaxf(0.0f, y))) * a->ext_factor;
                theta = theta_interp * (1.0f - ramp_mix) + theta_extrap * ramp_mix;
                mscale *= 1.0f + 0.1f * logf(1.0f / a->freq_scale);
            }
            const float cs = cosf(theta) * mscale, sn = sinf(theta) * mscale;
            e0 = neox ? ip : i0; e1 = neox ? ip + n_dims / 2 : i0 + 1;
            const float x0 = x[e0], x1 = x[e1];
            r0 = x0 * cs - x1 * sn; r1 = x0 * sn + x1 * cs;
        } else { e0 = i0; e1 = i0 + 1; r0 = x[e0]; r1 = x[e1]; }
        // (every pair reads and writes only its own two elements: in place is safe)
        if (tid < half) { x[e0] = __half2float(__float2half_rn(r0)); x[e1] = __half2float(__float2half_rn(r1)); }
        else            { kh[e0] = __float2half_rn(r0); kh[e1] = __float2half_rn(r1); }
    } else if (tid < 2 * half + hd) {
        const int d = tid - 2 * half;
        vh[d] = __float2half_rn(sv[d]);
    }
    plan_lds_barrier();
    if (sp == 0 && h % gq == 0 && tid < hd) {                                   // this token's cache row: one workgroup per kv head stores it
        ((__half *) kdst)[g * hd + tid] = kh[tid];
        *(__half *) (vdst + (int64_t) (g * hd + tid) * a->v_dst_nb) = vh[tid];
    }
    const char * kcache = a->k_cache;
    const int64_t k_nb_pos = a->k_nb_pos;
    const int slot = (int) ((kdst - kcache) / k_nb_pos);                       // the position whose row is being stored right now: read from LDS
    PLAN_STAMP(2);

    // 3. scores of this split's positions: one wave per position (8 in flight), a lane owns dims (2l, 2l+1) [+128]
    const char * kbase = kcache + (int64_t) g * a->k_nb_head;
    const float scale = a->scale;
    const float q0 = 2 * lane < hd ? sq[2 * lane] : 0.f, q1 = 2 * lane < hd ? sq[2 * lane + 1] : 0.f;
    const float q2 = 2 * lane + 128 < hd ? sq[2 * lane + 128] : 0.f, q3 = 2 * lane + 128 < hd ? sq[2 * lane + 129] : 0.f;
    constexpr int SB = 8;
#pragma unroll 1
    for (int jb = wave; jb < cnt; jb += SB * GEMV_WAVES) {
        __half2 kv[SB], kw[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int j = j0 + jb + u * GEMV_WAVES;
            kv[u] = __half2(); kw[u] = __half2();
            if (jb + u * GEMV_WAVES < cnt) {
                if (j == slot) { if (2 * lane < hd) kv[u] = *(const __half2 *) (kh + 2 * lane); if (2 * lane + 128 < hd) kw[u] = *(const __half2 *) (kh + 2 * lane + 128); }
                else {
                    const char * row = kbase + (int64_t) j * k_nb_pos;
                    if (2 * lane < hd) kv[u] = *(const __half2 *) (row + 4 * lane);
                    if (2 * lane + 128 < hd) kw[u] = *(const __half2 *) (row + 4 * lane + 256);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            if (jb + u * GEMV_WAVES < cnt) {
                const float2 f = __half22float2(kv[u]), f2 = __half22float2(kw[u]);
                float d = q0 * f.x + q1 * f.y + q2 * f2.x + q3 * f2.y;
                d = wave_sum(d);
                // (a fully masked position stays -inf whatever its cache row holds: never-written rows may be anything, 0 * NaN included)
                if (lane == 0) { const float m = sc[jb + u * GEMV_WAVES]; sc[jb + u * GEMV_WAVES] = m == -INFINITY ? -INFINITY : __fadd_rn(__fmul_rn(d, scale), m); }
            }
        }
    }
    plan_lds_barrier();
    PLAN_STAMP(3);
    // 4. local softmax statistics.  A split holds at most a few hundred scores: ONE wave forms maximum, exponentials, sum and (single split) the
    //    probabilities with register reductions, the other 15 wait at one barrier -- the workgroup-wide form cost three barriers for the same numbers.
    if (wave == 0) {
        float mx0 = -INFINITY;
        for (int j = lane; j < cnt; j += 64) mx0 = fmaxf(mx0, sc[j]);
        mx0 = wave_max_f(mx0);
        float l0;
        if (a->p_f16) {
            // The non-flash graph of the reference (SOFT_MAX, then MUL_MAT(v, kq) whose f16 src0 makes the CPU round kq to f16): the whole window is in
            // this workgroup, so the probabilities are formed exactly as ggml_compute_forward_soft_max_f32 does -- sum of the exponentials in f64,
            // p = e * (float) (1 / sum) -- and rounded to f16 before they meet V.  (An f32 P.V is closer to the exact product, but the next matmul
            // re-quantizes its input and amplifies any 1e-4 difference from the CPU to ~1 % of the logits: DESIGN.md section 3b.)
            double ds = 0.0;
            for (int j = lane; j < cnt; j += 64) {
                const float e = mx0 == -INFINITY ? 0.0f : expf(__fsub_rn(sc[j], mx0));
                sc[j] = e; ds += (double) e;
            }
            ds = wave_sum_f64(ds);
            const float inv = (float) (1.0 / ds);
            for (int j = lane; j < cnt; j += 64) sc[j] = f16_round(__fmul_rn(sc[j], inv));      // (a lane re-reads only what it wrote)
            l0 = 1.0f;
        } else {
            float ls = 0.0f;
            for (int j = lane; j < cnt; j += 64) {
                const float p = mx0 == -INFINITY ? 0.0f : expf(__fsub_rn(sc[j], mx0));
                sc[j] = p; ls += p;
            }
            l0 = wave_sum(ls);
        }
        if (lane == 0) { maxs[0] = mx0; sums[0] = l0; }
    }
    plan_lds_barrier();
    const float mx = maxs[0], l = sums[0];
    PLAN_STAMP(4);
    // 5. o[d] = sum_j p_j v[j][d]   (positions with p == 0 are skipped: masked cache rows may hold anything)
    const char * vbase = a->v_cache + (int64_t) g * a->v_nb_head;
    const int64_t v_nb_pos = a->v_nb_pos, v_nb_dim = a->v_nb_dim;
    if (v_nb_dim == 2) {
        // rows per position (the -fa layout): a wave takes positions jb = wave, wave+16, ...; a lane owns dims (2l, 2l+1) [+128]
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll 1
        for (int jb = wave; jb < cnt; jb += SB * GEMV_WAVES) {
            __half2 vv[SB], vw[SB]; float p[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int jj = jb + u * GEMV_WAVES, j = j0 + jj;
                vv[u] = __half2(); vw[u] = __half2(); p[u] = jj < cnt ? sc[jj] : 0.0f;
                if (p[u] != 0.0f) {
                    if (j == slot) { if (2 * lane < hd) vv[u] = *(const __half2 *) (vh + 2 * lane); if (2 * lane + 128 < hd) vw[u] = *(const __half2 *) (vh + 2 * lane + 128); }
                    else {
                        const char * row = vbase + (int64_t) j * v_nb_pos;
                        if (2 * lane < hd) vv[u] = *(const __half2 *) (row + 4 * lane);
                        if (2 * lane + 128 < hd) vw[u] = *(const __half2 *) (row + 4 * lane + 256);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const float2 f = __half22float2(vv[u]), f2 = __half22float2(vw[u]);
                o0 += p[u] * f.x; o1 += p[u] * f.y; o2 += p[u] * f2.x; o3 += p[u] * f2.y;
            }
        }
        if (2 * lane < hd) { red[wave * hd + 2 * lane] = o0; red[wave * hd + 2 * lane + 1] = o1; }
        if (2 * lane + 128 < hd) { red[wave * hd + 2 * lane + 128] = o2; red[wave * hd + 2 * lane + 129] = o3; }
        plan_lds_barrier();
        if (tid < hd) {
            float o = 0.0f;
#pragma unroll
            for (int i = 0; i < GEMV_WAVES; ++i) o += red[i * hd + tid];
            red[tid] = o;                                                      // (row 0 of red now holds o; every thread touches only its column)
        }
    } else {
        // transposed cache (positions contiguous per dim): a wave takes dims d = wave + 16 i; lanes run over the positions; the loads of
        // 8 dims are in flight together (one memory round trip per 64 positions instead of one per dim)
#pragma unroll 1
        for (int d0 = wave; d0 < hd; d0 += SB * GEMV_WAVES) {
            float o[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) o[u] = 0.0f;
#pragma unroll 1
            for (int jj = lane; jj < cnt; jj += 64) {
                const float p = sc[jj];
                const int j = j0 + jj;
                __half vv[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int d = d0 + u * GEMV_WAVES;
                    vv[u] = __half();
                    if (p != 0.0f && d < hd) vv[u] = j == slot ? vh[d] : *(const __half *) (vbase + (int64_t) d * v_nb_dim + (int64_t) j * v_nb_pos);
                }
#pragma unroll
                for (int u = 0; u < SB; ++u) o[u] += p * __half2float(vv[u]);
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int d = d0 + u * GEMV_WAVES;
                if (d < hd) { const float t = wave_sum(o[u]); if (lane == 0) red[d] = t; }     // (d wave-uniform)
            }
        }
    }
    plan_lds_barrier();
    // 6. publish: the normalized output when the head is not split, else (o, m, l) of this split for the merge stage
    if (n_split == 1) {
        if (tid < hd) {
            const float r = a->p_f16 ? red[tid] : __fdiv_rn(red[tid], l);
            publish(a->out_gran + h * hd + tid, r, tag);
            if (a->plain) a->out_plain[h * hd + tid] = r;
        }
    } else {
        Granule * part = a->part + (size_t) ((size_t) h * n_split + sp) * (hd + 2);
        if (tid < hd) publish(part + tid, red[tid], tag);
        else if (tid == hd) publish(part + hd, mx, tag);
        else if (tid == hd + 1) publish(part + hd + 1, l, tag);
    }
    PLAN_STAMP(5);
    return true;
}

// ---- attention of one token, K / V window in LDS ---------------------------------------------------------------------
// The same stage for a split whose window fits the LDS (AttnStage::kv_lds, decided at plan creation: 2 x per x head_dim f16 beside everything else;
// llama-bench's tg128 always does).  Built around what a wave pays per instruction when it runs the serial part of a stage (2.8-5 ns each, 40-95 ns
// per dependent LDS trip: profiles/round3_experiments.md) -- the round-2 stage above spent 14 us per layer on ~6000 issued instructions per wave:
//   0  in parallel: waves 0..2 poll q / k / v;  one wave forms the rope table of this position (cos, sin per pair; the repeated f32 multiplication
//      of ggml_rope_cache_init, 64 predicated steps) and the cache slot;  the others copy the cache window into LDS with LDS-DMA (16 bytes per lane)
//      and fetch the mask.  Nothing here depends on the launch's own results except the polls.
//   1  rope: one rotated pair per thread from the table; q -> f16 (the CPU's f16 vec_dot rounds src1), this token's k / v rows -> the cache and
//      their place in the LDS window (no "is this the current slot" case later)
//   2  scores: one POSITION per lane (no cross-lane reduction): 16-byte LDS reads of its K row, rotated by the lane so that the 64 rows do not
//      collide on the banks
//   3  softmax by wave 0 over at most 5 scores per lane held in registers
//   4  P.V from LDS; publish.
// FLASH graphs with the window in one workgroup (fa_seq) reproduce ggml_compute_forward_flash_attn_ext_f16 (ggml-cpu/ops.cpp:6686-6900): the CPU walks
// the positions in order with a running maximum and keeps V.P in an F16 accumulator -- rescaled (f32 multiply, rounded to f16) whenever the maximum
// grows, and v * expf(s - M) added with an f32 fma rounded to f16 per position.  That rounding chain is 1e-3 of the result (a layer's output differed
// by NMSE 5e-5 from the CPU's while an f32 accumulator is closer to the exact product); here a thread owns one dim and walks the positions in the
// CPU's order with the same two roundings, the rescale factors having been formed in parallel (prefix maximum) by step 3.
template <int CTRL> __device__ __forceinline__ float dpp_keep_f(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xF, 0xF, false));
}

// t[I] of ggml_vec_dot_f16's reduction tree for one K row against q (see plan_attn_lds step 2): the accumulators (j, l = I) and (j, l = I + 4), j = 0..3
template <int I>
__device__ __forceinline__ float attn_score_t(const __half * krow, const __half * qh, int nblk) {
    float alo[4] = { 0.f, 0.f, 0.f, 0.f }, ahi[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll 2
    for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4 kq = *(const uint4 *) (krow + 32 * blk + 8 * j), qq = *(const uint4 *) (qh + 32 * blk + 8 * j);
            const unsigned kw = I < 2 ? kq.x : kq.y, kw4 = I < 2 ? kq.z : kq.w, qw = I < 2 ? qq.x : qq.y, qw4 = I < 2 ? qq.z : qq.w;
            const float2 kf = __half22float2(*(const __half2 *) &kw), kf4 = __half22float2(*(const __half2 *) &kw4);
            const float2 qf = __half22float2(*(const __half2 *) &qw), qf4 = __half22float2(*(const __half2 *) &qw4);
            alo[j] = __builtin_fmaf((I & 1) ? kf.y : kf.x, (I & 1) ? qf.y : qf.x, alo[j]);
            ahi[j] = __builtin_fmaf((I & 1) ? kf4.y : kf4.x, (I & 1) ? qf4.y : qf4.x, ahi[j]);
        }
    }
    const float s_lo = __fadd_rn(__fadd_rn(alo[0], alo[2]), __fadd_rn(alo[1], alo[3]));
    const float s_hi = __fadd_rn(__fadd_rn(ahi[0], ahi[2]), __fadd_rn(ahi[1], ahi[3]));
    return __fadd_rn(s_lo, s_hi);
}

static __device__ __noinline__ bool plan_attn_lds(AttnC a_in, const StageCtx & c_in, unsigned tag_in) {
    const AttnC a = (AttnC) (uintptr_t) uniform_ptr((const void *) (uintptr_t) a_in);      // (see uniform_ptr)
    extern __shared__ __attribute__((aligned(16))) uint8_t plan_lds_attn2[];                 // (LDS pointers re-derived from the LDS symbol: see plan_attn)
    StageCtx c;
    c.image = __builtin_amdgcn_readfirstlane(c_in.image);
    c.lds = plan_lds_attn2; c.ctl = (int *) (plan_lds_attn2 + c.image); c.part = (double *) (plan_lds_attn2 + c.image + 64);
    c.stg = (float *) (plan_lds_attn2 + c.image + 64 + 8 * GEMV_WAVES + 1024);
    c.sync = uniform_ptr(c_in.sync); c.timeout = c_in.timeout;
    c.grid = (unsigned) __builtin_amdgcn_readfirstlane((int) c_in.grid); c.epoch = (unsigned) __builtin_amdgcn_readfirstlane((int) c_in.epoch);
    c.even = c_in.even; c.stage = __builtin_amdgcn_readfirstlane(c_in.stage);
    c.next_desc = uniform_ptr(c_in.next_desc); c.next_attn = uniform_ptr(c_in.next_attn); c.next_desc2 = uniform_ptr(c_in.next_desc2); c.pre_lds = c_in.pre_lds;
    const unsigned tag = (unsigned) __builtin_amdgcn_readfirstlane((int) tag_in);
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int tid = (int) threadIdx.x;
    const int hd = a->hd, n_split = a->n_split;
    if ((int) blockIdx.x >= a->n_head * n_split) return true;                    // (uniform: the whole workgroup has nothing to do)
    const int h = (int) blockIdx.x / n_split, sp = (int) blockIdx.x % n_split, gq = a->n_head / a->n_head_kv, g = h / gq;
    const int half = hd >> 1, per_pad = (a->per + 31) & ~31;
    // LDS (inside the staging area)
    __half * qh = (__half *) c.stg;                                            // roped q, f16 [hd]
    float * misc = (float *) (qh + hd);                                        // [16]: 0 maximum, 1 sum, 2 cache slot (int), 3 last position + 1 (int)
    float * sq = misc + 16, * sk = sq + hd, * sv = sk + hd;                    // q / k / v as polled
    float * cst = sv + hd;                                                     // cos [hd/2] | sin [hd/2]
    float * red = cst + hd;                                                    // [8][hd] P.V partials
    float * pbuf = sq;                                                         // step 2's partial dot products: the 12 hd floats from sq to the end of red (free between steps 1 and 4)
    float * sc = red + 8 * hd;                                                 // [per_pad] mask -> scores -> probabilities
    float * msv = sc + per_pad;                                                // [per_pad] fa_seq: the accumulator's rescale factor per position
    __half * kwin = (__half *) (msv + per_pad);                                // [cnt][hd + 8]: rows padded by 16 bytes, so that the 64 rows a wave reads side by side spread over the banks
    const int krs = hd + 8;                                                    //   (16-byte aligned: every area above is a multiple of 64 bytes; no integer round trip, which would make the pointer generic)
    __half * vwin = kwin + (size_t) per_pad * krs;                             // rows: [cnt][hd];  transposed cache: [hd][cnt]
    // loads that depend on nothing this launch computes
    const int32_t * pos_p = a->pos, * nkv_p = a->n_kv_dev;
    char * const * kdst_p = a->k_dst, * const * vdst_p = a->v_dst;
    const int pos = pos_p[0];
    char * const kdst = *kdst_p, * const vdst = *vdst_p;
    int n_kv = a->n_kv;                                                        // the window of THIS run (the plan is sized for a->n_kv)
    if (nkv_p) n_kv = min(n_kv, max(1, nkv_p[0]));
    const int per = (n_kv + n_split - 1) / n_split;
    const int j0 = sp * per, j1 = min(n_kv, j0 + per), cnt = max(0, j1 - j0);
    const int64_t v_nb_dim = a->v_nb_dim;
    const bool v_rows = v_nb_dim == 2;
    if (!v_rows && ((per | n_kv) & 7) != 0) return plan_attn(a_in, c_in, tag_in);   // (uniform) a transposed window is copied 8 positions at a time
    if (a->entry_barrier) plan_lds_barrier();                                   // (after a GEMV stage the staging area is free already: its readers finished a barrier ago)
    plan_prefetch_desc(c, wave, lane);
    PLAN_STAMP(0);
    unsigned long long exp_tab = 0ull;                                         // wave 0, flash graphs: expf_libm's table, fetched while the stage waits for q / k / v
    if (wave == 0 && a->fa_seq) exp_tab = PLAN_EXP2F_T[lane & 31];

    // ---- 0. polls | rope table | window copy ----
    const int nch = (hd + 127) >> 7, n_poll = 3 * nch, n_tab = (half + 63) >> 6;
    if (wave < n_poll) {
        PollCtx pc; pc.sync = c.sync; pc.timeout = c.timeout; pc.t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        const int which = wave / nch, ch = wave % nch;
        VecSrc vs;
        if (which == 0)      { vs.plain = a->q.plain; vs.gran = a->q.gran; vs.tag_off = a->q.tag_off; }
        else if (which == 1) { vs.plain = a->k.plain; vs.gran = a->k.gran; vs.tag_off = a->k.tag_off; }
        else                 { vs.plain = a->v.plain; vs.gran = a->v.gran; vs.tag_off = a->v.tag_off; }
        vs.pad = 0;
        const SrcView s = src_view(vs, (which == 0 ? h : g) * hd, hd, c.epoch);
        float * dst = which == 0 ? sq : which == 1 ? sk : sv;
        float v0 = 0.f, v1 = 0.f;
        bool ok_all = true;
        for (;;) {
            const bool ok = src_try(s, ch, lane, v0, v1) || 128 * ch + 2 * lane >= hd;
            if (__ballot(!ok) == 0ull) break;
            if (!poll_backoff(pc, spins, lane)) { ok_all = false; break; }
        }
        const int e = 128 * ch + 2 * lane;
        if (ok_all && e < hd) { dst[e] = v0; dst[e + 1] = v1; }
        if (!ok_all && lane == 0) c.ctl[CTL_OK] = 0;
    } else if (wave < n_poll + n_tab) {
        // rope table (ops_glue.hip k_rope / ggml-cpu/ops.cpp:5088-5270): theta_ip = pos * theta_scale^ip by REPEATED f32 multiplication, as the CPU's cache
        // init does; the chain is walked by every lane up to its own ip (predicated: the multiplier is 1 beyond it)
        const int ip = 64 * (wave - n_poll) + lane;
        const int n_dims = a->n_dims;
        if (2 * ip < n_dims) {
            const float * ffp = a->freq_factors;
            const float ff = ffp ? ffp[ip] : 1.0f;
            const float tscale = a->theta_scale;
            float th = (float) pos;
            const int steps = (n_dims >> 1) - 1;                               // (uniform) the longest chain
#pragma unroll 8
            for (int j = 0; j < steps; ++j) th = __fmul_rn(th, j < ip ? tscale : 1.0f);
            const float theta_extrap = __fdiv_rn(th, ff);
            const float theta_interp = a->freq_scale * theta_extrap;
            float theta = theta_interp, mscale = a->attn_factor;
            if (a->ext_factor != 0.0f) {
                const float y = ((float) ip - a->corr0) / fmaxf(0.001f, a->corr1 - a->corr0);
                const float ramp_mix = (1.0f - fminf(1.0f, fmaxf(0.0f, y))) * a->ext_factor;
                theta = theta_interp * (1.0f - ramp_mix) + theta_extrap * ramp_mix;
                mscale *= 1.0f + 0.1f * logf(1.0f / a->freq_scale);
            }
            cst[ip] = cosf(theta) * mscale; cst[half + ip] = sinf(theta) * mscale;
        }
        if (wave == n_poll && lane == 0) {                                      // the cache cell this token's row goes to
            const unsigned long long diff = (unsigned long long) (kdst - a->k_cache), np = (unsigned long long) a->k_nb_pos;
            ((int *) misc)[2] = ((diff | np) >> 32) == 0ull ? (int) ((unsigned) diff / (unsigned) np) : (int) (diff / np);
        }
    } else {
        const int wi = wave - n_poll - n_tab, n_w = GEMV_WAVES - n_poll - n_tab;
        if (cnt > 0) {
            // LDS-DMA: 16 bytes per lane, each lane its own source address, lane l of an instruction landing at M0 + 16 l.  An ITEM is U consecutive
            // 16-byte units in memory (a K / V row, or the window's positions of one dim of a transposed cache); 64 / U whole items per instruction,
            // item i landing at byte 16 (U + pad) i of the area.  (One integer division per copy: ~25 instructions of a wave that issues one per 3-5 ns.)
            auto copy = [&](unsigned lds_base, const char * src_base, int64_t item_stride, int n_items, int U, int pad) {
                const int UL = U + pad;                                        // lanes (= 16-byte LDS slots) per item: the pad lanes are switched off
                const int ipi = 64 / UL, li = lane / UL, lc = lane - li * UL;
                for (int i0 = wi * ipi; i0 < n_items; i0 += n_w * ipi) {
                    if (li < ipi && lc < U && i0 + li < n_items) {
                        const char * src = src_base + (int64_t) (i0 + li) * item_stride + 16 * lc;
                        const unsigned dst = (unsigned) __builtin_amdgcn_readfirstlane((int) (lds_base + 16u * (unsigned) (i0 * UL)));
                        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory");
                    }
                }
            };
            const int cpr = hd >> 3;                                           // 16-byte units per row
            const unsigned kw_lds = (unsigned) (size_t) kwin, vw_lds = (unsigned) (size_t) vwin;
            const int64_t knp = a->k_nb_pos;
            copy(kw_lds, a->k_cache + (int64_t) g * a->k_nb_head + (int64_t) j0 * knp, knp, cnt, cpr, 1);
            const char * vb = a->v_cache + (int64_t) g * a->v_nb_head;
            if (v_rows) { const int64_t vnp = a->v_nb_pos; copy(vw_lds, vb + (int64_t) j0 * vnp, vnp, cnt, cpr, 0); }
            else copy(vw_lds, vb + 2 * (int64_t) j0, v_nb_dim, hd, cnt >> 3, 0);  // positions contiguous per dim, 8 per unit: LDS holds [dim][cnt]  (cnt % 8 == 0, j0 % 8 == 0: checked on entry)
        }
        const char * maskp = a->mask; const int mask_f16 = a->mask_f16;
        for (int jj = wi * 64 + lane; jj < cnt; jj += n_w * 64)                 // the additive mask of this split's positions (0 without a mask)
            sc[jj] = maskp ? (mask_f16 ? __half2float(((const __half *) maskp)[j0 + jj]) : ((const float *) maskp)[j0 + jj]) : 0.0f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // the window has landed (the copies were issued first: the mask's trip overlaps theirs)
    }
    // the descriptor fields the later steps need, fetched now (a scalar load first used in step 4 is a 0.2-0.7 us stall there)
    const int n_dims_r = a->n_dims, neox_r = a->neox, p_f16 = a->p_f16, fa_seq = a->fa_seq, plain_out = a->plain;
    const float scale_r = a->scale;
    Granule * const out_gran = a->out_gran, * const part_gran = a->part; float * const out_plain = a->out_plain;
    const int64_t v_dst_nb = a->v_dst_nb;
    asm volatile("" :: "s"(n_dims_r), "s"(neox_r), "s"(p_f16), "s"(fa_seq), "s"(plain_out), "s"(scale_r), "s"(out_gran), "s"(part_gran), "s"(out_plain), "s"(v_dst_nb));
    plan_lds_barrier();
    if (!c.ctl[CTL_OK]) return false;
    PLAN_STAMP(1);

    // ---- 1. rope; this token's rows ----
    const int slot = ((const int *) misc)[2];
    const bool in_win = slot >= j0 && slot < j1, store = sp == 0 && h % gq == 0;     // one workgroup per kv head stores the row
    if (tid < hd) {
        const bool isq = tid < half;
        const float * x = isq ? sq : sk;
        const int ip = isq ? tid : tid - half, i0 = 2 * ip;
        const int n_dims = n_dims_r;
        float r0, r1; int e0, e1;
        if (i0 < n_dims) {
            const float cs = cst[ip], sn = cst[half + ip];
            e0 = neox_r ? ip : i0; e1 = neox_r ? ip + n_dims / 2 : i0 + 1;
            const float x0 = x[e0], x1 = x[e1];
            r0 = x0 * cs - x1 * sn; r1 = x0 * sn + x1 * cs;
        } else { e0 = i0; e1 = i0 + 1; r0 = x[e0]; r1 = x[e1]; }
        const __half h0 = __float2half_rn(r0), h1 = __float2half_rn(r1);
        if (isq) { qh[e0] = h0; qh[e1] = h1; }
        else {
            if (in_win) { kwin[(size_t) (slot - j0) * krs + e0] = h0; kwin[(size_t) (slot - j0) * krs + e1] = h1; }
            if (store) { ((__half *) kdst)[g * hd + e0] = h0; ((__half *) kdst)[g * hd + e1] = h1; }
        }
    } else if (tid < 2 * hd) {
        const int d = tid - hd;
        const __half hv = __float2half_rn(sv[d]);
        if (in_win) { if (v_rows) vwin[(size_t) (slot - j0) * hd + d] = hv; else vwin[(size_t) d * cnt + (slot - j0)] = hv; }
        if (store) *(__half *) (vdst + (int64_t) (g * hd + d) * v_dst_nb) = hv;
    }
    plan_lds_barrier();
    PLAN_STAMP(2);

    // ---- 2. scores: lane = position, IN THE CPU'S SUMMATION ORDER ----
    // ggml_vec_dot_f16 (ggml-cpu/vec.cpp:128-168, AVX2: GGML_F16_STEP 32, 8 lanes, 4 accumulators): element e of the row feeds accumulator
    // (j, l) = ((e % 32) / 8, e % 8) by an f32 fma, blocks of 32 in order; then per l: (a0 + a2) + (a1 + a3) =: s[l]; t[i] = s[i] + s[i + 4];
    // result (t0 + t1) + (t2 + t3).  The chains of different l are independent: they are dealt to 1, 2 or 4 waves per 64 positions (each takes
    // whole t[i] subtrees), whose partial results meet in LDS and are added in the tree's order.  The score therefore has the CPU's bits, and with
    // it the f16 roundings further down (probabilities of the non-flash graph, the flash accumulator) fall the same way.
    {
        const int G = (cnt + 63) >> 6;                                          // position groups of 64 (uniform; <= 5)
        const int lg = G <= 1 ? 0 : G <= 2 ? 1 : G <= 4 ? 2 : 3;                // waves are dealt to 2^lg group slots; slots >= G idle
        const int grp = wave & ((1 << lg) - 1), part = wave >> lg;
        const int pstride = 64 << lg;
        int lp = lg <= 2 ? 2 : 1;                                               // 4 parts (2 with 8 slots) ...
        while (lp > 0 && (pstride << lp) > 12 * hd) --lp;                       // ... as far as their results fit pbuf (plan creation made sure one part does)
        const int n_parts = 1 << lp, ni = 4 >> lp;                              // part p owns t[i], i in [p ni, (p + 1) ni)
        const int jj = 64 * grp + lane;
        if (grp < G && part < n_parts && jj < cnt) {
            const __half * krow = kwin + (size_t) jj * krs;
            const int nblk = hd >> 5;
            float tt[4] = { 0.f, 0.f, 0.f, 0.f };
            for (int ii = 0; ii < ni; ++ii) {                                   // (uniform)
                const int i = part * ni + ii;
                float t;
                switch (i) {                                                    // (uniform; the element index becomes a constant: half selects instead of shifts)
                case 0:  t = attn_score_t<0>(krow, qh, nblk); break;
                case 1:  t = attn_score_t<1>(krow, qh, nblk); break;
                case 2:  t = attn_score_t<2>(krow, qh, nblk); break;
                default: t = attn_score_t<3>(krow, qh, nblk); break;
                }
                tt[ii] = t;
            }
            pbuf[part * pstride + jj] = ni == 1 ? tt[0] : ni == 2 ? __fadd_rn(tt[0], tt[1]) : __fadd_rn(__fadd_rn(tt[0], tt[1]), __fadd_rn(tt[2], tt[3]));
        }
        plan_lds_barrier();
        if (wave < G) {                                                         // wave = position group
            const int j2 = 64 * wave + lane;
            if (j2 < cnt) {
                float dot;
                if (n_parts == 4)      dot = __fadd_rn(__fadd_rn(pbuf[j2], pbuf[pstride + j2]), __fadd_rn(pbuf[2 * pstride + j2], pbuf[3 * pstride + j2]));
                else if (n_parts == 2) dot = __fadd_rn(pbuf[j2], pbuf[pstride + j2]);
                else                   dot = pbuf[j2];
                // (a fully masked position stays -inf whatever its cache row holds: never-written rows may be anything, 0 * NaN included)
                const float m = sc[j2];
                sc[j2] = m == -INFINITY ? -INFINITY : __fadd_rn(__fmul_rn(dot, scale_r), m);
            }
        }
    }
    plan_lds_barrier();
    PLAN_STAMP(3);

    // ---- 3. softmax statistics: wave 0, at most NV scores per lane in registers ----
    constexpr int NV = 5;                                                       // (per <= 320: guaranteed by the LDS budget check at plan creation)
    if (wave == 0) {
        float s[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) s[v] = 64 * v + lane < cnt ? sc[64 * v + lane] : -INFINITY;
        float mx0 = s[0];
#pragma unroll
        for (int v = 1; v < NV; ++v) mx0 = fmaxf(mx0, s[v]);
        mx0 = wave_max_f(mx0);
        float l0 = 1.0f;
        if (fa_seq) {
            // the CPU's online softmax, positions in order jj = 64 v + lane: Mprev = the maximum before jj; a new maximum rescales the accumulator by
            // expf(Mprev - s) and adds v with weight 1, otherwise the weight is expf(s - Mprev).  Masked positions (-inf) are skipped by the CPU.
            float carry = -INFINITY; int last = 0;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (64 * v < cnt) {                                             // (uniform)
                    float x = s[v];                                             // inclusive prefix maximum over the lanes
                    x = fmaxf(x, dpp_keep_f<0x111>(-INFINITY, x)); x = fmaxf(x, dpp_keep_f<0x112>(-INFINITY, x));
                    x = fmaxf(x, dpp_keep_f<0x114>(-INFINITY, x)); x = fmaxf(x, dpp_keep_f<0x118>(-INFINITY, x));      // row_shr 1, 2, 4, 8: each row of 16 scanned
                    const float r0 = readlane_f(x, 15), r1 = fmaxf(r0, readlane_f(x, 31)), r2 = fmaxf(r1, readlane_f(x, 47));
                    x = fmaxf(x, lane < 16 ? -INFINITY : lane < 32 ? r0 : lane < 48 ? r1 : r2);
                    float prev = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * ((lane + 63) & 63), __float_as_int(x)));   // lane - 1's inclusive maximum
                    prev = fmaxf(lane == 0 ? -INFINITY : prev, carry);
                    carry = fmaxf(carry, readlane_f(x, 63));
                    const float sj = s[v];
                    // (both exponentials are evaluated by every lane: expf_libm reads its table across the lanes)
                    const bool live_j = sj != -INFINITY, newmax = live_j && sj > prev;
                    const float e = expf_libm(newmax ? __fsub_rn(prev, sj) : live_j ? __fsub_rn(sj, prev) : 0.0f, exp_tab);
                    const float ms = newmax ? e : 1.0f, vs = newmax ? 1.0f : live_j ? e : 0.0f;
                    if (64 * v + lane < cnt) { sc[64 * v + lane] = vs; msv[64 * v + lane] = ms; }
                    const unsigned long long live = __ballot(vs != 0.0f || ms != 1.0f);
                    if (live) last = 64 * v + 64 - __builtin_clzll(live);
                }
            }
            if (lane == 0) ((int *) misc)[3] = last;                           // (the sum S is walked with the accumulator in step 4, in the CPU's order)
        } else if (p_f16) {
            // The non-flash graph of the reference (SOFT_MAX, then MUL_MAT(v, kq) whose f16 src0 makes the CPU round kq to f16): the whole window is in
            // this workgroup, so the probabilities are formed exactly as ggml_compute_forward_soft_max_f32 does -- sum of the exponentials in f64,
            // p = e * (float) (1 / sum) -- and rounded to f16 before they meet V.
            double ds = 0.0;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (64 * v < cnt) {                                             // (uniform: a group of 64 beyond the window costs nothing)
                    s[v] = (64 * v + lane < cnt && mx0 != -INFINITY) ? expf(__fsub_rn(s[v], mx0)) : 0.0f;
                    ds += (double) s[v];
                }
            }
            ds = wave_sum_f64(ds);
            const float inv = (float) (1.0 / ds);
#pragma unroll
            for (int v = 0; v < NV; ++v) if (64 * v < cnt) { if (64 * v + lane < cnt) sc[64 * v + lane] = f16_round(__fmul_rn(s[v], inv)); }
        } else {
            float ls = 0.0f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (64 * v < cnt) {
                    const float p = (64 * v + lane < cnt && mx0 != -INFINITY) ? expf(__fsub_rn(s[v], mx0)) : 0.0f;
                    if (64 * v + lane < cnt) sc[64 * v + lane] = p;
                    ls += p;
                }
            }
            l0 = wave_sum(ls);
        }
        if (lane == 0) { misc[0] = mx0; misc[1] = l0; }
    }
    plan_lds_barrier();
    const float mx = misc[0], l = misc[1];
    PLAN_STAMP(4);

    // ---- 4. o[d] = sum_j p_j v[j][d]   (positions with p == 0 are skipped: masked cache rows may hold anything) ----
    float o_own = 0.0f;                                                         // fa_seq: thread d's result
    if (fa_seq) {
        if (tid < hd) {
            const int last = ((const int *) misc)[3];
            float acc = 0.0f, S = 0.0f;                                         // acc: an f16 value at all times (the CPU's VKQ16);  S = S * ms + vs as the CPU walks it
#pragma unroll 4
            for (int j = 0; j < last; ++j) {
                const float vs = sc[j], ms = msv[j];
                const float vv = __half2float(vwin[(size_t) j * hd + tid]);
                if (vs != 0.0f || ms != 1.0f) {                                 // (uniform) the CPU skips masked positions altogether
                    if (ms != 1.0f) acc = f16_round(__fmul_rn(acc, ms));         // ggml_vec_scale_f16
                    acc = f16_round(__builtin_fmaf(vv, vs, acc));                // ggml_vec_mad_f16 (f32 fma, then back to f16)
                    S = __fadd_rn(__fmul_rn(S, ms), vs);
                }
            }
            o_own = __fmul_rn(acc, __fdiv_rn(1.0f, S));                         // V *= 1 / S
        }
    } else if (v_rows) {
        // a wave (8 of them) takes positions w, w + 8, ...; a lane owns dims (2l, 2l+1) [+128]
        if (wave < 8) {
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll 4
            for (int j = wave; j < cnt; j += 8) {
                const float p = sc[j];
                if (p != 0.0f) {                                                // (uniform)
                    const __half * row = vwin + (size_t) j * hd;
                    if (2 * lane < hd)       { const float2 f = __half22float2(*(const __half2 *) (row + 2 * lane)); o0 += p * f.x; o1 += p * f.y; }
                    if (2 * lane + 128 < hd) { const float2 f = __half22float2(*(const __half2 *) (row + 2 * lane + 128)); o2 += p * f.x; o3 += p * f.y; }
                }
            }
            if (2 * lane < hd) { red[wave * hd + 2 * lane] = o0; red[wave * hd + 2 * lane + 1] = o1; }
            if (2 * lane + 128 < hd) { red[wave * hd + 2 * lane + 128] = o2; red[wave * hd + 2 * lane + 129] = o3; }
        }
        plan_lds_barrier();
        if (tid < hd) {
            float o = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) o += red[i * hd + tid];
            o_own = o;
        }
    } else {
        // transposed window [dim][cnt]: the CPU's MUL_MAT(v, p) is ggml_vec_dot_f16 over the positions (see step 2: blocks of 32, accumulators
        // (j, l)).  Thread (d = tid / 8, l = tid % 8) owns the four accumulators (j, l) of dim d; the eight threads of a dim then walk the tree on DPP.
        // Positions beyond the last whole block are added the way the CPU's tail loop does (f32 product, f64 sum).
        for (int d = tid >> 3; d < hd; d += GEMV_THREADS >> 3) {                // (hd is a multiple of 32: whole waves stay together)
            const int l = tid & 7;
            const __half * vrow = vwin + (size_t) d * cnt;
            const int np = cnt & ~31;
            float acc[4] = { 0.f, 0.f, 0.f, 0.f };
            for (int e0 = 0; e0 < np; e0 += 32) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float p = sc[e0 + 8 * j + l];
                    // (a probability of 0 must not meet a never-written cache cell: 0 * NaN; fma(v, 0, acc) == acc for every finite v)
                    if (p != 0.0f) acc[j] = __builtin_fmaf(__half2float(vrow[e0 + 8 * j + l]), p, acc[j]);
                }
            }
            float sl = __fadd_rn(__fadd_rn(acc[0], acc[2]), __fadd_rn(acc[1], acc[3]));
            sl = __fadd_rn(sl, dpp_f<0x104>(sl));                               // row_shl:4 -- lane l < 4 of each 8: s[l] + s[l + 4]   (the other lanes' values are not used)
            sl = __fadd_rn(sl, dpp_f<0xB1>(sl));                                // t0 + t1 (lane 0), t2 + t3 (lane 2)
            sl = __fadd_rn(sl, dpp_f<0x4E>(sl));                                // (t0 + t1) + (t2 + t3)
            if (l == 0) {
                if (np < cnt) {
                    double sumf = (double) sl;
                    for (int e = np; e < cnt; ++e) { const float p = sc[e]; if (p != 0.0f) sumf += (double) __fmul_rn(__half2float(vrow[e]), p); }
                    sl = (float) sumf;
                }
                red[d] = sl;
            }
        }
        plan_lds_barrier();
        if (tid < hd) o_own = red[tid];
    }
    PLAN_STAMP(7);
    // ---- publish: the normalized output when the head is not split, else (o, m, l) of this split for the merge stage ----
    if (n_split == 1) {
        if (tid < hd) {
            const float r = (p_f16 || fa_seq) ? o_own : __fdiv_rn(o_own, l);
            publish(out_gran + h * hd + tid, r, tag);
            if (plain_out) out_plain[h * hd + tid] = r;
        }
    } else {
        Granule * part = part_gran + (size_t) ((size_t) h * n_split + sp) * (hd + 2);
        if (tid < hd) publish(part + tid, o_own, tag);
        else if (tid == hd) publish(part + hd, mx, tag);
        else if (tid == hd + 1) publish(part + hd + 1, l, tag);
    }
    PLAN_STAMP(5);
    return true;
}

// merge the KV splits of a head: out = sum_s e^{m_s - M} o_s / sum_s e^{m_s - M} l_s     (workgroup h * n_split does head h)
static __device__ __noinline__ bool plan_attn_combine(AttnC a_in, const StageCtx & c_in, unsigned tag_in) {
    const AttnC a = (AttnC) (uintptr_t) uniform_ptr((const void *) (uintptr_t) a_in);
    const unsigned tag = (unsigned) __builtin_amdgcn_readfirstlane((int) tag_in);
    extern __shared__ __attribute__((aligned(16))) uint8_t plan_lds_comb[];
    StageCtx c;
    c.image = __builtin_amdgcn_readfirstlane(c_in.image);
    c.lds = plan_lds_comb; c.ctl = (int *) (plan_lds_comb + c.image); c.part = (double *) (plan_lds_comb + c.image + 64);
    c.stg = (float *) (plan_lds_comb + c.image + 64 + 8 * GEMV_WAVES + 1024);
    c.sync = uniform_ptr(c_in.sync); c.timeout = c_in.timeout;
    c.grid = (unsigned) __builtin_amdgcn_readfirstlane((int) c_in.grid); c.epoch = (unsigned) __builtin_amdgcn_readfirstlane((int) c_in.epoch);
    c.even = c_in.even; c.stage = __builtin_amdgcn_readfirstlane(c_in.stage);
    c.next_desc = nullptr; c.next_attn = nullptr; c.next_desc2 = nullptr; c.pre_lds = c_in.pre_lds;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int tid = (int) threadIdx.x;
    const int hd = a->hd, n_split = a->n_split;
    if ((int) blockIdx.x >= a->n_head * n_split || (int) blockIdx.x % n_split != 0) return true;
    const int h = (int) blockIdx.x / n_split;
    const int n = n_split * (hd + 2);
    float * buf = c.stg;
    plan_lds_barrier();
    PLAN_STAMP(0);
    {
        VecSrc vs; vs.plain = nullptr; vs.gran = a->part + (size_t) h * n; vs.tag_off = 0; vs.pad = 0;
        SrcView s = src_view(vs, 0, n, 0); s.expect = tag - 1;                  // the partials carry the ATTN stage's tag (the stage before this one)
        PollCtx pc; pc.sync = c.sync; pc.timeout = c.timeout; pc.t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        bool ok_all = true;
        for (int ch = wave; ch < ((n + 127) >> 7); ch += GEMV_WAVES) {
            float v0, v1;
            for (;;) {
                const bool ok = src_try(s, ch, lane, v0, v1) || 128 * ch + 2 * lane >= n;
                if (__ballot(!ok) == 0ull) break;
                if (!poll_backoff(pc, spins, lane)) { ok_all = false; break; }
            }
            if (!ok_all) break;
            const int e = 128 * ch + 2 * lane;
            if (e < n) { buf[e] = v0; buf[e + 1] = v1; }
        }
        if (!ok_all && lane == 0) c.ctl[CTL_OK] = 0;
    }
    plan_lds_barrier();
    if (!c.ctl[CTL_OK]) return false;
    if (tid < hd) {
        float M = -INFINITY;
        for (int s = 0; s < n_split; ++s) M = fmaxf(M, buf[s * (hd + 2) + hd]);
        float L = 0.0f, o = 0.0f;
        for (int s = 0; s < n_split; ++s) {
            const float m = buf[s * (hd + 2) + hd];
            const float w = m == -INFINITY ? 0.0f : expf(__fsub_rn(m, M));
            L += w * buf[s * (hd + 2) + hd + 1];
            o += w * buf[s * (hd + 2) + tid];
        }
        const float r = __fdiv_rn(o, L);
        publish(a->out_gran + h * hd + tid, r, tag);
        if (a->plain) a->out_plain[h * hd + tid] = r;
    }
    PLAN_STAMP(2);
    return true;
}

template <unsigned SET>
__global__ void __launch_bounds__(GEMV_THREADS)
k_plan(const PlanStage * stages_g, int n_stages, unsigned * sync, int even, unsigned long long timeout_ticks, int lds_image_bytes, unsigned epoch) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    StageC stages = (StageC) stages_g;
    if (__hip_atomic_load(sync + PLAN_SYNC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;   // sticky: a plan that timed out stays dead

    StageCtx c;
    c.lds = lds; c.ctl = (int *) (lds + lds_image_bytes); c.part = (double *) (lds + lds_image_bytes + 64);
    c.stg = (float *) (lds + lds_image_bytes + 64 + 8 * GEMV_WAVES + 1024);
    c.pre_lds = (unsigned) (size_t) (lds + lds_image_bytes + 64 + 8 * GEMV_WAVES);
    c.sync = sync; c.timeout = timeout_ticks; c.grid = gridDim.x; c.even = even; c.epoch = epoch; c.image = lds_image_bytes;
    if (threadIdx.x == 0) c.ctl[CTL_OK] = 1;
    plan_lds_barrier();

#pragma unroll 1
    for (int s = 0; s < n_stages; ++s) {
        StageC st = stages + s;
        // A GROUP: sub-stages of one activation vector whose matrices have different weight types (wq | wk in Q4_K, wv in Q6_K) run concurrently, each on
        // its own share of the workgroups (by weight bytes), instead of one after the other on all of them: the second used to cost a whole stage chain
        // (descriptor, prime, memory latency: 4.6 us) for 3.4 MB.  Every member forms the activation image itself.
        const int grp = st->group;
        if (grp > 1) {
            int m = 0;
            for (int i = 1; i < grp; ++i) if ((int) blockIdx.x >= (stages + s + i)->wg_base) m = i;      // (wg_base ascending over the members)
            st = stages + s + m;
        }
        c.stage = (int) (st - stages);
        s += grp > 1 ? grp - 1 : 0;
        // (a GEMV stage may leave workgroups out: the one behind an attention stage is not given to the workgroups that ran the attention -- they arrive last,
        //  and with rows of their own the whole stage would end a descriptor + prime + memory latency later)
        if (grp == 1 && ((int) blockIdx.x < st->wg_base || (int) blockIdx.x >= st->wg_base + st->wg_count)) continue;
        c.next_desc = s + 1 < n_stages ? (const uint8_t *) (stages_g + s + 1) : nullptr;
        c.next_desc2 = s + 2 < n_stages ? (const uint8_t *) (stages_g + s + 2) : nullptr;
        c.next_attn = (const uint8_t *) st->next_attn;
        bool ok = true;
        const int kind = st->kind;
        if (kind == PLAN_K_GEMV) {
            switch (st->type) {
            case MI355Q_TYPE_Q4_K: if constexpr ((SET & tbit(MI355Q_TYPE_Q4_K)) != 0) ok = plan_stage<MI355Q_TYPE_Q4_K>(st, c); break;
            case MI355Q_TYPE_Q5_K: if constexpr ((SET & tbit(MI355Q_TYPE_Q5_K)) != 0) ok = plan_stage<MI355Q_TYPE_Q5_K>(st, c); break;
            case MI355Q_TYPE_Q6_K: if constexpr ((SET & tbit(MI355Q_TYPE_Q6_K)) != 0) ok = plan_stage<MI355Q_TYPE_Q6_K>(st, c); break;
            case MI355Q_TYPE_Q8_0: if constexpr ((SET & tbit(MI355Q_TYPE_Q8_0)) != 0) ok = plan_stage<MI355Q_TYPE_Q8_0>(st, c); break;
            case MI355Q_TYPE_Q4_0: if constexpr ((SET & tbit(MI355Q_TYPE_Q4_0)) != 0) ok = plan_stage<MI355Q_TYPE_Q4_0>(st, c); break;
            case MI355Q_TYPE_IQ4_NL: if constexpr ((SET & tbit(MI355Q_TYPE_IQ4_NL)) != 0) ok = plan_stage<MI355Q_TYPE_IQ4_NL>(st, c); break;
            case MI355Q_TYPE_IQ4_XS: if constexpr ((SET & tbit(MI355Q_TYPE_IQ4_XS)) != 0) ok = plan_stage<MI355Q_TYPE_IQ4_XS>(st, c); break;
            default: break;
            }
        } else if (kind == PLAN_K_ATTN) {
            ok = (st->flags & PLAN_F_ATTN_LDS) ? plan_attn_lds((AttnC) st->attn, c, epoch + st->tag_off) : plan_attn((AttnC) st->attn, c, epoch + st->tag_off);
        } else {
            ok = plan_attn_combine((AttnC) st->attn, c, epoch + st->tag_off);
        }
        if (!ok) return;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct Plan {
    int           device = 0, n_cu = 0, grid = 0, n_stages = 0, even = 0;
    unsigned      set = 0;
    size_t        lds_image = 0, lds_total = 0;
    int64_t       weight_bytes = 0;
    unsigned char * d_block = nullptr; size_t block_bytes = 0; // the one device allocation the four pointers below point into
    PlanStage *   d_stages = nullptr;
    AttnStage *   d_attn = nullptr;
    unsigned *    d_sync = nullptr;
    Granule *     d_gran = nullptr; size_t gran_count = 0;
    unsigned long long runs = 0, timeout_ticks = 0;
};

int gemv_fast_family(int type);

// Device blocks of destroyed plans, kept for the next plan: a generation builds one plan per KV window (32 positions) and drops the least recently used
// one, and hipMalloc + hipFree of the ~9 MB block (descriptors + one granule per stage output element) was most of the build's 1.6 ms.  At most
// PLAN_POOL blocks are held (per process; the caller of destroy has synchronized the plan's stream, so a block handed on is idle).
enum { PLAN_POOL = 4 };
static std::mutex g_pool_mu;
static struct { unsigned char * p; size_t bytes; int dev; } g_pool[PLAN_POOL];
static bool plan_block_take(int dev, size_t need, unsigned char ** out, size_t * out_bytes) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int best = -1;
    for (int i = 0; i < PLAN_POOL; ++i) if (g_pool[i].p && g_pool[i].dev == dev && g_pool[i].bytes >= need && g_pool[i].bytes <= 2 * need + (1u << 20) && (best < 0 || g_pool[i].bytes < g_pool[best].bytes)) best = i;
    if (best < 0) return false;
    *out = g_pool[best].p; *out_bytes = g_pool[best].bytes; g_pool[best].p = nullptr;
    return true;
}
static void plan_block_give(int dev, unsigned char * p, size_t bytes) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (int i = 0; i < PLAN_POOL; ++i) if (!g_pool[i].p) { g_pool[i].p = p; g_pool[i].bytes = bytes; g_pool[i].dev = dev; return; }
    }
    (void) hipFree(p);
}

static const void * plan_kernel(unsigned set) {
    if ((set & ~SET_K46) == 0)  return (const void *) k_plan<SET_K46>;
    if ((set & ~SET_K456) == 0) return (const void *) k_plan<SET_K456>;
    if ((set & ~SET_80) == 0)   return (const void *) k_plan<SET_80>;
    if ((set & ~SET_ALL) == 0)  return (const void *) k_plan<SET_ALL>;
    if ((set & ~SET_IQ4) == 0)  return (const void *) k_plan<SET_IQ4>;
    return nullptr;                                            // (a mix of the IQ4 types with Q4_K / Q8_0 / Q4_0: no instantiation)
}

#ifdef MI355Q_STAMPS
extern "C" int mi355q_debug_set_plan_stamps(void * dev_buf, int n_stages) {
    unsigned long long * p = (unsigned long long *) dev_buf;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_plan_stamps), &p, sizeof(p)) != hipSuccess) return -4;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_plan_stamp_stages), &n_stages, sizeof(int)) == hipSuccess ? 0 : -4;
}
#endif

} // namespace mi355q

using namespace mi355q;

extern "C" {

void mi355q_set_error(const char * msg);          // api.hip

// outputs published so far while the stage list is built: [ptr, ptr + n) f32 <-> granule offset, producing stage
namespace { struct OutRange { const float * p; int64_t n; size_t gran_off; unsigned tag_off; int64_t id; }; }

int mi355q_regs_plan_create(mi355q_plan ** out, const mi355q_stage * stages, int n_stages, int flags) {
    if (!out || !stages || n_stages < 1) { mi355q_set_error("plan_create: null argument / no stages"); return MI355Q_ERR_SHAPE; }
    const auto t_create0 = std::chrono::steady_clock::now();
    int dev = 0;
    static int cu_of[64];                                      // (hipGetDeviceProperties fills a 1.5 KB struct through the driver: once per device, not once per plan)
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { mi355q_set_error("plan_create: no device"); return MI355Q_ERR_HIP; }
    if (!cu_of[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { mi355q_set_error("plan_create: no device"); return MI355Q_ERR_HIP; }
        cu_of[dev] = prop.multiProcessorCount;
    }
    const int n_cu = cu_of[dev];

    std::vector<PlanStage> v;
    std::vector<AttnStage> va;
    std::vector<int> attn_of;                                  // per internal stage: index into va or -1
    std::vector<OutRange> outs;
    size_t gran_count = 0;
    auto new_out = [&](const float * p, int64_t n, unsigned tag_off, int64_t id) {
        OutRange r = { p, n, gran_count, tag_off, id };
        gran_count += (size_t) ((n + 1) & ~(int64_t) 1);       // keep every vector 16-byte aligned
        outs.push_back(r);
        return r.gran_off;
    };
    // operand -> plain or the granules of the LATEST earlier output that contains it
    auto resolve = [&](const float * p, int64_t n, int64_t id, VecSrc & vs) -> bool {
        vs.plain = p; vs.gran = nullptr; vs.tag_off = 0; vs.pad = 0;
        if (!p) return true;
        auto take = [&](const OutRange & r) { vs.gran = (const Granule *) (uintptr_t) (r.gran_off + (size_t) (p - r.p) + 1); vs.tag_off = r.tag_off; vs.plain = nullptr; };   // (offset + 1: patched to a pointer below)
        if (id != 0) {                                         // labelled operand: the latest output with this id; the addresses give the offset inside it
            for (size_t i = outs.size(); i-- > 0;) if (outs[i].id == id) { if (p < outs[i].p || p + n > outs[i].p + outs[i].n) return false; take(outs[i]); return true; }
            return false;                                      // no stage produces this value
        }
        // By address: the LATEST earlier output that CONTAINS the operand produced it.  Later outputs that merely overlap it are other tensors to
        // which the caller's allocator has handed the (by then dead) memory: inside the plan every value lives in its own granules, so they do not matter.
        bool overlapped = false;
        for (size_t i = outs.size(); i-- > 0;) {
            const OutRange & r = outs[i];
            if (p >= r.p && p + n <= r.p + r.n) { if (r.gran_off == (size_t) -1) return false; take(r); return true; }      // (inside an x_out vector: stored plainly DURING the launch, not readable by a later stage)
            if (p < r.p + r.n && r.p < p + n) overlapped = true;
        }
        return !overlapped;                                    // a plain operand whose memory a stage of this plan overwrites: not expressible
    };
    unsigned set = 0; size_t lds_max = 0, stg_max = 0; int64_t bytes = 0;
    for (int s = 0; s < n_stages; ++s) {
        const mi355q_stage & in = stages[s];
        if (in.kind == MI355Q_STAGE_ATTN) {
            const mi355q_attn * at = in.attn;
            if (!at || !at->q || !at->k || !at->v || !at->pos || !at->k_cache || !at->v_cache || !at->k_dst || !at->v_dst || !at->out) { mi355q_set_error("plan_create: attn stage: null pointer"); return MI355Q_ERR_SHAPE; }
            if (at->n_head < 1 || at->n_head_kv < 1 || at->n_head % at->n_head_kv || at->head_dim < 32 || at->head_dim > 256 || at->head_dim % 32 || at->n_kv < 1) { mi355q_set_error("plan_create: attn stage: bad head geometry"); return MI355Q_ERR_SHAPE; }
            if (at->rope.mode != 0 && at->rope.mode != 2) { mi355q_set_error("plan_create: attn stage: rope mode must be 0 or 2"); return MI355Q_ERR_UNSUPPORTED; }
            if (at->rope.n_dims <= 0 || at->rope.n_dims % 2 || at->rope.n_dims > at->head_dim) { mi355q_set_error("plan_create: attn stage: rope n_dims"); return MI355Q_ERR_SHAPE; }
            if (at->v_nb_dim != 2 && at->v_nb_pos != 2) { mi355q_set_error("plan_create: attn stage: V cache must be contiguous along head_dim or along positions"); return MI355Q_ERR_UNSUPPORTED; }
            if (at->n_head > n_cu) { mi355q_set_error("plan_create: attn stage: more heads than CUs"); return MI355Q_ERR_UNSUPPORTED; }
            AttnStage A = {};
            const int hd = at->head_dim;
            if (!resolve(at->q, (int64_t) at->n_head * hd, at->q_id, A.q) || !resolve(at->k, (int64_t) at->n_head_kv * hd, at->k_id, A.k) || !resolve(at->v, (int64_t) at->n_head_kv * hd, at->v_id, A.v)) { mi355q_set_error("plan_create: attn operand straddles an earlier output"); return MI355Q_ERR_SHAPE; }
            A.pos = at->pos; A.n_kv_dev = at->n_kv_dev; A.freq_factors = at->freq_factors; A.k_cache = (const char *) at->k_cache; A.v_cache = (const char *) at->v_cache;
            A.k_nb_pos = at->k_nb_pos; A.k_nb_head = at->k_nb_head; A.v_nb_pos = at->v_nb_pos; A.v_nb_dim = at->v_nb_dim; A.v_nb_head = at->v_nb_head; A.v_dst_nb = at->v_dst_nb;
            A.k_dst = (char * const *) at->k_dst; A.v_dst = (char * const *) at->v_dst; A.mask = (const char *) at->mask; A.mask_f16 = at->mask_f16;
            A.n_head = at->n_head; A.n_head_kv = at->n_head_kv; A.hd = hd; A.n_kv = at->n_kv; A.scale = at->scale;
            // KV splits per head: one workgroup per 256 positions, at most #CU / n_head.  A window of <= 256 positions is ONE workgroup per head and
            // needs no merge stage (a dependent hop costs more than reading 256 cache rows).
            A.n_split = (at->n_kv + 255) / 256; if (A.n_split > n_cu / at->n_head) A.n_split = n_cu / at->n_head; if (A.n_split < 1) A.n_split = 1;
            if (const char * e = getenv("MI355Q_PLAN_KV_SPLIT")) { const int sp = atoi(e); if (sp >= 1 && sp <= n_cu / at->n_head) A.n_split = sp; }
            A.per = (at->n_kv + A.n_split - 1) / A.n_split;
            A.p_f16 = A.n_split == 1 && at->v_nb_pos == 2;     // the non-flash graph with the window in one workgroup: the CPU's f16-rounded probabilities
            A.plain = (in.flags & MI355Q_STAGE_NO_PLAIN) ? 0 : 1; A.out_plain = at->out;
            A.n_dims = at->rope.n_dims; A.neox = at->rope.mode == 2; A.freq_scale = at->rope.freq_scale; A.ext_factor = at->rope.ext_factor; A.attn_factor = at->rope.attn_factor;
            A.theta_scale = powf(at->rope.freq_base, -2.0f / at->rope.n_dims);
            {   // ggml_rope_yarn_corr_dims, ggml.c:3729-3743 (as mi355q_op_rope)
                auto corr_dim = [&](float n_rot) { return at->rope.n_dims * logf(at->rope.n_ctx_orig / (n_rot * 2 * 3.14159265358979323846f)) / (2 * logf(at->rope.freq_base)); };
                const float start = floorf(corr_dim(at->rope.beta_fast)), end = ceilf(corr_dim(at->rope.beta_slow));
                A.corr0 = start > 0 ? start : 0; A.corr1 = end < at->rope.n_dims - 1 ? end : (float) (at->rope.n_dims - 1);
            }
            // internal stages: attention per (head, split) and, when the heads are split, the merge
            PlanStage p = {}; p.kind = PLAN_K_ATTN; p.tag_off = (unsigned) v.size() + 1; p.flags = PLAN_F_NEW_X;
            if (A.n_split == 1) {
                A.out_gran = (Granule *) (uintptr_t) (new_out(at->out, (int64_t) at->n_head * hd, p.tag_off, at->out_id) + 1);
                v.push_back(p); attn_of.push_back((int) va.size());
            } else {
                const size_t part_off = gran_count; gran_count += (size_t) at->n_head * A.n_split * (hd + 2); gran_count = (gran_count + 1) & ~(size_t) 1;
                A.part = (Granule *) (uintptr_t) (part_off + 1);
                v.push_back(p); attn_of.push_back((int) va.size());
                PlanStage q = {}; q.kind = PLAN_K_COMBINE; q.tag_off = (unsigned) v.size() + 1; q.flags = PLAN_F_NEW_X;
                A.out_gran = (Granule *) (uintptr_t) (new_out(at->out, (int64_t) at->n_head * hd, q.tag_off, at->out_id) + 1);
                v.push_back(q); attn_of.push_back((int) va.size());
            }
            va.push_back(A);
            const size_t need = (size_t) 4 * (3 * hd + hd /* kh, vh */ + 2 * GEMV_WAVES + GEMV_WAVES * hd + A.per) + 64;
            const size_t need2 = (size_t) 4 * A.n_split * (hd + 2) + 64;
            if (need > stg_max) stg_max = need;
            if (need2 > stg_max) stg_max = need2;
            bytes += (int64_t) 2 * at->n_kv * at->n_head_kv * hd * 2;
            continue;
        }
        if (in.kind != MI355Q_STAGE_GEMV) { mi355q_set_error("plan_create: unknown stage kind"); return MI355Q_ERR_UNSUPPORTED; }
        if (in.n_mats < 1 || in.n_mats > GEMV_MAX_MATS || !in.x || in.k <= 0 || (in.k & 31) || ((uintptr_t) in.x & 7)) { mi355q_set_error("plan_create: bad stage"); return MI355Q_ERR_SHAPE; }
        if (in.x_kind < MI355Q_X_PLAIN || in.x_kind > MI355Q_X_UNARY_MUL) { mi355q_set_error("plan_create: unknown x_kind"); return MI355Q_ERR_UNSUPPORTED; }
        if (in.x_kind == MI355Q_X_UNARY_MUL && (!in.x1 || (in.x_unary != MI355Q_UNARY_SILU && in.x_unary != MI355Q_UNARY_RELU && in.x_unary != MI355Q_UNARY_SIGMOID))) { mi355q_set_error("plan_create: X_UNARY_MUL needs x1 and SILU / RELU / SIGMOID"); return MI355Q_ERR_UNSUPPORTED; }
        if (in.x_kind == MI355Q_X_PLAIN && in.x1) { mi355q_set_error("plan_create: X_PLAIN takes one operand"); return MI355Q_ERR_SHAPE; }
        if (in.x_kind == MI355Q_X_NORM && (((uintptr_t) in.norm_w & 15) || (in.k & 3))) { mi355q_set_error("plan_create: norm weights must be 16-byte aligned"); return MI355Q_ERR_ALIGN; }
        if (in.x1 && ((uintptr_t) in.x1 & 7)) { mi355q_set_error("plan_create: x1 must be 8-byte aligned"); return MI355Q_ERR_ALIGN; }
        if (in.x_out && (((uintptr_t) in.x_out & 15) || (in.k & 3) || in.x_kind == MI355Q_X_PLAIN)) { mi355q_set_error("plan_create: x_out needs X_NORM / X_UNARY_MUL, 16-byte alignment and k % 4 == 0"); return MI355Q_ERR_ALIGN; }
        {   // a plain store of this stage must not land on a plain operand the same stage still reads in other workgroups (ggml-alloc makes ADD in place)
            auto overlaps = [](const float * a, int64_t na, const float * b, int64_t nb) { return a && b && a < b + nb && b < a + na; };
            VecSrc t0, t1; const bool r0 = resolve(in.x, in.k, in.x_id, t0), r1 = resolve(in.x1, in.k, in.x1_id, t1);
            const bool p0 = r0 && t0.plain != nullptr, p1 = r1 && t1.plain != nullptr;
            if (in.sum_out && !(in.flags & MI355Q_STAGE_NO_PLAIN) && ((p0 && overlaps(in.sum_out, in.k, in.x, in.k)) || (p1 && overlaps(in.sum_out, in.k, in.x1, in.k)))) { mi355q_set_error("plan_create: sum_out overlaps a plain operand of the same stage"); return MI355Q_ERR_SHAPE; }
            if (in.x_out && ((p0 && overlaps(in.x_out, in.k, in.x, in.k)) || (p1 && overlaps(in.x_out, in.k, in.x1, in.k)))) { mi355q_set_error("plan_create: x_out overlaps a plain operand of the same stage"); return MI355Q_ERR_SHAPE; }
            for (int i = 0; i < in.n_mats; ++i) if (!(in.flags & MI355Q_STAGE_NO_PLAIN) && ((p0 && overlaps(in.mats[i].y, in.mats[i].m, in.x, in.k)) || (p1 && overlaps(in.mats[i].y, in.mats[i].m, in.x1, in.k)))) { mi355q_set_error("plan_create: an output overlaps a plain operand of the same stage"); return MI355Q_ERR_SHAPE; }
        }
        const bool paired = in.y_kind == MI355Q_Y_UNARY_MUL;
        if (paired && (in.n_mats != 2 || in.mats[0].type != in.mats[1].type || in.mats[0].m != in.mats[1].m ||
                       (in.y_unary != MI355Q_UNARY_SILU && in.y_unary != MI355Q_UNARY_RELU && in.y_unary != MI355Q_UNARY_SIGMOID))) { mi355q_set_error("plan_create: Y_UNARY_MUL needs two matrices of one type and size and SILU / RELU / SIGMOID"); return MI355Q_ERR_UNSUPPORTED; }
        if (in.y_kind != MI355Q_Y_ROWS && !paired) { mi355q_set_error("plan_create: unknown y_kind"); return MI355Q_ERR_UNSUPPORTED; }
        VecSrc x0, x1;
        if (!resolve(in.x, in.k, in.x_id, x0) || !resolve(in.x1, in.k, in.x1_id, x1)) { mi355q_set_error("plan_create: an activation operand straddles an earlier output"); return MI355Q_ERR_SHAPE; }
        // the matrices of a stage share ONE quantized image of the activations: their types must pair with the same activation format
        for (int i = 1; i < in.n_mats; ++i)
            if (gemv_fast_family(in.mats[i].type) != gemv_fast_family(in.mats[0].type)) { mi355q_set_error("plan_create: the matrices of a stage must share the activation format (Q8_K or Q8_0 family)"); return MI355Q_ERR_UNSUPPORTED; }
        bool done[GEMV_MAX_MATS] = { false, false, false, false };
        bool first = true;
        const size_t first_sub = v.size();
        for (int i = 0; i < in.n_mats; ++i) {
            if (done[i]) continue;
            const int type = in.mats[i].type;
            const int fam = gemv_fast_family(type);
            if (fam < 0 || !(tbit(type) & SET_ANY) || !mi355q_weights_are_planar(type, in.k)) { mi355q_set_error("plan_create: weight type / k has no planar streaming kernel in the plan"); return MI355Q_ERR_UNSUPPORTED; }
            PlanStage p = {};
            p.kind = PLAN_K_GEMV; p.tag_off = (unsigned) v.size() + 1;
            int64_t rows = 0; int n = 0;
            const size_t sub_base = gran_count;
            p.yg = (Granule *) (uintptr_t) (sub_base + 1);
            for (int j = i; j < in.n_mats; ++j) {
                if (done[j] || in.mats[j].type != type) continue;
                done[j] = true;
                const mi355q_mat & m = in.mats[j];
                if (!m.w || !m.y || m.m < 0) { mi355q_set_error("plan_create: null matrix pointer"); return MI355Q_ERR_SHAPE; }
                if (((uintptr_t) m.w | (uintptr_t) m.w_stride) & 15) { mi355q_set_error("plan_create: planar rows must be 16-byte aligned"); return MI355Q_ERR_ALIGN; }
                if (m.w_stride < mi355q_row_size(type, in.k)) { mi355q_set_error("plan_create: w_stride smaller than a row"); return MI355Q_ERR_SHAPE; }
                p.w[n] = (const uint8_t *) m.w; p.y[n] = m.y; p.w_stride[n] = m.w_stride; p.row_begin[n] = (int) rows;
                if (!paired || n == 0) {   // the stage's granules form one block indexed by concatenated row: matrix n starts at gran_count + rows
                    OutRange r = { m.y, m.m, sub_base + (size_t) rows, p.tag_off, in.y_id[j] };
                    outs.push_back(r);
                }
                rows += m.m; bytes += m.m * mi355q_row_size(type, in.k); ++n;
            }
            if (rows > 0x7FFFFFF0) { mi355q_set_error("plan_create: too many rows"); return MI355Q_ERR_UNSUPPORTED; }
            gran_count += (size_t) ((rows + 1) & ~(int64_t) 1);
            for (int j = n; j < GEMV_MAX_MATS; ++j) p.row_begin[j] = 0x7FFFFFFF;
            p.total_rows = (int) rows; p.n_mats = n; p.type = type; p.k = (int) in.k;
            p.x0 = x0; p.x1 = x1; p.x_kind = in.x_kind; p.x_unary = in.x_unary; p.eps = in.eps; p.norm_w = in.norm_w;
            // a fresh activation image at the head of a stage; sub-stages of other weight types continue on the same image
            const bool reuse = !first;
            p.flags = (reuse ? 0 : PLAN_F_NEW_X) | ((in.flags & MI355Q_STAGE_NO_PLAIN) ? 0 : PLAN_F_PLAIN_Y);
            if (first && in.x_kind == MI355Q_X_NORM && in.x1 && in.sum_out) {
                p.flags |= PLAN_F_SUM | ((in.flags & MI355Q_STAGE_NO_PLAIN) ? 0 : PLAN_F_SUM_PLAIN);
                p.sum_plain = in.sum_out;
                p.sum_gran = (Granule *) (uintptr_t) (new_out(in.sum_out, in.k, p.tag_off, in.sum_id) + 1);
            }
            if (first) { p.x_out = in.x_out; if (in.x_out) { OutRange xr = { in.x_out, in.k, (size_t) -1, 0, -1 }; outs.push_back(xr); } }
            {   // one plain vector of whole 256-element spans: the direct gather + quantize form (MI355Q_PLAN_DIRECT=0 turns it off: A/B measurements)
                const bool no_direct = getenv("MI355Q_PLAN_DIRECT") && atoi(getenv("MI355Q_PLAN_DIRECT")) == 0;
                if (first && !no_direct && in.x_kind == MI355Q_X_PLAIN && !in.x1 && in.k % 256 == 0) p.flags |= PLAN_F_DIRECT;
            }
            if (paired) { rows = in.mats[0].m; p.total_rows = (int) rows; p.flags |= PLAN_F_PAIRED; p.x_unary |= in.y_unary << 8; }   // per-workgroup PAIRS; the granule block holds m elements
            int64_t rpw = (rows + n_cu - 1) / n_cu; if (rpw < 1) rpw = 1;
            p.rows_per_wg = (int) rpw;
            p.wg_base = 0; p.wg_count = n_cu; p.group = 1; p.pub_wg = (int) (v.size() % (size_t) n_cu);
            p.prime = plan_depth(type);
            v.push_back(p); attn_of.push_back(-1);
            set |= tbit(type);
            const size_t colb = ((size_t) lds_col_bytes(fam, (int) in.k) + 15) & ~(size_t) 15;
            if (colb > lds_max) lds_max = colb;
            if ((size_t) in.k * 4 + 64 > stg_max) stg_max = (size_t) in.k * 4 + 64;
            first = false;
        }
        {   // sub-stages of different weight types -> one GROUP on disjoint workgroup ranges, sized by weight bytes (MI355Q_PLAN_GROUPS=0: one after the other)
            const bool no_groups = getenv("MI355Q_PLAN_GROUPS") && atoi(getenv("MI355Q_PLAN_GROUPS")) == 0;
            const size_t n_sub = v.size() - first_sub;
            if (n_sub > 1 && !no_groups && (int) n_sub <= n_cu) {
                double tot = 0; std::vector<double> wb(n_sub);
                for (size_t m = 0; m < n_sub; ++m) { wb[m] = (double) v[first_sub + m].total_rows * (double) mi355q_row_size(v[first_sub + m].type, in.k); tot += wb[m]; }
                int base = 0;
                for (size_t m = 0; m < n_sub; ++m) {
                    PlanStage & q = v[first_sub + m];
                    int cnt = m + 1 == n_sub ? n_cu - base : (int) (n_cu * wb[m] / tot + 0.5);
                    const int left = (int) (n_sub - m - 1);
                    if (cnt < 1) cnt = 1;
                    if (cnt > n_cu - base - left) cnt = n_cu - base - left;
                    q.wg_base = base; q.wg_count = cnt; q.group = m == 0 ? (int) n_sub : 1;
                    q.rows_per_wg = (q.total_rows + cnt - 1) / cnt; if (q.rows_per_wg < 1) q.rows_per_wg = 1;
                    q.flags |= PLAN_F_NEW_X | (v[first_sub].flags & PLAN_F_DIRECT);
                    q.pub_wg = base + (int) ((first_sub + m) % (size_t) cnt);
                    base += cnt;
                }
            }
        }
    }
    if (set == 0) { mi355q_set_error("plan_create: a plan needs at least one GEMV stage"); return MI355Q_ERR_UNSUPPORTED; }
    {   // the stage that consumes an attention output (wo) runs on the workgroups that did NOT run the attention (MI355Q_PLAN_WO_SKIP=0: on all)
        const bool no_skip = getenv("MI355Q_PLAN_WO_SKIP") && atoi(getenv("MI355Q_PLAN_WO_SKIP")) == 0;
        for (size_t si = 1; si < v.size() && !no_skip; ++si) {
            PlanStage & p = v[si];
            if (p.kind != PLAN_K_GEMV || p.group != 1 || p.wg_count != n_cu || (p.flags & PLAN_F_PAIRED) || v[si - 1].kind == PLAN_K_GEMV) continue;
            size_t ai = si - 1; while (ai > 0 && v[ai].kind == PLAN_K_COMBINE) --ai;
            if (v[ai].kind != PLAN_K_ATTN) continue;
            const AttnStage & A = va[attn_of[ai]];
            const int n_attn = A.n_head * A.n_split;
            if (n_attn > n_cu / 2) continue;
            p.wg_base = n_attn; p.wg_count = n_cu - n_attn;
            p.rows_per_wg = (p.total_rows + p.wg_count - 1) / p.wg_count; if (p.rows_per_wg < 1) p.rows_per_wg = 1;
            p.pub_wg = p.wg_base + (int) (si % (size_t) p.wg_count);
        }
    }
    {   // attention stages whose K / V window (2 x per x hd f16) fits beside everything else run from LDS (plan_attn_lds)
        const size_t fixed = lds_max + 64 + 8 * GEMV_WAVES + 1024, cap = 160 * 1024 - 64;
        const char * e = getenv("MI355Q_PLAN_KV_LDS"), * e2 = getenv("MI355Q_PLAN_FA_EXACT");
        for (size_t si = 0; si < v.size(); ++si) {
            if (v[si].kind != PLAN_K_ATTN) continue;
            AttnStage & A = va[attn_of[si]];
            const int per_pad = (A.per + 31) & ~31;
            const size_t need = (size_t) 50 * A.hd + 64 + (size_t) 8 * per_pad + (size_t) 4 * per_pad * A.hd + (size_t) 16 * per_pad + 64;     // (plan_attn_lds: 12.5 hd + 16 + 2 per floats, K rows of hd + 8 f16, V)
            const bool aligned = (A.k_nb_pos & 15) == 0 && (A.k_nb_head & 15) == 0 && ((uintptr_t) A.k_cache & 15) == 0 && ((uintptr_t) A.v_cache & 15) == 0 && (A.v_nb_head & 15) == 0 &&
                                 (A.v_nb_dim == 2 ? (A.v_nb_pos & 15) == 0 : (A.v_nb_dim & 15) == 0 && A.v_nb_pos == 2);
            int p2 = 64; while (p2 < A.per) p2 <<= 1;                          // (step 2's partial results of one part: 64 << lg floats inside 12 hd)
            A.kv_lds = (!e || atoi(e) != 0) && aligned && A.per <= 320 && p2 <= 12 * A.hd && fixed + ((std::max(need, stg_max) + 15) & ~(size_t) 15) <= cap;
            A.fa_seq = A.kv_lds && A.v_nb_dim == 2 && A.n_split == 1 && (!e2 || atoi(e2) != 0);
            A.entry_barrier = si == 0 || v[si - 1].kind != PLAN_K_GEMV;
            if (A.kv_lds) { v[si].flags |= PLAN_F_ATTN_LDS; if (need > stg_max) stg_max = need; }
            if (getenv("MI355Q_PLAN_VERBOSE")) fprintf(stderr, "mi355q plan: attention stage %zu: n_kv %d, %d split(s) of %d positions, window %s LDS%s (%zu bytes, %zu fixed)\n", si, A.n_kv, A.n_split, A.per, A.kv_lds ? "in" : "not in", A.fa_seq ? ", sequential f16 accumulation" : "", need, fixed);
        }
    }
    const size_t lds_total = lds_max + 64 + 8 * GEMV_WAVES + 1024 + ((stg_max + 15) & ~(size_t) 15);
    if (lds_total > 160 * 1024 - 64) { mi355q_set_error("plan_create: k / attention window too large for the LDS staging area"); return MI355Q_ERR_UNSUPPORTED; }
    if (const char * e = getenv("MI355Q_PLAN_PRIME")) { const int pr = atoi(e); for (auto & p : v) if (p.kind == PLAN_K_GEMV && pr >= 0 && pr < p.prime) p.prime = pr; }

    Plan * pl = new Plan();
    pl->device = dev; pl->n_cu = n_cu; pl->n_stages = (int) v.size(); pl->set = set; pl->lds_image = lds_max; pl->lds_total = lds_total; pl->weight_bytes = bytes;
    pl->even = (flags & MI355Q_FLAG_ROUND_EVEN) ? 1 : 0;
    pl->gran_count = gran_count + 2;
    pl->timeout_ticks = 100ull * 1000 * 20;                    // 20 ms of the 100 MHz real-time counter per poll
    if (const char * e = getenv("MI355Q_PLAN_TIMEOUT_MS")) pl->timeout_ticks = 100ull * 1000 * (unsigned long long) atoll(e);
    const void * kern = plan_kernel(set);
    if (!kern) { delete pl; mi355q_set_error("plan_create: this mix of weight types has no kernel instantiation"); return MI355Q_ERR_UNSUPPORTED; }
    int per_cu = 0;
    static thread_local struct { const void * kern; size_t lds; int dev; } fits = { nullptr, 0, -1 };   // (the last answer: the next KV window's plan asks the same question)
    if (fits.kern == kern && fits.lds >= lds_total && fits.dev == dev) per_cu = 1;
    else if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, GEMV_THREADS, lds_total) != hipSuccess || per_cu < 1) {
        delete pl; mi355q_set_error("plan_create: persistent kernel does not fit a CU"); return MI355Q_ERR_HIP;
    }
    fits = { kern, lds_total, dev };
    pl->grid = n_cu;                                           // one workgroup per CU, all co-resident (checked again by the cooperative launch)
    // ONE device block -- [sync words | stage descriptors | attention descriptors | granules] -- and one upload of the first three parts: a plan is built
    // per KV window while a generation runs (backend/decode-plan.inc), and every hipMalloc / blocking hipMemcpy of the four-block form cost the token it fell on
    auto up256 = [](size_t n) { return (n + 255) & ~(size_t) 255; };
    const size_t off_stages = up256(PLAN_SYNC_WORDS * sizeof(unsigned)), off_attn = off_stages + up256(v.size() * sizeof(PlanStage));
    const size_t off_gran = off_attn + up256(va.size() * sizeof(AttnStage)), block = off_gran + pl->gran_count * sizeof(Granule);
    const auto t_alloc0 = std::chrono::steady_clock::now();
    bool ok = plan_block_take(dev, block, &pl->d_block, &pl->block_bytes) || (pl->block_bytes = block, hipMalloc((void **) &pl->d_block, block) == hipSuccess);
    const auto t_alloc1 = std::chrono::steady_clock::now();
    if (ok) {
        pl->d_sync = (unsigned *) pl->d_block; pl->d_stages = (PlanStage *) (pl->d_block + off_stages); pl->d_gran = (Granule *) (pl->d_block + off_gran);
        pl->d_attn = va.empty() ? nullptr : (AttnStage *) (pl->d_block + off_attn);
        // patch the (offset + 1) placeholders into device pointers
        auto fix = [&](const Granule * p) { return p ? pl->d_gran + ((uintptr_t) p - 1) : nullptr; };
        auto fixs = [&](VecSrc & s) { s.gran = fix(s.gran); };
        for (auto & A : va) { fixs(A.q); fixs(A.k); fixs(A.v); A.part = (Granule *) fix(A.part); A.out_gran = (Granule *) fix(A.out_gran); }
        for (size_t i = 0; i < v.size(); ++i) {
            PlanStage & p = v[i];
            fixs(p.x0); fixs(p.x1); p.sum_gran = (Granule *) fix(p.sum_gran);
            p.yg = (Granule *) fix(p.yg);
            p.attn = attn_of[i] >= 0 ? pl->d_attn + attn_of[i] : nullptr;
            size_t nx = i + 1;                                  // (for the members of a group: the stage behind the group)
            if (v[i].kind == PLAN_K_GEMV && v[i].group == 1 && v[i].wg_count < n_cu) while (nx < v.size() && v[nx].kind == PLAN_K_GEMV && v[nx].group == 1 && v[nx].wg_count < n_cu && v[nx].wg_base > v[i].wg_base) ++nx;
            if (v[i].kind == PLAN_K_GEMV && v[i].group > 1) nx = i + (size_t) v[i].group;
            p.next_attn = nx < v.size() && attn_of[nx] >= 0 ? pl->d_attn + attn_of[nx] : nullptr;
        }
        std::vector<unsigned char> image(off_gran, 0);         // (sync words start at zero)
        memcpy(image.data() + off_stages, v.data(), v.size() * sizeof(PlanStage));
        if (!va.empty()) memcpy(image.data() + off_attn, va.data(), va.size() * sizeof(AttnStage));
        ok = hipMemset(pl->d_gran, 0, pl->gran_count * sizeof(Granule)) == hipSuccess &&          // (null stream: ordered before the blocking copy below returns)
             hipMemcpy(pl->d_block, image.data(), off_gran, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (getenv("MI355Q_PLAN_VERBOSE")) {
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "mi355q plan: %zu stages, device block %zu bytes; host time: descriptors %.0f us, device block %.0f us, upload + granule reset %.0f us\n",
                v.size(), block, us(t_create0, t_alloc0), us(t_alloc0, t_alloc1), us(t_alloc1, std::chrono::steady_clock::now()));
    }
    if (!ok) {
        if (pl->d_block) (void) hipFree(pl->d_block);
        delete pl; mi355q_set_error("plan_create: device allocation failed"); return MI355Q_ERR_HIP;
    }
    *out = (mi355q_plan *) pl;
    return MI355Q_OK;
}

int mi355q_regs_plan_run(mi355q_plan * plan, void * stream) {
    Plan * pl = (Plan *) plan;
    if (!pl) { mi355q_set_error("plan_run: null plan"); return MI355Q_ERR_SHAPE; }
    // tags = epoch + stage index + 1, epoch = run count * (stages + 1): never 0, never repeated until the 32-bit epoch wraps;
    // before it does, the granules are zeroed (on the launch stream) and the count starts over
    const unsigned long long span = (unsigned long long) pl->n_stages + 1;
    if ((pl->runs + 2) * span >= 0xFFFFFFFFull) {
        if (hipMemsetAsync(pl->d_gran, 0, pl->gran_count * sizeof(Granule), (hipStream_t) stream) != hipSuccess) { mi355q_set_error("plan_run: granule reset failed"); return MI355Q_ERR_HIP; }
        pl->runs = 0;
    }
    unsigned epoch = (unsigned) (pl->runs * span);
    ++pl->runs;
    const PlanStage * st = pl->d_stages; int n = pl->n_stages; unsigned * sync = pl->d_sync; int even = pl->even;
    int image = (int) pl->lds_image;
    unsigned long long timeout = pl->timeout_ticks;
    void * args[] = { (void *) &st, (void *) &n, (void *) &sync, (void *) &even, (void *) &timeout, (void *) &image, (void *) &epoch };
    // A PLAIN launch of one workgroup per CU (checked against the occupancy query at creation).  hipLaunchCooperativeKernel gives the same
    // residency and only adds a launch-time check of the grid size -- at +15-19 us of host time per launch, through a second (cooperative) HSA
    // queue whose teardown at process exit crashed inside the HSA runtime under rocprofv3 (MI355X_MICROARCH.md, coop-launch row).
    // Every poll is bounded, so a workgroup that found no CU (another persistent kernel holding them) ends in status() == 1, not in a hang.
    const hipError_t rc = hipLaunchKernel(plan_kernel(pl->set), dim3((unsigned) pl->grid), dim3(GEMV_THREADS), args, pl->lds_total, (hipStream_t) stream);
    if (rc != hipSuccess) { mi355q_set_error(hipGetErrorString(rc)); return MI355Q_ERR_HIP; }
    return MI355Q_OK;
}

/* test hook: the number of runs the plan believes it has made (the epoch source); setting it close to the wrap exercises the granule reset */
int mi355q_regs_plan_debug_set_runs(mi355q_plan * plan, unsigned long long runs) {
    Plan * pl = (Plan *) plan;
    if (!pl) return MI355Q_ERR_SHAPE;
    pl->runs = runs;
    return MI355Q_OK;
}
int mi355q_regs_plan_debug_words(mi355q_plan * plan, unsigned * out32) {
    Plan * pl = (Plan *) plan;
    if (!pl || !out32) return MI355Q_ERR_SHAPE;
    return hipMemcpy(out32, pl->d_sync, PLAN_SYNC_WORDS * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess ? MI355Q_OK : MI355Q_ERR_HIP;
}

/* 0 = healthy; 1 = a poll timed out (the plan is dead: destroy it).  Synchronizes with the device. */
int mi355q_regs_plan_status(mi355q_plan * plan) {
    Plan * pl = (Plan *) plan;
    if (!pl) return MI355Q_ERR_SHAPE;
    unsigned h[PLAN_SYNC_WORDS];
    if (hipMemcpy(h, pl->d_sync, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return MI355Q_ERR_HIP;
    return h[PLAN_SYNC_ABORT] ? 1 : 0;
}

int mi355q_regs_plan_status_async(mi355q_plan * plan, unsigned * host_flag, void * stream) {
    Plan * pl = (Plan *) plan;
    if (!pl || !host_flag) return MI355Q_ERR_SHAPE;
    if (hipMemcpyAsync(host_flag, pl->d_sync + PLAN_SYNC_ABORT, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t) stream) != hipSuccess) return MI355Q_ERR_HIP;
    return MI355Q_OK;
}

int64_t mi355q_regs_plan_weight_bytes(const mi355q_plan * plan) { return plan ? ((const Plan *) plan)->weight_bytes : 0; }
int     mi355q_regs_plan_launch_stages(const mi355q_plan * plan) { return plan ? ((const Plan *) plan)->n_stages : 0; }

int mi355q_regs_plan_destroy(mi355q_plan * plan) {
    Plan * pl = (Plan *) plan;
    if (!pl) return MI355Q_OK;
    plan_block_give(pl->device, pl->d_block, pl->block_bytes);
    delete pl;
    return MI355Q_OK;
}

} // extern "C"
