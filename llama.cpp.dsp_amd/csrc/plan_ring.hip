// plan.hip -- a token-generation step as ONE persistent launch (MI355X-first replacement for the reference's "one kernel
// per node + CUDA graph" decode loop, ggml-cuda.cu:2470-2781): the N = 1 quantized matmuls AND the glue between them.
//
// A PLAN is an ordered list of STAGES, walked by every workgroup of one launch of #CU workgroups x 16 waves (one per CU, all resident):
//
//   GEMV    up to 4 planar weight matrices of one type against one activation vector x.  The prologue forms x in LDS --
//           x0 | rms_norm(x0 + x1) * w | unary(x0) * x1, quantized exactly as the CPU does (act_quant.cuh) -- then the
//           workgroup multiplies its rows; arithmetic per row is gemv_fast.hip's (gemv_stream.cuh is shared): bit-identical.
//   ATTN    rope(q), rope(k), this token's K / V cache stores and causal attention of the one token over the f16 cache,
//           one workgroup per (head, KV split); COMBINE merges the splits of a head (log-sum-exp).
//
// Round 3: the WEIGHT STREAM IS DECOUPLED FROM THE DEPENDENCY CHAIN.  Wave 15 of every workgroup is a LOADER: it walks the
// stage list on its own and copies the workgroup's weight rows of every GEMV stage, in stage order, from HBM into a ring of
// 1-KiB pages in LDS (~110-140 KiB per CU) with LDS-DMA (global_load_lds_dwordx4, saddr form, nt, up to four pages per M0
// write, 32 instructions in flight) -- weights depend on nothing the launch computes, so it never waits for an activation,
// only for ring space.  Waves 0..14 are CONSUMERS: they poll the stage's operands, build the quantized image, then take the
// rows that are (or soon will be) resident in the ring: wait for "landed" to cover the row, ds_read_b128 it, same consumers
// as gemv_fast.hip, publish.  While the consumers sit in a hand-off (4-6 us in round 2, during which HBM idled) the loader
// fills the ring with the next stages' rows, ~30 MB chip-wide; the round-2 kernel held 8 KiB of registers per wave and could
// not run ahead of the stage it was in (profiles/round2_plan_timeline.md; the measured data path alone: tools/micro/ring_stream.hip,
// 6.5-6.8 TB/s against 6.9 TB/s for a plain 16-wave stream).
//
//   * handshake in LDS words, no hardware barrier after the first one: the loader publishes the count of pages that have
//     LANDED (vmcnt retires in order: after s_waitcnt vmcnt(32) everything but the last 32 pages is in LDS); every consumer
//     wave keeps a HEAD word = the first page it still needs, the loader fills while fill - min(head) < ring pages.  The
//     loader takes no part in s_barrier, so the consumers synchronize among themselves through an LDS counter.
//   * DATAFLOW between stages, not barriers: a value produced during the run is published element by element as an 8-byte
//     {f32 value, u32 tag} GRANULE (one naturally aligned agent-scope store; tag = launch epoch + producing stage) in a
//     plan-private buffer; a consumer polls exactly the granules it needs with agent-scope (sc1) loads until every tag
//     matches (MI355X_MICROARCH.md, hand-off price list: the data-tagged granule is the cheapest cross-CU edge).
//   * no re-arming: tags grow monotonically over launches (the host re-zeroes the granules before the 32-bit epoch wraps).
//   * every wait has a wall-clock bound (s_memrealtime): on timeout the workgroup raises the plan's sticky abort flag and
//     leaves, so the grid always drains (one workgroup per CU: all are resident unless another persistent kernel holds CUs).
//
// Bound: HBM read of W.  Algorithmic bytes per launch = sum over stages, matrices of m * row_size(type, k) (+ the KV cache
// window of ATTN stages: 2 * n_kv * n_head_kv * head_dim * 2 bytes).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <vector>

#include "gemv_stream.cuh"

namespace mi355q { namespace ring {

#ifndef MI355Q_PLAN_NC
#define MI355Q_PLAN_NC 15
#endif
constexpr int PLAN_NC   = MI355Q_PLAN_NC;         // consumer waves (<= 15); wave PLAN_NC is the loader, the waves behind it leave at once
constexpr int PLAN_CT   = PLAN_NC * WAVE;         // consumer threads
#ifndef MI355Q_PLAN_DMA
#define MI355Q_PLAN_DMA 32
#endif
constexpr int PLAN_DMA  = MI355Q_PLAN_DMA;                     // LDS-DMA instructions in flight (ring_stream.hip: 32 beats 48: what is in flight is not yet usable)
constexpr int PLAN_MAXS = PLAN_NC >= 12 ? 8 : 16;                      // 256-element spans of the activation vector per consumer wave: k <= 8 * 15 * 256
enum { PLAN_F_NEW_X = 2, PLAN_F_PLAIN_Y = 4, PLAN_F_SUM = 8, PLAN_F_SUM_PLAIN = 16, PLAN_F_PAIRED = 32 };
enum { PLAN_K_GEMV = 0, PLAN_K_ATTN = 1, PLAN_K_COMBINE = 2 };
enum { PLAN_SYNC_ABORT = 0, PLAN_SYNC_WORDS = 32 };
// LDS control block (byte offsets): health flag, landed page count, consumer barrier counter, per-wave head words, f64 partials of the norm
enum { CB_OK = 0, CB_LANDED = 4, CB_BAR = 8, CB_HEAD = 64, CB_PART = 128,
       CB_GEN = 256,                              // row-slot generations: two sets of 32 words (GEMV stages alternate)
       CB_CNT = 512,                              // row-slot arrival counters (32 words): the wave whose arrival completes a row closes it
       CB_PCNT = 768,                             // PAIRED stages: arrivals of a pair's two dot products (64 words)
       CB_PDV = 1024,                             // PAIRED stages: the two dot products of a pair (2 x 64 words)
       CB_STATE = 1536,                           // (diagnostic build) one word per wave: where it is
       CB_DESC = 1792,                            // the loader's copies of the next stage descriptors (16 slots of 256 bytes)
       CB_BYTES = 1792 + 16 * 256 };
constexpr int PLAN_TERM_STEPS = 64;               // steps whose per-lane terms fit the term buffer (64 x 256 bytes behind the control block)
constexpr int PLAN_TERM_BYTES = PLAN_TERM_STEPS * 256;

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
    float scale;
    // rope (ggml_rope_cache_init; see ops_glue.hip k_rope)
    int n_dims, neox; float freq_scale, ext_factor, attn_factor, theta_scale, corr0, corr1;
};

struct alignas(64) PlanStage {
    // -- line 0: everything the LOADER needs (one 64-byte scalar load) --
    const uint8_t * w[GEMV_MAX_MATS];             // rows of a matrix are contiguous (stride == row_bytes: checked at creation)
    int             row_begin[GEMV_MAX_MATS];     // first concatenated row of each matrix (unused entries: INT_MAX)
    int             total_rows, rows_per_wg, row_bytes;
    int             kf;                           // kind | flags << 8 | glog << 24
    // -- the consumers' part --
    float *         y[GEMV_MAX_MATS];
    Granule *       yg;                           // granules of the stage's outputs, indexed by CONCATENATED row
    int             k, n_mats, type, flags;
    int             kind, glog, x_kind, x_unary;
    VecSrc          x0, x1;
    const float *   norm_w;
    Granule *       sum_gran; float * sum_plain; float * x_out;
    const AttnStage * attn;
    float           eps; unsigned tag_off;
};
static_assert(sizeof(PlanStage) == 256, "PlanStage is four 64-byte lines");
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

#ifdef MI355Q_STAMPS
// Diagnostic build only (libmi355q_dbg.so): waves 0 and 14 of every workgroup record 100 MHz wall-clock stamps per stage:
// g_plan_stamps[((stage*grid + wg)*2 + (wave==14))*8 + i].  The product library contains none of this.
__device__ unsigned long long * g_plan_stamps = nullptr;
__device__ int g_plan_stamp_stages = 0;
#define PLAN_STAMP(i) do { if (g_plan_stamps && lane == 0 && (wave == 0 || wave == PLAN_NC - 1) && c.stage < g_plan_stamp_stages) \
    g_plan_stamps[(((size_t) c.stage * c.grid + blockIdx.x) * 2 + (wave ? 1 : 0)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// per-phase cycle sums of the step loop (wave 0 of every workgroup): g_plan_stamps + n_stages * grid * 16 + (stage * grid + wg) * 8 + i
// where every consumer wave is (dumped into the plan's sync words 8..23 by the wave that gives up): CB_PART + 128 + 4 * wave
#define PLAN_STATE(code) do { if (lane == 0) cb_st(c.cb + CB_STATE + 4u * (threadIdx.x >> 6), ((unsigned) c.stage << 24) | (unsigned) (code)); } while (0)
#define PLAN_PROF_DECL unsigned long long prof_t = __builtin_amdgcn_s_memtime(), prof_acc[6] = { 0, 0, 0, 0, 0, 0 }
#define PLAN_PROF(i) do { const unsigned long long prof_n = __builtin_amdgcn_s_memtime(); prof_acc[i] += prof_n - prof_t; prof_t = prof_n; } while (0)
#define PLAN_PROF_FLUSH do { if (g_plan_stamps && lane == 0 && c.stage < g_plan_stamp_stages) { \
    if (wave == 0) for (int pi = 0; pi < 5; ++pi) g_plan_stamps[(size_t) g_plan_stamp_stages * c.grid * 16 + ((size_t) c.stage * c.grid + blockIdx.x) * 8 + pi] = prof_acc[pi]; \
    unsigned long long * pw = g_plan_stamps + (size_t) g_plan_stamp_stages * c.grid * 24 + (((size_t) c.stage * c.grid + blockIdx.x) * 16 + wave) * 4; \
    pw[0] = prof_acc[2]; pw[1] = prof_acc[3]; pw[2] = __builtin_amdgcn_s_memrealtime(); pw[3] = prof_acc[0] + prof_acc[1] + prof_acc[4]; } } while (0)
#else
#define PLAN_STAMP(i) do { } while (0)
#define PLAN_STATE(code) do { } while (0)
#define PLAN_PROF_DECL do { } while (0)
#define PLAN_PROF(i) do { } while (0)
#define PLAN_PROF_FLUSH do { } while (0)
#endif

// ---- wave-uniform values, said so: the compiler keeps loop-carried state in SGPRs only while it can prove every branch around it uniform;
// values read back from LDS or from a lane are uniform by construction here, and readfirstlane tells it
__device__ __forceinline__ int      ufl(int x)       { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ unsigned uflu(unsigned x) { return (unsigned) __builtin_amdgcn_readfirstlane((int) x); }
template <typename P> __device__ __forceinline__ P uniform_ptr(P p) {
    const unsigned long long v = (unsigned long long) (uintptr_t) p;
    const unsigned lo = uflu((unsigned) v), hi = uflu((unsigned) (v >> 32));
    return (P) (uintptr_t) (((unsigned long long) hi << 32) | lo);
}

// ---- the control words live in LDS and are touched with explicit DS instructions on their LDS byte address (a `volatile` generic pointer
// compiles to flat accesses with sc0 sc1 and a vmcnt(0) wait behind each, which would drain the loader's DMA queue at every publish)
__device__ __forceinline__ void     cb_st(unsigned addr, unsigned v)  { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void     cb_add(unsigned addr, unsigned v) { asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
// lane 0 counts an arrival on an LDS word that wraps to 0 after `limit` (ds_inc: old >= limit ? 0 : old + 1); every lane gets the OLD value
__device__ __forceinline__ unsigned cb_arrive(unsigned addr, unsigned limit, int lane) {
    unsigned old = 0;
    if (lane == 0) asm volatile("ds_inc_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(addr), "v"(limit) : "memory");
    return (unsigned) __builtin_amdgcn_readlane((int) old, 0);
}
// lane 0 adds n to an LDS word; every lane gets the OLD value
__device__ __forceinline__ unsigned cb_arrive_n(unsigned addr, unsigned n, int lane) {
    unsigned old = 0;
    if (lane == 0) asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(addr), "v"(n) : "memory");
    return (unsigned) __builtin_amdgcn_readlane((int) old, 0);
}
__device__ __forceinline__ unsigned cb_ld(unsigned addr) { unsigned v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory"); return uflu(v); }

// what a wave needs to know about the launch
struct Ctx {
    uint8_t * lds; unsigned cb, ctl_off;          // LDS base (generic pointer to the kernel's LDS symbol) and the control block's LDS byte address
    unsigned ring_off, ring_bytes, ring_magic, np, np_magic;   // the ring: byte offset inside lds, size, floor(2^32 / size), pages, floor(2^32 / pages)
    unsigned * sync; unsigned long long timeout;
    unsigned grid, epoch; int even, stage;
};

// x mod ring_bytes for a stream byte position (< 2^32): q = mulhi(x, floor(2^32 / d)) is the quotient or one less
__device__ __forceinline__ unsigned ring_pos(const Ctx & c, unsigned x) {
    unsigned r = x - __umulhi(x, c.ring_magic) * c.ring_bytes;
    if (r >= c.ring_bytes) r -= c.ring_bytes;
    return r;
}

struct PollCtx { unsigned * sync; unsigned long long timeout, t0; int stage; };
// false = give up (timeout or the plan's abort flag is up)
__device__ __forceinline__ bool poll_backoff(const PollCtx & pc, unsigned & spins, int lane) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 63u) == 0u) {
        unsigned ab = 0;
        if (lane == 0) ab = __hip_atomic_load(pc.sync + PLAN_SYNC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ab = uflu(ab);
        if (ab != 0u) return false;
        if (__builtin_amdgcn_s_memrealtime() - pc.t0 > pc.timeout) {
            if (lane == 0 && __hip_atomic_exchange(pc.sync + PLAN_SYNC_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) { pc.sync[1] = 7u /* a producer poll */; pc.sync[2] = (unsigned) pc.stage; pc.sync[3] = blockIdx.x; pc.sync[4] = threadIdx.x >> 6; }
            return false;
        }
    }
    return true;
}
// one more round of a wait on an LDS word of this workgroup (landed pages, barrier arrivals, slot generations): false = the workgroup is dead (another
// wave gave up) or the wait has lasted ~2^18 polls (tens of milliseconds): the plan's abort word is raised and everybody leaves
// (`why`, `a`, `b`: what was waited for -- kept in the plan's sync words 1..7 by the first wave that gives up: mi355q_ring_plan_debug_words)
enum { WHY_LANDED = 1, WHY_SLOT = 2, WHY_COUNT = 3, WHY_PAIR = 4, WHY_BARRIER = 5, WHY_RING = 6 };
__device__ __forceinline__ void plan_give_up(const Ctx & c, int lane, unsigned why, unsigned a, unsigned b) {
    if (lane == 0) {
        cb_st(c.cb + CB_OK, 0u);
        if (__hip_atomic_exchange(c.sync + PLAN_SYNC_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            c.sync[1] = why; c.sync[2] = (unsigned) c.stage; c.sync[3] = blockIdx.x; c.sync[4] = threadIdx.x >> 6; c.sync[5] = a; c.sync[6] = b;
            c.sync[7] = cb_ld(c.cb + CB_LANDED);
#ifdef MI355Q_STAMPS
            for (int i = 0; i < 16; ++i) c.sync[8 + i] = cb_ld(c.cb + CB_STATE + 4u * (unsigned) i);
            for (int i = 0; i < 8; ++i) c.sync[24 + i] = cb_ld(c.cb + CB_HEAD + 4u * (unsigned) (2 * i));
#endif
        }
    }
}
__device__ __forceinline__ bool lds_spin(const Ctx & c, unsigned & spins, int lane, unsigned why, unsigned a, unsigned b) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 255u) == 0u) {
        if (cb_ld(c.cb + CB_OK) == 0u) return false;
        if (spins > (1u << 16)) { plan_give_up(c, lane, why, a, b); return false; }
    }
    return true;
}

// barrier among the 15 consumer waves (the loader never joins one): arrive on an LDS counter, spin until all have.  Every consumer wave calls it
// the same number of times.  An arriving wave's earlier LDS writes are complete (lgkmcnt) before its arrival is counted.
// Per-wave progress that changes from stage to stage: kept OUT of Ctx (a Ctx handed to a function by mutable reference makes the compiler forget
// that c.lds is the kernel's LDS symbol, and every LDS access becomes a flat one).
struct Prog { int gemv_idx;         // GEMV stages done (every wave counts the same)
              unsigned base_page;    // stream position (BYTES since the launch began) at which the current stage's rows start
              unsigned bar_target; };  // consumer barrier: arrivals expected at the next cbar()
__device__ __forceinline__ bool cbar(const Ctx & c, Prog & pr, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) cb_add(c.cb + CB_BAR, 1u);
    pr.bar_target += (unsigned) PLAN_NC;
    unsigned spins = 0;
    while ((int) (cb_ld(c.cb + CB_BAR) - pr.bar_target) < 0) if (!lds_spin(c, spins, lane, WHY_BARRIER, pr.bar_target, 0)) return false;
    return cb_ld(c.cb + CB_OK) != 0u;
}

__device__ __forceinline__ void publish(Granule * gp, float v, unsigned tag) {
    __hip_atomic_store(gp, ((Granule) tag << 32) | (Granule) __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- stage geometry: where the rows of a workgroup sit in the stream ----------------------------------------------------------------
// The stream of a workgroup is TIGHTLY PACKED: stream row q of a stage (q = 0, 1, ...) occupies bytes [q * row_bytes, (q + 1) * row_bytes) behind the
// stage's first byte, and a stage begins where the previous one ended -- no alignment anywhere (row_bytes is a multiple of 16), so a consumer finds
// a row with one multiplication.  Which row of which matrix stream row q is:
//   plain stage   concatenated row r_lo + q of the workgroup's range [r_lo, r_hi)
//   PAIRED stage  (y = unary(W0 x) * (W1 x): a pair's two rows must meet) pairs in groups of G = 2^glog: the G rows of matrix 0, then the same G
//                 rows of matrix 1, group after group -- a pair's second row is at most 2 G rows behind its first one, and the loader still copies
//                 runs of G contiguous rows
// The loader copies SEGMENTS (contiguous byte ranges of one matrix) back to back; a ring page may hold the end of one and the start of the next.
__device__ __forceinline__ void stage_rows(StageC st, int & r_lo, int & r_hi) {
    const int rpw = st->rows_per_wg, total = st->total_rows;
    r_lo = (int) blockIdx.x * rpw; r_hi = min(r_lo + rpw, total);
    if (r_lo > r_hi) r_lo = r_hi;
}
__device__ __forceinline__ unsigned stage_stream_bytes(StageC st) {
    int r_lo, r_hi; stage_rows(st, r_lo, r_hi);
    return (unsigned) (r_hi - r_lo) * ((st->flags & PLAN_F_PAIRED) ? 2u : 1u) * (unsigned) st->row_bytes;
}

// ---- the loader ------------------------------------------------------------------------------------------------------------------------
// K consecutive 1-KiB pages with ONE M0 write: the instruction offset advances the global and the LDS address alike.  M0 is written inside the
// statement that uses it (cdna_hip_programming.md 5.7); the source base is wave-uniform (SGPR pair: saddr form, no address VALU at all).
template <int K> __device__ __forceinline__ void dma_pages(const uint8_t * g_in, unsigned voff, unsigned lds_addr_in) {
    const uint8_t * g = uniform_ptr(g_in); const unsigned lds_addr = uflu(lds_addr_in);
    if constexpr (K == 1) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" :: "v"(voff), "s"(g), "s"(lds_addr) : "memory");
    if constexpr (K == 2) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024 nt" :: "v"(voff), "s"(g), "s"(lds_addr) : "memory");
    if constexpr (K == 3) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048 nt" :: "v"(voff), "s"(g), "s"(lds_addr) : "memory");
    if constexpr (K == 4) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072 nt" :: "v"(voff), "s"(g), "s"(lds_addr) : "memory");
}

// pos: stream bytes requested so far (since the launch began); tail: the first page a consumer still needs; landed: pages published as resident
struct Loader { unsigned pos, tail, landed, spins; bool dead;
#ifdef MI355Q_STAMPS
                unsigned long long t_blocked, t_mark, t_vm, t_tail, t_issue; unsigned n_dma, n_iter;
#endif
};

__device__ __forceinline__ void loader_tail(Loader & L, const Ctx & c, int lane) {     // min over the consumers' head words (word 15 stays 0xFFFFFFFF)
    unsigned h;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(h) : "v"(c.cb + CB_HEAD + 4u * (unsigned) (lane & 15)) : "memory");
    h = min(h, (unsigned) dpp_i<0xB1>((int) h)); h = min(h, (unsigned) dpp_i<0x4E>((int) h)); h = min(h, (unsigned) dpp_i<0x141>((int) h)); h = min(h, (unsigned) dpp_i<0x140>((int) h));
    L.tail = uflu(h);
}
__device__ __forceinline__ void loader_publish(Loader & L, const Ctx & c, unsigned landed, int lane) {
    if ((int) (landed - L.landed) > 0) { L.landed = landed; if (lane == 0) cb_st(c.cb + CB_LANDED, landed); }
}
// copy one segment (bytes % 16 == 0) to the stream position where the previous one ended
__device__ __forceinline__ void loader_segment(Loader & L, const Ctx & c, const uint8_t * g_in, unsigned bytes_in, int lane) {
    const uint8_t * g = uniform_ptr(g_in);
    const unsigned bytes = uflu(bytes_in);
    const unsigned voff = 16u * (unsigned) lane;
    unsigned done = 0;
    while (done < bytes && !L.dead) {
        L.pos = uflu(L.pos); L.tail = uflu(L.tail); L.landed = uflu(L.landed); L.spins = uflu(L.spins); done = uflu(done);
        const unsigned page = L.pos >> 10, inpage = L.pos & 1023u, left = bytes - done;
        // pages of the ring this request may use: [page, tail + np)   (a consumer's head may be AHEAD of pos: a wave that waits for rows not yet requested)
        auto room = [&]() { const int used = (int) (page - L.tail); return (int) c.np - (used > 0 ? used : 0); };
        int space = room();
#ifdef MI355Q_STAMPS
        const unsigned long long tt0 = __builtin_amdgcn_s_memtime();
#endif
        if (space < 8) { loader_tail(L, c, lane); space = room(); }
#ifdef MI355Q_STAMPS
        L.t_tail += __builtin_amdgcn_s_memtime() - tt0;
#endif
        if (space <= 0) {
#ifdef MI355Q_STAMPS
            if (L.t_mark == 0) L.t_mark = __builtin_amdgcn_s_memtime();
#endif
            // The ring is full: the consumers have at least np - PLAN_DMA published pages in front of them.  A wave's next unit can lie further ahead
            // than that (15 units on, plus the rest of its row), so what is still in flight must be PUBLISHED or the wave never frees its pages: if the
            // ring is still full at the second look, wait for everything requested and publish it.  (Doing that at the first look made every small
            // refill of a momentarily full ring pay a whole memory latency.)
            if (L.landed != page) {                             // a ladder: each look publishes what has landed meanwhile without waiting long for the rest
                const unsigned lag = page - L.landed;
                if (lag > 24u)      { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); loader_publish(L, c, page - 24u, lane); }
                else if (lag > 16u) { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); loader_publish(L, c, page - 16u, lane); }
                else if (lag > 8u)  { asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  loader_publish(L, c, page - 8u, lane); }
                else                { asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  loader_publish(L, c, page, lane); }
            }
            __builtin_amdgcn_s_sleep(1);
            if ((++L.spins & 1023u) == 0u) {
                if (cb_ld(c.cb + CB_OK) == 0u) L.dead = true;
                else if (L.spins > (1u << 20)) { plan_give_up(c, lane, WHY_RING, page, L.tail); L.dead = true; }     // ~ a fraction of a second of polling
            }
            continue;
        }
#ifdef MI355Q_STAMPS
        if (L.t_mark) { L.t_blocked += __builtin_amdgcn_s_memtime() - L.t_mark; L.t_mark = 0; }
        ++L.n_iter;
#endif
        L.spins = 0;
        unsigned rpage = page - __umulhi(page, c.np_magic) * c.np; if (rpage >= c.np) rpage -= c.np;      // page % np
        const unsigned dst = (unsigned) (size_t) c.lds + c.ring_off + rpage * 1024u;        // (low 32 bits of a shared-memory pointer = its LDS byte address)
#ifdef MI355Q_STAMPS
        const unsigned long long ti0 = __builtin_amdgcn_s_memtime();
#endif
        const uint8_t * src = g + done;
        if (inpage == 0u && left >= 4096u && space >= 4 && c.np - rpage >= 4u) {
            // bulk: up to 16 whole pages per round of bookkeeping, four per M0 write (a lone wave issues an instruction every ~5 cycles: the ~60
            // scalar instructions of a round, paid per 2-3 pages, held the stream at 3.2 TB/s -- rocprofv3: the loader wave 59 % busy issuing)
            const unsigned n4 = uflu(min(min(left >> 12, 4u), min((unsigned) space >> 2, (c.np - rpage) >> 2)));
            for (unsigned j = 0; j < n4; ++j) dma_pages<4>(src + 4096u * j, voff, dst + 4096u * j);
            done += n4 << 12; L.pos += n4 << 12;
#ifdef MI355Q_STAMPS
            L.n_dma += 4u * n4;
#endif
        } else if (inpage != 0u || left < 1024u) {                      // part of a page: lanes [inpage / 16, (inpage + n) / 16) copy src .. src + n
            const unsigned n = min(1024u - inpage, left);
            const unsigned l0 = inpage >> 4, l1 = (inpage + n) >> 4;
            if ((unsigned) lane >= l0 && (unsigned) lane < l1) dma_pages<1>(src - inpage, voff, dst);
            done += n; L.pos += n;
#ifdef MI355Q_STAMPS
            ++L.n_dma;
#endif
        } else {
            unsigned k = min(min(4u, left >> 10), min((unsigned) space, c.np - rpage));
            k = uflu(k);
            if (k == 4u)      dma_pages<4>(src, voff, dst);
            else if (k == 3u) dma_pages<3>(src, voff, dst);
            else if (k == 2u) dma_pages<2>(src, voff, dst);
            else              dma_pages<1>(src, voff, dst);
            done += k << 10; L.pos += k << 10;
#ifdef MI355Q_STAMPS
            L.n_dma += k;
#endif
        }
        // at most PLAN_DMA instructions (each touches one page, pages in rising order, landing in issue order) are still in flight: every page
        // before the last PLAN_DMA completely requested ones has landed
#ifdef MI355Q_STAMPS
        const unsigned long long tv0 = __builtin_amdgcn_s_memtime();
        L.t_issue += tv0 - ti0;
#endif
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(PLAN_DMA) : "memory");
#ifdef MI355Q_STAMPS
        L.t_vm += __builtin_amdgcn_s_memtime() - tv0;
#endif
        if ((L.pos >> 10) > (unsigned) PLAN_DMA) loader_publish(L, c, (L.pos >> 10) - (unsigned) PLAN_DMA, lane);
    }
}

// line 0 of a stage descriptor, in SGPRs
struct LDesc { const uint8_t * w0, * w1, * w2, * w3; const AttnStage * attn; int b1, b2, b3, total_rows, rows_per_wg, row_bytes, kf; };
constexpr int PLAN_DESC_AHEAD = 8, PLAN_DESC_SLOTS = 16;        // descriptors are copied into LDS this many stages before the loader reads them

// Descriptors miss every cache after a token's worth of streaming; a scalar load that goes to HBM costs 1-2 us, and the loader passes ~200 stages per
// token (small stages -- wk | wv, wo -- do not even cover one such miss).  So the loader DMA-copies the descriptor of stage s + 8 (all four lines)
// into a ring of LDS slots while it streams stage s, and reads line 0 from there; the copy also brings the lines into this XCD's L2, where the
// consumers' scalar loads find them a few microseconds later.  Descriptor copies count in vmcnt like every other DMA instruction.
__device__ __forceinline__ void desc_request(const PlanStage * g, int s, const Ctx & c, int lane) {
    if (lane < 16) {
        const uint8_t * src = (const uint8_t *) (g + s) + 16 * lane;
        const unsigned dst = uflu((unsigned) (size_t) c.lds + c.ctl_off + CB_DESC + 256u * (unsigned) (s & (PLAN_DESC_SLOTS - 1)));
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory");
    }
}
__device__ __forceinline__ LDesc desc_read(int s, const Ctx & c, int lane) {
    // lane j < 16 reads dword j of line 0; the four pointers are dword pairs (0,1) .. (6,7), row_begin[1..3] dwords 9..11, then total_rows, rows_per_wg, row_bytes, kf
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(c.cb + CB_DESC + 256u * (unsigned) (s & (PLAN_DESC_SLOTS - 1)) + 4u * (unsigned) (lane & 15)) : "memory");
    auto rl = [&](int j) { return (unsigned) __builtin_amdgcn_readlane((int) v, j); };
    auto p64 = [&](int j) { return (const uint8_t *) (uintptr_t) (((unsigned long long) rl(j + 1) << 32) | rl(j)); };
    LDesc d;
    d.w0 = p64(0); d.w1 = p64(2); d.w2 = p64(4); d.w3 = p64(6);
    d.b1 = (int) rl(9); d.b2 = (int) rl(10); d.b3 = (int) rl(11);
    d.total_rows = (int) rl(12); d.rows_per_wg = (int) rl(13); d.row_bytes = (int) rl(14); d.kf = (int) rl(15);
    d.attn = nullptr;
    return d;
}

static __device__ __forceinline__ void plan_loader(const PlanStage * stages_g, int n_stages, const Ctx & c) {
    const int lane = lane_id();
    __builtin_amdgcn_s_setprio(1);
    Loader L; L.pos = 0; L.tail = 0; L.landed = 0; L.spins = 0; L.dead = false;
#ifdef MI355Q_STAMPS
    L.t_blocked = 0; L.t_mark = 0; L.t_vm = 0; L.t_tail = 0; L.t_issue = 0; L.n_dma = 0; L.n_iter = 0;
#endif
    for (int s = 0; s < n_stages && s < PLAN_DESC_AHEAD; ++s) desc_request(stages_g, s, c, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned since = 0;                                          // DMA pages requested since the descriptor of the stage about to be read was requested
#pragma unroll 1
    for (int s = 0; s < n_stages && !L.dead; ++s) {
        // the copy of this stage's descriptor was requested PLAN_DESC_AHEAD stages ago: it has landed if more than PLAN_DMA requests followed it and
        // the usual wait has been done since; a run of stages in which this workgroup has (almost) no rows needs the explicit wait
        if (s >= PLAN_DESC_AHEAD && (L.pos >> 10) - since <= (unsigned) PLAN_DMA + 1u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const LDesc d = desc_read(s, c, lane);
        if (s + PLAN_DESC_AHEAD < n_stages) { desc_request(stages_g, s + PLAN_DESC_AHEAD, c, lane); }
        // (bookkeeping for the check above: the page count when the descriptor PLAN_DESC_AHEAD - 1 stages ahead of the NEXT one was requested is not kept
        //  per slot; the conservative stand-in is the count at the START of this stage, i.e. every request of the last PLAN_DESC_AHEAD stages but the
        //  current one is ignored -- the explicit wait then happens a little more often than needed, never too rarely)
        since = L.pos >> 10;
        if ((d.kf & 0xFF) != PLAN_K_GEMV) {
            // an ATTN / COMBINE stage: touch its attention descriptor (two cache lines' worth of fields the consumers read first)
            StageC st = (StageC) stages_g + s;
            const __attribute__((address_space(4))) AttnStage * a = (const __attribute__((address_space(4))) AttnStage *) st->attn;
            const int u1 = a->n_head, u2 = a->mask_f16; const float u3 = a->corr1; asm volatile("" :: "s"(u1), "s"(u2), "s"(u3));
            continue;
        }
        const unsigned rb = (unsigned) d.row_bytes;
        int r_lo = (int) blockIdx.x * d.rows_per_wg, r_hi = min(r_lo + d.rows_per_wg, d.total_rows);
        if (r_lo > r_hi) r_lo = r_hi;
        if ((d.kf >> 8) & PLAN_F_PAIRED) {
            const int np = r_hi - r_lo, G = 1 << ((d.kf >> 24) & 0xFF);
#pragma unroll 1
            for (int g0 = 0; g0 < np; g0 += G) {
                const int n_g = min(G, np - g0);
                const size_t off = (size_t) (r_lo + g0) * rb;
                loader_segment(L, c, d.w0 + off, (unsigned) n_g * rb, lane);
                loader_segment(L, c, d.w1 + off, (unsigned) n_g * rb, lane);
            }
        } else {
            const int mb[GEMV_MAX_MATS + 1] = { 0, d.b1, d.b2, d.b3, 0x7FFFFFFF };
            const uint8_t * ws[GEMV_MAX_MATS] = { d.w0, d.w1, d.w2, d.w3 };
#pragma unroll
            for (int i = 0; i < GEMV_MAX_MATS; ++i) {
                const int lo = max(r_lo, mb[i]), hi = min(r_hi, mb[i + 1]);
                if (hi > lo) loader_segment(L, c, ws[i] + (size_t) (lo - mb[i]) * rb, (unsigned) (hi - lo) * rb, lane);
            }
        }
#ifdef MI355Q_STAMPS
        if (g_plan_stamps && lane == 0 && s < g_plan_stamp_stages) {
            g_plan_stamps[(((size_t) s * c.grid + blockIdx.x) * 2) * 8 + 5] = __builtin_amdgcn_s_memrealtime();   // the loader has requested this stage's last byte
            unsigned long long * pp = g_plan_stamps + (size_t) g_plan_stamp_stages * c.grid * 16 + ((size_t) s * c.grid + blockIdx.x) * 8;
            pp[5] = ((unsigned long long) L.tail << 32) | (L.pos >> 10); pp[6] = L.t_blocked; pp[7] = L.t_tail;      // (cumulative since the launch began)
        }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    loader_publish(L, c, (L.pos + 1023u) >> 10, lane);           // (the stream's last page may be partly filled: it is complete as far as there is data)
}

// ---- the consumers' view of a resident row -------------------------------------------------------------------------------------------------
// `row` = LDS address of the row's first byte; WRAP: the row runs over the ring's end (rare: its bytes behind the end continue at the ring's start;
// ring size and wrap point are multiples of 1024 and every field offset a multiple of its size, so no single read straddles)
struct RowView { const uint8_t * ring; unsigned off, ring_bytes; };
template <typename V, bool WRAP> __device__ __forceinline__ V ring_ld(const RowView & r, unsigned x) {
    unsigned a = r.off + x;
    if constexpr (WRAP) { if (a >= r.ring_bytes) a -= r.ring_bytes; }
    return *(const V *) (r.ring + a);
}
// chunk_load<T> of gemv_stream.cuh with the planar row in LDS: same fields, same lanes
template <int T, bool W> __device__ __forceinline__ void chunk_lds(Chunk & ch, const RowView & r, int nb, int s, int lane) {
    ch.q = ring_ld<uint4, W>(r, 1024 * s + 16 * lane);
    if constexpr (T == MI355Q_TYPE_Q4_K) ch.a = ring_ld<uint4, W>(r, 128 * nb + 128 * s + 16 * (lane >> 3));
    if constexpr (T == MI355Q_TYPE_Q5_K) { ch.b = ring_ld<uint4, W>(r, 128 * nb + 256 * s + (32 * (lane >> 3) + 16 * (lane & 1))); ch.a = ring_ld<uint4, W>(r, 160 * nb + 128 * s + 16 * (lane >> 3)); }
    if constexpr (T == MI355Q_TYPE_Q6_K) {
        const int j = lane & 7;
        ch.a  = ring_ld<uint4, W>(r, 128 * nb + 512 * s + (64 * (lane >> 3) + 32 * (j >> 2) + 16 * (j & 1)));
        ch.sc = ring_ld<uint16_t, W>(r, 192 * nb + 128 * s + 2 * lane);
        ch.dh = ring_ld<uint16_t, W>(r, 208 * nb + 16 * s + 2 * (lane >> 3));
    }
    if constexpr (T == MI355Q_TYPE_Q8_0) ch.dh = ring_ld<uint16_t, W>(r, 32 * nb + 64 * s + 2 * (lane >> 1));
    if constexpr (T == MI355Q_TYPE_Q4_0 || T == MI355Q_TYPE_IQ4_NL) ch.dh = ring_ld<uint16_t, W>(r, 16 * nb + 128 * s + 2 * lane);
    if constexpr (T == MI355Q_TYPE_IQ4_XS) { const uint2 h = ring_ld<uint2, W>(r, 128 * nb + 64 * s + 8 * (lane >> 3)); ch.a.x = h.x; ch.a.y = h.y; }
}

struct RowGeom { int nb, nchunks, steps; };
template <int T> __device__ __forceinline__ RowGeom row_geom(int k) {
    RowGeom g;
    g.nb = (T == MI355Q_TYPE_Q8_0 || T == MI355Q_TYPE_Q4_0 || T == MI355Q_TYPE_IQ4_NL) ? k >> 5 : k >> 8;
    g.nchunks = row_chunks(T, k); g.steps = (g.nchunks + 63) >> 6;
    return g;
}

// ---- operand gather ------------------------------------------------------------------------------------------------
// A SPAN is 256 consecutive elements of a vector; lane l owns elements 256 sp + 4l .. 4l+3.  Granules: two 16-byte agent-scope (sc1) buffer
// loads per lane = {v0, tag0, v1, tag1}, {v2, tag2, v3, tag3}; plain vectors: one 16-byte load.  Lanes beyond the vector read zeros (the buffer
// resource bounds the access) and are excluded from the tag check.
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
// one attempt at two elements (attention operands): returns the values and whether they are valid (tags match)
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
// one attempt at a lane's four elements of span sp
__device__ __forceinline__ bool span_try(const SrcView & s, int sp, int lane, float4 & v) {
    if (s.tagged) {
        const plan_u4 a = __builtin_amdgcn_raw_buffer_load_b128(s.rs, sp * 2048 + 32 * lane, 0, 16 /* sc1 */);
        const plan_u4 b = __builtin_amdgcn_raw_buffer_load_b128(s.rs, sp * 2048 + 32 * lane + 16, 0, 16);
        v = make_float4(__uint_as_float(a.x), __uint_as_float(a.z), __uint_as_float(b.x), __uint_as_float(b.z));
        return a.y == s.expect && a.w == s.expect && b.y == s.expect && b.w == s.expect;
    }
    const plan_u4 a = __builtin_amdgcn_raw_buffer_load_b128(s.rs, sp * 1024 + 16 * lane, 0, 0);
    v = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
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

template <int FAM>
__device__ __forceinline__ void plan_quantize_span(const float4 v, int span, uint8_t * lds, int k, bool even, int lane) {
    if constexpr (FAM == FAM_Q8K) quantize_span_to_lds<FAM_Q8K, false>(v, span, lds, k, lane);
    else if (even)                quantize_span_to_lds<FAM_Q80, true>(v, span, lds, k, lane);
    else                          quantize_span_to_lds<FAM_Q80, false>(v, span, lds, k, lane);
}

// Gather the stage's activation vector t[k] into REGISTERS: span wave + 15 i -> t[i].  PLAIN t = x0, NORM t = x0 (+ x1), UNARY_MUL t = unary(x0) * x1.
// Four spans (up to sixteen 16-byte loads per lane) are in flight per poll.  NORM: returns this wave's partial sum of squares in `ssq`; the
// workgroup stage % grid also publishes t (the next residual's operand).  false = the poll gave up.
__device__ __forceinline__ bool plan_gather(StageC st, const Ctx & c, int k, int wave, int lane, float4 (&t)[PLAN_MAXS], double & ssq) {
    const int x_kind = st->x_kind, uop = st->x_unary & 0xFF;
    const bool two = st->x1.plain != nullptr || st->x1.gran != nullptr;
    VecSrc v0, v1;
    v0.plain = st->x0.plain; v0.gran = st->x0.gran; v0.tag_off = st->x0.tag_off; v0.pad = 0;
    v1.plain = st->x1.plain; v1.gran = st->x1.gran; v1.tag_off = st->x1.tag_off; v1.pad = 0;
    const SrcView s0 = src_view(v0, 0, k, c.epoch);
    const SrcView s1 = src_view(two ? v1 : v0, 0, k, c.epoch);
    const bool pub = (st->flags & PLAN_F_SUM) && blockIdx.x == (unsigned) c.stage % c.grid;
    const unsigned tag = c.epoch + st->tag_off;
    const int spans = (k + 255) >> 8;
    PollCtx pc; pc.sync = c.sync; pc.timeout = c.timeout; pc.t0 = __builtin_amdgcn_s_memrealtime(); pc.stage = c.stage;
    unsigned spins = 0;
    ssq = 0.0;
#pragma unroll
    for (int r = 0; r < PLAN_MAXS; r += 4) {
        if (wave + r * PLAN_NC >= spans) break;                 // (uniform)
        float4 a[4], b[4];
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sp = wave + (r + i) * PLAN_NC;
                a[i] = make_float4(0.f, 0.f, 0.f, 0.f); b[i] = a[i];
                if (sp < spans) {                               // (uniform)
                    const bool inside = 256 * sp + 4 * lane < k;
                    ok = (span_try(s0, sp, lane, a[i]) || !inside) && ok;
                    if (two) ok = (span_try(s1, sp, lane, b[i]) || !inside) && ok;
                }
            }
            if (__ballot(!ok) == 0ull) break;
            if (!poll_backoff(pc, spins, lane)) return false;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sp = wave + (r + i) * PLAN_NC;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int e = 256 * sp + 4 * lane;
            if (sp < spans && e < k) {
                if (x_kind == MI355Q_X_UNARY_MUL) { v.x = __fmul_rn(unary_f(uop, a[i].x), b[i].x); v.y = __fmul_rn(unary_f(uop, a[i].y), b[i].y); v.z = __fmul_rn(unary_f(uop, a[i].z), b[i].z); v.w = __fmul_rn(unary_f(uop, a[i].w), b[i].w); }
                else if (two)                     { v.x = __fadd_rn(a[i].x, b[i].x); v.y = __fadd_rn(a[i].y, b[i].y); v.z = __fadd_rn(a[i].z, b[i].z); v.w = __fadd_rn(a[i].w, b[i].w); }
                else                              v = a[i];
                if (x_kind == MI355Q_X_NORM) {
                    // (ggml_float)(x*x): the square is rounded to f32 first; element order inside a lane as the CPU's loop
                    ssq += (double) __fmul_rn(v.x, v.x); ssq += (double) __fmul_rn(v.y, v.y); ssq += (double) __fmul_rn(v.z, v.z); ssq += (double) __fmul_rn(v.w, v.w);
                    if (pub) {
                        publish(st->sum_gran + e, v.x, tag); publish(st->sum_gran + e + 1, v.y, tag); publish(st->sum_gran + e + 2, v.z, tag); publish(st->sum_gran + e + 3, v.w, tag);
                        if (st->flags & PLAN_F_SUM_PLAIN) *(float4 *) (st->sum_plain + e) = v;
                    }
                }
            }
            t[r + i] = v;
        }
    }
    return true;
}

// One GEMV stage, start to finish, for weight type T (consumer waves):
//   [gather x into registers (polls its producers) -> glue -> quantize x -> LDS image] -> the rows of this wave, as the loader lands them
template <int T>
static __device__ __forceinline__ bool plan_stage(StageC st, const Ctx & c, Prog & pr) {
    // an opaque copy of the lane id per stage: everything derived from it is recomputed here (a few VALU ops) instead
    // of being hoisted out of the stage loop for every type's loaders, quantizers and pollers and kept live (and spilled)
    int lane = lane_id(); asm volatile("" : "+v"(lane));
    const int wave = ufl((int) (threadIdx.x >> 6));
    PLAN_STAMP(0);
    const int k = st->k, flags = st->flags;
    const unsigned rb = (unsigned) st->row_bytes;
    int r_lo, r_hi; stage_rows(st, r_lo, r_hi);
    const bool paired = (flags & PLAN_F_PAIRED) != 0;
    const int n_items = r_hi - r_lo;                            // rows (PAIRED: pairs) of this workgroup
    const RowGeom g = row_geom<T>(k);
    // The consumers take the stage's rows in STREAM ORDER in UNITS of up to four consecutive 1-KiB steps of one row (a whole row when it has <= 4
    // steps), unit i to wave i % 15: stream row q = i / units_per_row at byte q * rb.
    const int n_srows = paired ? 2 * n_items : n_items;
    const int NS = g.steps;
    const int U = NS > 4 ? 4 : NS, UPR = (NS + U - 1) / U;      // steps per unit, units per row
    const unsigned base = pr.base_page;                         // (bytes)
    const unsigned stage_bytes = (unsigned) n_srows * rb;
    const int dq = PLAN_NC / UPR, du = PLAN_NC % UPR;           // a wave's next unit is 15 units on: dq rows and du units further
    int q0 = wave / UPR, s0 = (wave % UPR) * U;                 // this wave's first unit: row q0, first step s0
    if (NS <= 4 && paired) { const int G0 = 1 << st->glog, g0 = (wave >> st->glog) << st->glog; q0 = 2 * g0 + (wave - g0); if (wave >= n_items) q0 = n_srows; (void) G0; }   // (short rows: a wave's first ITEM is pair `wave`, whose first row is stream row 2 g0 + i)
    // this wave needs nothing of the stream before its first step of this stage: tell the loader at once (it fills the ring while we poll)
    if (lane == 0) cb_st(c.cb + CB_HEAD + 4u * (unsigned) wave, q0 < n_srows ? (base + (unsigned) q0 * rb + 1024u * (unsigned) s0) >> 10 : (base + stage_bytes) >> 10);
    // the row-slot generations and pair flags of THIS stage were zeroed during the previous GEMV stage (or at launch); zero the other set for the next one
    const unsigned par = (unsigned) (pr.gemv_idx & 1), gen_cb = c.cb + CB_GEN + 128u * par;

    if (flags & PLAN_F_NEW_X) {
        constexpr int FAM = (T == MI355Q_TYPE_Q8_0 || T == MI355Q_TYPE_Q4_0 || T == MI355Q_TYPE_IQ4_NL) ? FAM_Q80 : FAM_Q8K;
        const int x_kind = st->x_kind;
        const float * nw = x_kind == MI355Q_X_NORM ? st->norm_w : nullptr;
        float4 w_first = make_float4(1.f, 1.f, 1.f, 1.f);      // the norm weights of this wave's first span: fetched before the producers are polled
        if (nw && wave * 256 + 4 * lane < k) w_first = *(const float4 *) (nw + wave * 256 + 4 * lane);
        float4 t[PLAN_MAXS];
        double ssq;
        PLAN_STATE(1);
        const bool ok = plan_gather(st, c, k, wave, lane, t, ssq);
        PLAN_STATE(2);
        if (!ok && lane == 0) cb_st(c.cb + CB_OK, 0u);
        PLAN_STAMP(2);
        if (x_kind == MI355Q_X_NORM) {
            ssq = wave_sum_f64(ssq);
            if (lane == 0) *(double *) (c.lds + c.ctl_off + CB_PART + 8 * wave) = ssq;
        }
        // every consumer wave is done with the previous image (its rows of the last stage, or the attention scratch) and has gathered
        if (!cbar(c, pr, lane)) return false;
        PLAN_STAMP(6);
        float scale = 1.0f;
        if (x_kind == MI355Q_X_NORM) {
            const double * part = (const double *) (c.lds + c.ctl_off + CB_PART);
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < PLAN_NC; ++i) s += part[i];
            const float mean = (float) (s / (double) k);
            const float root = (float) sqrt((double) __fadd_rn(mean, st->eps));      // both roundings of the CPU (ops_glue.hip k_add_rms_norm_mul)
            scale = (float) (1.0 / (double) root);
        }
        const int spans = (k + 255) >> 8;
        float * x_out = st->x_out;
        const bool pub_x = x_out != nullptr && blockIdx.x == (unsigned) c.stage % c.grid;
#pragma unroll
        for (int i = 0; i < PLAN_MAXS; ++i) {
            const int span = wave + i * PLAN_NC;
            if (span >= spans) break;                           // (uniform)
            const int e = span * 256 + 4 * lane;
            float4 v = t[i];
            if (x_kind == MI355Q_X_NORM && e < k) {
                v.x = __fmul_rn(v.x, scale); v.y = __fmul_rn(v.y, scale); v.z = __fmul_rn(v.z, scale); v.w = __fmul_rn(v.w, scale);
                if (nw) { const float4 ww = i == 0 ? w_first : *(const float4 *) (nw + e); v.x = __fmul_rn(v.x, ww.x); v.y = __fmul_rn(v.y, ww.y); v.z = __fmul_rn(v.z, ww.z); v.w = __fmul_rn(v.w, ww.w); }
            }
            if (pub_x && e < k) *(float4 *) (x_out + e) = v;     // the formed vector itself is a graph value somebody else reads (result_norm / embeddings, a LoRA branch)
            plan_quantize_span<FAM>(v, span, c.lds, k, c.even != 0, lane);
        }
        PLAN_STAMP(7);
        PLAN_STATE(4);
        if (!cbar(c, pr, lane)) return false;
    }
    else if (!cbar(c, pr, lane)) return false;                 // (a stage that continues on the previous image: the barrier orders the slot reset below)
    PLAN_STAMP(3);
    if (wave == 0 && lane < 32) cb_st(c.cb + CB_GEN + 128u * (par ^ 1u) + 4u * (unsigned) lane, 0u);
    ActView av[1];
    av[0].base = c.lds; av[0].k = k;
    // ROW SLOTS.  A row's steps are computed by different waves; each leaves its lanes' terms (the value gemv_fast.hip adds to the lane's accumulator at
    // that step) in terms[slot][step][lane], and the wave that took the row's LAST step adds them up in step order -- the very additions of the one-wave
    // loop, so the sum is bit-identical -- reduces over the lanes and publishes.  Slot = q % R; a slot is reused when its generation says the row R
    // before has been summed.  Stream order keeps the ring a true FIFO: the pages in use are the ~30 the waves work on, whatever the row length.
    const int R = min(32, max(2, PLAN_TERM_STEPS / NS));
    float * terms = (float *) (c.lds + c.ctl_off + CB_BYTES);
    RowView rv; rv.ring = c.lds + c.ring_off; rv.ring_bytes = c.ring_bytes; rv.off = 0;
    unsigned landed_seen = 0, spins = 0;
    PLAN_PROF_DECL;
    // this wave's current unit (all wave-uniform: the readfirstlanes in unit() say so and keep the state in SGPRs)
    int cq = q0, cs = s0, cslot = q0 % R, clap = q0 / R;
    unsigned coff = (unsigned) q0 * rb, croff = 0, nroff = 0;
    bool more = q0 < n_srows;
    // request the FIRST chunk of the unit that starts at step s of row q once the row has landed (the small fields of its blocks are the row's last
    // bytes) and its slot is free.  wait = false: one look at each condition, no waiting (0 = not yet); returns -1 when a wait gave up
    auto request = [&](int q, int s, int slot, int lap, unsigned off, Chunk & ch, bool wait) -> int {
        const unsigned need = (base + off + rb + 1023u) >> 10;
        while ((int) (landed_seen - need) < 0) {
            landed_seen = cb_ld(c.cb + CB_LANDED);
            if ((int) (landed_seen - need) >= 0) break;
            if (!wait) return 0;
            if (!lds_spin(c, spins, lane, WHY_LANDED, need, (unsigned) q * 64u + (unsigned) s)) return -1;
        }
        if (wait) PLAN_PROF(3);
        if (lap > 0) while ((int) (cb_ld(gen_cb + 4u * (unsigned) slot) - (unsigned) lap) < 0) {
            if (!wait) return 0;
            if (!lds_spin(c, spins, lane, WHY_SLOT, (unsigned) q * 64u + (unsigned) s, (unsigned) slot * 65536u + (unsigned) lap)) return -1;
        }
        spins = 0;
        nroff = uflu(ring_pos(c, base + off));
        rv.off = nroff;
        if (64 * s + lane < g.nchunks) {
            if (nroff + rb <= c.ring_bytes) chunk_lds<T, false>(ch, rv, g.nb, s, lane);      // (uniform branch)
            else                           chunk_lds<T, true>(ch, rv, g.nb, s, lane);
        }
        return 1;
    };
    // one unit: request the first chunk of this wave's NEXT unit into `other` if that needs no waiting, then the steps of the current one (its first
    // chunk is in `mine`; the following ones are read while the previous step is computed) and their terms; the wave whose arrival completes a row
    // closes it (nobody waits for anybody)
    auto unit = [&](Chunk & mine, Chunk & other) -> bool {
        cq = ufl(cq); cs = ufl(cs); cslot = ufl(cslot); clap = ufl(clap); coff = uflu(coff); croff = uflu(croff); landed_seen = uflu(landed_seen);
        int nq = cq + dq, ns = cs + du * U;
        if (ns >= UPR * U) { ns -= UPR * U; ++nq; }
        const int drow = nq - cq;
        const unsigned noff = coff + (unsigned) drow * rb;
        int nslot = cslot + drow, nlap = clap;
        while (nslot >= R) { nslot -= R; ++nlap; }
        const bool nmore = nq < n_srows;
        const int s_end = min(NS, cs + U);
        // (the LDS unit executes a wave's instructions in order: this word is written after the reads of every earlier unit have been performed)
        if (lane == 0) cb_st(c.cb + CB_HEAD + 4u * (unsigned) wave, (base + coff + 1024u * (unsigned) cs) >> 10);
        PLAN_PROF(0);
        PLAN_STATE(0x50000 | (nq & 0xFFFF));
        int have = 0;
        if (nmore) { have = ufl(request(nq, ns, nslot, nlap, noff, other, false)); if (have < 0) return false; }
        PLAN_STATE(0x60000 | (cq & 0xFFFF));
        PLAN_PROF(1);
        rv.off = croff;
        const bool wraps = croff + rb > c.ring_bytes;
        Chunk cur = mine, nxt;
#pragma unroll 1
        for (int s = cs; s < s_end; ++s) {
            if (s + 1 < s_end && 64 * (s + 1) + lane < g.nchunks) {
                if (!wraps) chunk_lds<T, false>(nxt, rv, g.nb, s + 1, lane);
                else        chunk_lds<T, true>(nxt, rv, g.nb, s + 1, lane);
            }
            float acc[1] = { 0.0f };
            if (64 * s + lane < g.nchunks) Consume<T, 1>::run(cur, s, lane, av, acc);
            terms[(cslot * NS + s) * 64 + lane] = acc[0];
            cur = nxt;
        }
        asm volatile("" ::: "memory");                          // (program order: the term stores are issued before the arrival; the LDS unit keeps that order)
        const unsigned n_mine = (unsigned) (s_end - cs);
        const unsigned arrived = cb_arrive_n(c.cb + CB_CNT + 4u * (unsigned) cslot, n_mine, lane) + n_mine;
        PLAN_PROF(2);
        if (arrived == (unsigned) NS) {                          // the row is complete: its terms, added in step order
            PLAN_STATE(0x70000 | (cq & 0xFFFF));
            float a = 0.0f;
#pragma unroll 1
            for (int ss = 0; ss < NS; ++ss) a += terms[(cslot * NS + ss) * 64 + lane];
            const float d = wave_sum(a);
            asm volatile("" ::: "memory");
            if (lane == 0) { cb_st(c.cb + CB_CNT + 4u * (unsigned) cslot, 0u); cb_st(gen_cb + 4u * (unsigned) cslot, (unsigned) clap + 1u); }      // (after the reads above: the slot may take its next row)
            const unsigned tag = c.epoch + st->tag_off;
            const bool plain = (st->flags & PLAN_F_PLAIN_Y) != 0;
            if (!paired) {
                if (lane == 0) {
                    const int gr = r_lo + cq;
                    publish(st->yg + gr, d, tag);
                    if (plain) {
                        const int b1 = st->row_begin[1], b2 = st->row_begin[2], b3 = st->row_begin[3];
                        const int mi = (gr >= b1) + (gr >= b2) + (gr >= b3);
                        float * y = mi == 0 ? st->y[0] : mi == 1 ? st->y[1] : mi == 2 ? st->y[2] : st->y[3];
                        y[gr - (mi == 0 ? 0 : mi == 1 ? b1 : mi == 2 ? b2 : b3)] = d;
                    }
                }
            } else {
                const int glog = st->glog, G = 1 << glog;
                const int g0 = (cq >> (glog + 1)) << glog, w = cq & (2 * G - 1), n_g = min(G, n_items - g0);
                const int which = w >= n_g ? 1 : 0, p = g0 + w - which * n_g;
                // the pair's two dot products meet in LDS: each closer leaves its value, the second to arrive forms the product
                if (lane == 0) cb_st(c.cb + CB_PDV + 256u * (unsigned) which + 4u * (unsigned) (p & 63), __float_as_uint(d));
                if (cb_arrive(c.cb + CB_PCNT + 4u * (unsigned) (p & 63), 1u, lane) == 1u) {
                    const float dg = which ? __uint_as_float(cb_ld(c.cb + CB_PDV + 4u * (unsigned) (p & 63))) : d;
                    const float du_ = which ? d : __uint_as_float(cb_ld(c.cb + CB_PDV + 256u + 4u * (unsigned) (p & 63)));
                    if (lane == 0) {
                        const float r = __fmul_rn(unary_f(st->x_unary >> 8, dg), du_);      // (a PAIRED stage keeps its output unary in the high byte of x_unary)
                        publish(st->yg + r_lo + p, r, tag);
                        if (plain) st->y[0][r_lo + p] = r;
                    }
                }
            }
            PLAN_PROF(2);
        }
        if (nmore && !have) { PLAN_PROF(2); if (ufl(request(nq, ns, nslot, nlap, noff, other, true)) < 0) return false; PLAN_PROF(4); }
        cq = nq; cs = ns; cslot = nslot; clap = nlap; coff = noff; croff = nroff; more = nmore;
        return true;
    };
    Chunk chA, chB;
    bool ok_rows = true;
    if (NS <= 4) {
        // SHORT ROWS (k <= 8192 for the 4-bit types): a wave takes whole rows -- PAIRED: whole pairs, both rows -- item i to wave i % 15, and sums a
        // row's steps in its own registers exactly as gemv_fast.hip does: no term buffer, no arrival counters, no slots.  The rows the 15 waves
        // work on span 15 (30) short rows of the stream, well inside the ring.
        const int glog = st->glog, G = 1 << glog;
        const unsigned tag = c.epoch + st->tag_off;
        const bool plain = (st->flags & PLAN_F_PLAIN_Y) != 0;
        auto row_dot = [&](unsigned off) -> float {             // the dot product of the resident row at stream byte `off` of this stage
            rv.off = uflu(ring_pos(c, base + off));
            const bool wraps = rv.off + rb > c.ring_bytes;
            float acc[1] = { 0.0f };
            Chunk cur, nxt;
            if (lane < g.nchunks) { if (!wraps) chunk_lds<T, false>(cur, rv, g.nb, 0, lane); else chunk_lds<T, true>(cur, rv, g.nb, 0, lane); }
#pragma unroll 1
            for (int s = 0; s < NS; ++s) {
                if (s + 1 < NS && 64 * (s + 1) + lane < g.nchunks) { if (!wraps) chunk_lds<T, false>(nxt, rv, g.nb, s + 1, lane); else chunk_lds<T, true>(nxt, rv, g.nb, s + 1, lane); }
                if (64 * s + lane < g.nchunks) Consume<T, 1>::run(cur, s, lane, av, acc);
                cur = nxt;
            }
            return wave_sum(acc[0]);
        };
        auto item_offs = [&](int it, unsigned & o0, unsigned & o1) {   // stream byte offsets of item `it`: its row, or (PAIRED) its two rows
            if (paired) { const int g0 = (it >> glog) << glog, n_g = min(G, n_items - g0); o0 = (unsigned) (2 * g0 + (it - g0)) * rb; o1 = o0 + (unsigned) n_g * rb; }
            else        { o0 = o1 = (unsigned) it * rb; }
        };
#pragma unroll 1
        for (int it = wave; it < n_items && ok_rows; it += PLAN_NC) {
            unsigned o0, o1; item_offs(ufl(it), o0, o1);
            o0 = uflu(o0); o1 = uflu(o1);
            if (lane == 0) cb_st(c.cb + CB_HEAD + 4u * (unsigned) wave, (base + o0) >> 10);
            const unsigned need = (base + o1 + rb + 1023u) >> 10;
            while ((int) (landed_seen - need) < 0) { landed_seen = cb_ld(c.cb + CB_LANDED); if ((int) (landed_seen - need) >= 0) break; if (!lds_spin(c, spins, lane, WHY_LANDED, need, (unsigned) it)) { ok_rows = false; break; } }
            if (!ok_rows) break;
            spins = 0;
            if (it == wave) PLAN_STAMP(1);
            PLAN_PROF(3);
            const float d0 = row_dot(o0);
            if (paired) {
                const float d1 = row_dot(o1);
                if (lane == 0) {
                    const float r = __fmul_rn(unary_f(st->x_unary >> 8, d0), d1);      // (a PAIRED stage keeps its output unary in the high byte of x_unary)
                    publish(st->yg + r_lo + it, r, tag);
                    if (plain) st->y[0][r_lo + it] = r;
                }
            } else if (lane == 0) {
                const int gr = r_lo + it;
                publish(st->yg + gr, d0, tag);
                if (plain) {
                    const int b1 = st->row_begin[1], b2 = st->row_begin[2], b3 = st->row_begin[3];
                    const int mi = (gr >= b1) + (gr >= b2) + (gr >= b3);
                    float * y = mi == 0 ? st->y[0] : mi == 1 ? st->y[1] : mi == 2 ? st->y[2] : st->y[3];
                    y[gr - (mi == 0 ? 0 : mi == 1 ? b1 : mi == 2 ? b2 : b3)] = d0;
                }
            }
            PLAN_PROF(2);
        }
    } else {
        if (more) { ok_rows = ufl(request(cq, cs, cslot, clap, coff, chA, true)) > 0; croff = nroff; PLAN_STAMP(1); }
#pragma unroll 1
        while (more && ok_rows) {
            ok_rows = unit(chA, chB);
            if (!more || !ok_rows) break;
            ok_rows = unit(chB, chA);
        }
    }
    if (!ok_rows) { if (lane == 0) cb_st(c.cb + CB_OK, 0u); return false; }
    if (lane == 0) cb_st(c.cb + CB_HEAD + 4u * (unsigned) wave, (base + stage_bytes) >> 10);
    PLAN_PROF_FLUSH;
    PLAN_STATE(9);
    ++pr.gemv_idx;
    pr.base_page = base + stage_bytes;
    PLAN_STAMP(4);
    return true;
}

// ---- attention of one token ------------------------------------------------------------------------------------------
// Workgroup b = (head h, KV split sp), consumer waves only.  LDS (the image area: no image is live during an ATTN stage): sq / sk / sv f32 [hd],
// kh / vh f16 [hd] (this token's cache row, rounded as stored), sc f32 [per] (scores, then probabilities), red f32 [15][hd] (P V partials), maxs / sums.
typedef const __attribute__((address_space(4))) AttnStage * AttnC;          // the descriptor is read with scalar loads

static __device__ __noinline__ bool plan_attn(AttnC a_in, const Ctx & c_in, unsigned tag_in, unsigned * bar_io) {
    const AttnC a = (AttnC) (uintptr_t) uniform_ptr((const void *) (uintptr_t) a_in);      // (see uniform_ptr)
    // LDS pointers are re-derived from the kernel's LDS symbol: taken from the caller's struct they would be generic pointers (flat loads)
    extern __shared__ __attribute__((aligned(16))) uint8_t plan_lds_attn[];
    Ctx c = c_in;
    c.lds = plan_lds_attn; c.cb = uflu(c_in.cb); c.sync = uniform_ptr(c_in.sync);
    c.grid = uflu(c_in.grid); c.epoch = uflu(c_in.epoch); c.stage = ufl(c_in.stage);
    Prog pr; pr.gemv_idx = 0; pr.base_page = 0; pr.bar_target = uflu(*bar_io);
    const unsigned tag = uflu(tag_in);
    const int lane = lane_id();
    const int wave = ufl((int) (threadIdx.x >> 6));
    const int tid = (int) threadIdx.x;
    const int hd = a->hd, n_split = a->n_split;
    if ((int) blockIdx.x >= a->n_head * n_split) return true;                    // (uniform: the whole workgroup has nothing to do)
    const int h = (int) blockIdx.x / n_split, sp = (int) blockIdx.x % n_split, gq = a->n_head / a->n_head_kv, g = h / gq;
    float * sq = (float *) c.lds, * sk = sq + hd, * sv = sk + hd;
    __half * kh = (__half *) (sv + hd), * vh = kh + hd;
    float * maxs = (float *) (vh + hd), * sums = maxs + GEMV_WAVES;
    float * red = sums + GEMV_WAVES;                                          // [15][hd]
    float * sc = red + PLAN_NC * hd;                                           // [per]
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
    bool ok_bar = cbar(c, pr, lane);                                               // the image area is free (the previous stage's rows are done)
    PLAN_STAMP(0);
    if (ok_bar) for (int jj = tid; jj < cnt; jj += PLAN_CT)                    // the additive mask of this split's positions (0 without a mask)
        sc[jj] = maskp ? (mask_f16 ? __half2float(((const __half *) maskp)[j0 + jj]) : ((const float *) maskp)[j0 + jj]) : 0.0f;

    // 1. q head h, k / v head g  (hd <= 256: at most two 128-element chunks each)
    if (ok_bar) {
        PollCtx pc; pc.sync = c.sync; pc.timeout = c.timeout; pc.t0 = __builtin_amdgcn_s_memrealtime(); pc.stage = c.stage;
        unsigned spins = 0;
        const int nch = (hd + 127) >> 7;
        bool ok_all = true;
        for (int item = wave; item < 3 * nch; item += PLAN_NC) {
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
        if (!ok_all && lane == 0) cb_st(c.cb + CB_OK, 0u);
    }
    if (!ok_bar || !cbar(c, pr, lane)) { *bar_io = pr.bar_target; return false; }
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
    if (!cbar(c, pr, lane)) { *bar_io = pr.bar_target; return false; }
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
    for (int jb = wave; jb < cnt; jb += SB * PLAN_NC) {
        __half2 kv[SB], kw[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int j = j0 + jb + u * PLAN_NC;
            kv[u] = __half2(); kw[u] = __half2();
            if (jb + u * PLAN_NC < cnt) {
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
            if (jb + u * PLAN_NC < cnt) {
                const float2 f = __half22float2(kv[u]), f2 = __half22float2(kw[u]);
                float d = q0 * f.x + q1 * f.y + q2 * f2.x + q3 * f2.y;
                d = wave_sum(d);
                // (a fully masked position stays -inf whatever its cache row holds: never-written rows may be anything, 0 * NaN included)
                if (lane == 0) { const float m = sc[jb + u * PLAN_NC]; sc[jb + u * PLAN_NC] = m == -INFINITY ? -INFINITY : __fadd_rn(__fmul_rn(d, scale), m); }
            }
        }
    }
    if (!cbar(c, pr, lane)) { *bar_io = pr.bar_target; return false; }
    PLAN_STAMP(3);
    // 4. local softmax statistics.  A split holds at most a few hundred scores: ONE wave forms maximum, exponentials, sum and (single split) the
    //    probabilities with register reductions, the other 14 wait at one barrier -- the workgroup-wide form cost three barriers for the same numbers.
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
            for (int j = lane; j < cnt; j += 64) sc[j] = __half2float(__float2half_rn(__fmul_rn(sc[j], inv)));      // (a lane re-reads only what it wrote)
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
    if (!cbar(c, pr, lane)) { *bar_io = pr.bar_target; return false; }
    const float mx = maxs[0], l = sums[0];
    PLAN_STAMP(4);
    // 5. o[d] = sum_j p_j v[j][d]   (positions with p == 0 are skipped: masked cache rows may hold anything)
    const char * vbase = a->v_cache + (int64_t) g * a->v_nb_head;
    const int64_t v_nb_pos = a->v_nb_pos, v_nb_dim = a->v_nb_dim;
    if (v_nb_dim == 2) {
        // rows per position (the -fa layout): a wave takes positions jb = wave, wave+15, ...; a lane owns dims (2l, 2l+1) [+128]
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll 1
        for (int jb = wave; jb < cnt; jb += SB * PLAN_NC) {
            __half2 vv[SB], vw[SB]; float p[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int jj = jb + u * PLAN_NC, j = j0 + jj;
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
        if (!cbar(c, pr, lane)) { *bar_io = pr.bar_target; return false; }
        if (tid < hd) {
            float o = 0.0f;
#pragma unroll
            for (int i = 0; i < PLAN_NC; ++i) o += red[i * hd + tid];
            red[tid] = o;                                                      // (row 0 of red now holds o; every thread touches only its column)
        }
    } else {
        // transposed cache (positions contiguous per dim): a wave takes dims d = wave + 15 i; lanes run over the positions; the loads of
        // 8 dims are in flight together (one memory round trip per 64 positions instead of one per dim)
#pragma unroll 1
        for (int d0 = wave; d0 < hd; d0 += SB * PLAN_NC) {
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
                    const int d = d0 + u * PLAN_NC;
                    vv[u] = __half();
                    if (p != 0.0f && d < hd) vv[u] = j == slot ? vh[d] : *(const __half *) (vbase + (int64_t) d * v_nb_dim + (int64_t) j * v_nb_pos);
                }
#pragma unroll
                for (int u = 0; u < SB; ++u) o[u] += p * __half2float(vv[u]);
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int d = d0 + u * PLAN_NC;
                if (d < hd) { const float t = wave_sum(o[u]); if (lane == 0) red[d] = t; }     // (d wave-uniform)
            }
        }
    }
    if (!cbar(c, pr, lane)) { *bar_io = pr.bar_target; return false; }
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
    *bar_io = pr.bar_target;
    return true;
}

// merge the KV splits of a head: out = sum_s e^{m_s - M} o_s / sum_s e^{m_s - M} l_s     (workgroup h * n_split does head h)
static __device__ __noinline__ bool plan_attn_combine(AttnC a_in, const Ctx & c_in, unsigned tag_in, unsigned * bar_io) {
    const AttnC a = (AttnC) (uintptr_t) uniform_ptr((const void *) (uintptr_t) a_in);
    const unsigned tag = uflu(tag_in);
    extern __shared__ __attribute__((aligned(16))) uint8_t plan_lds_comb[];
    Ctx c = c_in;
    c.lds = plan_lds_comb; c.cb = uflu(c_in.cb); c.sync = uniform_ptr(c_in.sync);
    c.grid = uflu(c_in.grid); c.epoch = uflu(c_in.epoch); c.stage = ufl(c_in.stage);
    Prog pr; pr.gemv_idx = 0; pr.base_page = 0; pr.bar_target = uflu(*bar_io);
    const int lane = lane_id();
    const int wave = ufl((int) (threadIdx.x >> 6));
    const int tid = (int) threadIdx.x;
    const int hd = a->hd, n_split = a->n_split;
    if ((int) blockIdx.x >= a->n_head * n_split || (int) blockIdx.x % n_split != 0) return true;
    const int h = (int) blockIdx.x / n_split;
    const int n = n_split * (hd + 2);
    float * buf = (float *) c.lds;
    bool ok_bar = cbar(c, pr, lane);
    PLAN_STAMP(0);
    if (ok_bar) {
        VecSrc vs; vs.plain = nullptr; vs.gran = a->part + (size_t) h * n; vs.tag_off = 0; vs.pad = 0;
        SrcView s = src_view(vs, 0, n, 0); s.expect = tag - 1;                  // the partials carry the ATTN stage's tag (the stage before this one)
        PollCtx pc; pc.sync = c.sync; pc.timeout = c.timeout; pc.t0 = __builtin_amdgcn_s_memrealtime(); pc.stage = c.stage;
        unsigned spins = 0;
        bool ok_all = true;
        for (int ch = wave; ch < ((n + 127) >> 7); ch += PLAN_NC) {
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
        if (!ok_all && lane == 0) cb_st(c.cb + CB_OK, 0u);
    }
    if (!ok_bar || !cbar(c, pr, lane)) { *bar_io = pr.bar_target; return false; }
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
    *bar_io = pr.bar_target;
    return true;
}

template <unsigned SET>
__global__ void __launch_bounds__(GEMV_THREADS)
k_plan(const PlanStage * stages_g, int n_stages, unsigned * sync, int even, unsigned long long timeout_ticks, int ctl_off, int ring_off, int np, unsigned ring_magic, unsigned epoch) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    StageC stages = (StageC) stages_g;
    if (__hip_atomic_load(sync + PLAN_SYNC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;   // sticky: a plan that timed out stays dead

    Ctx c;
    c.lds = lds; c.cb = (unsigned) (size_t) lds + (unsigned) ctl_off;
    c.ring_off = (unsigned) ring_off; c.np = (unsigned) np; c.ring_bytes = (unsigned) np * 1024u; c.ring_magic = ring_magic; c.np_magic = (unsigned) (0x100000000ull / (unsigned long long) np);
    c.sync = sync; c.timeout = timeout_ticks; c.grid = gridDim.x; c.even = even; c.epoch = epoch;
    c.stage = 0; c.ctl_off = (unsigned) ctl_off;
    Prog pr; pr.gemv_idx = 0; pr.base_page = 0; pr.bar_target = 0;
    if (threadIdx.x < CB_DESC / 4) {
        unsigned v = 0;
        if (threadIdx.x == CB_OK / 4) v = 1u;
        if (threadIdx.x >= CB_HEAD / 4 + PLAN_NC && threadIdx.x < CB_HEAD / 4 + 16) v = 0xFFFFFFFFu;          // the words behind the consumers' heads: the loader's minimum ignores them
        ((unsigned *) (lds + ctl_off))[threadIdx.x] = v;
    }
    __syncthreads();                                                          // the ONLY hardware barrier of the launch: the loader never joins another one
    const int wave = ufl((int) (threadIdx.x >> 6));
    c.even = even & 1;
    if (wave == PLAN_NC) { plan_loader(stages_g, n_stages, c); return; }
    if (wave > PLAN_NC) return;
    if (even & 2) {                                                           // dev (MI355Q_PLAN_LOADER_ONLY=1): the loader alone, nothing is computed -- how fast does the weight stream run by itself?
        if ((threadIdx.x & 63) == 0) cb_st(c.cb + CB_HEAD + 4u * (unsigned) wave, 0x7FFFFFFFu);
        return;
    }

#pragma unroll 1
    for (int s = 0; s < n_stages; ++s) {
        StageC st = stages + s;
        c.stage = s;
        bool ok = true;
        const int kind = st->kind;
        if (kind == PLAN_K_GEMV) {
            bool ran = false;
            switch (st->type) {
            case MI355Q_TYPE_Q4_K: if constexpr ((SET & tbit(MI355Q_TYPE_Q4_K)) != 0) { ok = plan_stage<MI355Q_TYPE_Q4_K>(st, c, pr); ran = true; } break;
            case MI355Q_TYPE_Q5_K: if constexpr ((SET & tbit(MI355Q_TYPE_Q5_K)) != 0) { ok = plan_stage<MI355Q_TYPE_Q5_K>(st, c, pr); ran = true; } break;
            case MI355Q_TYPE_Q6_K: if constexpr ((SET & tbit(MI355Q_TYPE_Q6_K)) != 0) { ok = plan_stage<MI355Q_TYPE_Q6_K>(st, c, pr); ran = true; } break;
            case MI355Q_TYPE_Q8_0: if constexpr ((SET & tbit(MI355Q_TYPE_Q8_0)) != 0) { ok = plan_stage<MI355Q_TYPE_Q8_0>(st, c, pr); ran = true; } break;
            case MI355Q_TYPE_Q4_0: if constexpr ((SET & tbit(MI355Q_TYPE_Q4_0)) != 0) { ok = plan_stage<MI355Q_TYPE_Q4_0>(st, c, pr); ran = true; } break;
            case MI355Q_TYPE_IQ4_NL: if constexpr ((SET & tbit(MI355Q_TYPE_IQ4_NL)) != 0) { ok = plan_stage<MI355Q_TYPE_IQ4_NL>(st, c, pr); ran = true; } break;
            case MI355Q_TYPE_IQ4_XS: if constexpr ((SET & tbit(MI355Q_TYPE_IQ4_XS)) != 0) { ok = plan_stage<MI355Q_TYPE_IQ4_XS>(st, c, pr); ran = true; } break;
            default: break;
            }
            if (!ran) { pr.base_page += stage_stream_bytes(st); ++pr.gemv_idx; }      // (a type outside this instantiation's set: never happens for a created plan)
        } else if (kind == PLAN_K_ATTN) {
            { unsigned bar = pr.bar_target; ok = plan_attn((AttnC) st->attn, c, epoch + st->tag_off, &bar); pr.bar_target = bar; }
        } else {
            { unsigned bar = pr.bar_target; ok = plan_attn_combine((AttnC) st->attn, c, epoch + st->tag_off, &bar); pr.bar_target = bar; }
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
    size_t        lds_total = 0;
    int           ctl_off = 0, ring_off = 0, np = 0;
    int64_t       weight_bytes = 0;
    PlanStage *   d_stages = nullptr;
    AttnStage *   d_attn = nullptr;
    unsigned *    d_sync = nullptr;
    Granule *     d_gran = nullptr; size_t gran_count = 0;
    unsigned long long runs = 0, timeout_ticks = 0;
};

} int gemv_fast_family(int type); namespace ring {

static const void * plan_kernel(unsigned set) {
    if ((set & ~SET_K46) == 0)  return (const void *) k_plan<SET_K46>;
    if ((set & ~SET_K456) == 0) return (const void *) k_plan<SET_K456>;
    if ((set & ~SET_80) == 0)   return (const void *) k_plan<SET_80>;
    if ((set & ~SET_ALL) == 0)  return (const void *) k_plan<SET_ALL>;
    if ((set & ~SET_IQ4) == 0)  return (const void *) k_plan<SET_IQ4>;
    return nullptr;                                            // (a mix of the IQ4 types with Q4_K / Q8_0 / Q4_0: no instantiation)
}

#ifdef MI355Q_STAMPS
extern "C" int mi355q_debug_set_ring_stamps(void * dev_buf, int n_stages) {
    unsigned long long * p = (unsigned long long *) dev_buf;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_plan_stamps), &p, sizeof(p)) != hipSuccess) return -4;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_plan_stamp_stages), &n_stages, sizeof(int)) == hipSuccess ? 0 : -4;
}
#endif

} } // namespace mi355q::ring

using namespace mi355q;
using namespace mi355q::ring;

extern "C" {

void mi355q_set_error(const char * msg);          // api.hip

// outputs published so far while the stage list is built: [ptr, ptr + n) f32 <-> granule offset, producing stage
namespace { struct OutRange { const float * p; int64_t n; size_t gran_off; unsigned tag_off; int64_t id; }; }

int mi355q_ring_plan_create(mi355q_plan ** out, const mi355q_stage * stages, int n_stages, int flags) {
    if (!out || !stages || n_stages < 1) { mi355q_set_error("plan_create: null argument / no stages"); return MI355Q_ERR_SHAPE; }
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { mi355q_set_error("plan_create: no device"); return MI355Q_ERR_HIP; }
    const int n_cu = prop.multiProcessorCount;

    std::vector<PlanStage> v;
    std::vector<AttnStage> va;
    std::vector<int> attn_of;                                  // per internal stage: index into va or -1
    std::vector<OutRange> outs;
    size_t gran_count = 0;
    auto new_out = [&](const float * p, int64_t n, unsigned tag_off, int64_t id) {
        OutRange r = { p, n, gran_count, tag_off, id };
        gran_count += (size_t) ((n + 3) & ~(int64_t) 3);       // keep every vector 32-byte aligned (a lane's four granules are two 16-byte loads)
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
    auto overlaps = [](const float * a, int64_t na, const float * b, int64_t nb) { return a && b && a < b + nb && b < a + na; };
    unsigned set = 0; size_t img_max = 0; int64_t bytes = 0; size_t row_max = 0;
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
            PlanStage p = {}; p.kind = PLAN_K_ATTN; p.kf = PLAN_K_ATTN; p.tag_off = (unsigned) v.size() + 1; p.flags = PLAN_F_NEW_X;
            if (A.n_split == 1) {
                A.out_gran = (Granule *) (uintptr_t) (new_out(at->out, (int64_t) at->n_head * hd, p.tag_off, at->out_id) + 1);
                v.push_back(p); attn_of.push_back((int) va.size());
            } else {
                const size_t part_off = gran_count; gran_count += (size_t) at->n_head * A.n_split * (hd + 2); gran_count = (gran_count + 3) & ~(size_t) 3;
                A.part = (Granule *) (uintptr_t) (part_off + 1);
                v.push_back(p); attn_of.push_back((int) va.size());
                PlanStage q = {}; q.kind = PLAN_K_COMBINE; q.kf = PLAN_K_COMBINE; q.tag_off = (unsigned) v.size() + 1; q.flags = PLAN_F_NEW_X;
                A.out_gran = (Granule *) (uintptr_t) (new_out(at->out, (int64_t) at->n_head * hd, q.tag_off, at->out_id) + 1);
                v.push_back(q); attn_of.push_back((int) va.size());
            }
            va.push_back(A);
            const size_t need = (size_t) 4 * (3 * hd + hd /* kh, vh */ + 2 * GEMV_WAVES + PLAN_NC * hd + A.per) + 64;
            const size_t need2 = (size_t) 4 * A.n_split * (hd + 2) + 64;
            if (need > img_max) img_max = need;
            if (need2 > img_max) img_max = need2;
            bytes += (int64_t) 2 * at->n_kv * at->n_head_kv * hd * 2;
            continue;
        }
        if (in.kind != MI355Q_STAGE_GEMV) { mi355q_set_error("plan_create: unknown stage kind"); return MI355Q_ERR_UNSUPPORTED; }
        if (in.n_mats < 1 || in.n_mats > GEMV_MAX_MATS || !in.x || in.k <= 0 || (in.k & 31) || ((uintptr_t) in.x & 15)) { mi355q_set_error("plan_create: bad stage (k % 32, x 16-byte aligned)"); return MI355Q_ERR_SHAPE; }
        if (in.k > (int64_t) PLAN_MAXS * PLAN_NC * 256) { mi355q_set_error("plan_create: k too large for the gather registers"); return MI355Q_ERR_UNSUPPORTED; }
        if (in.x_kind < MI355Q_X_PLAIN || in.x_kind > MI355Q_X_UNARY_MUL) { mi355q_set_error("plan_create: unknown x_kind"); return MI355Q_ERR_UNSUPPORTED; }
        if (in.x_kind == MI355Q_X_UNARY_MUL && (!in.x1 || (in.x_unary != MI355Q_UNARY_SILU && in.x_unary != MI355Q_UNARY_RELU && in.x_unary != MI355Q_UNARY_SIGMOID))) { mi355q_set_error("plan_create: X_UNARY_MUL needs x1 and SILU / RELU / SIGMOID"); return MI355Q_ERR_UNSUPPORTED; }
        if (in.x_kind == MI355Q_X_PLAIN && in.x1) { mi355q_set_error("plan_create: X_PLAIN takes one operand"); return MI355Q_ERR_SHAPE; }
        if (in.x_kind == MI355Q_X_NORM && (((uintptr_t) in.norm_w & 15) || (in.k & 3))) { mi355q_set_error("plan_create: norm weights must be 16-byte aligned"); return MI355Q_ERR_ALIGN; }
        if (in.x1 && ((uintptr_t) in.x1 & 15)) { mi355q_set_error("plan_create: x1 must be 16-byte aligned"); return MI355Q_ERR_ALIGN; }
        if ((in.sum_out && ((uintptr_t) in.sum_out & 15)) || (in.x_out && ((uintptr_t) in.x_out & 15))) { mi355q_set_error("plan_create: sum_out / x_out must be 16-byte aligned"); return MI355Q_ERR_ALIGN; }
        // a plain store of this stage must not land on a plain operand the same stage still reads in other workgroups (ggml-alloc makes ADD in place)
        if (in.sum_out && !(in.flags & MI355Q_STAGE_NO_PLAIN)) {
            VecSrc t0, t1; const bool r0 = resolve(in.x, in.k, in.x_id, t0), r1 = resolve(in.x1, in.k, in.x1_id, t1);
            if ((r0 && t0.plain && overlaps(in.sum_out, in.k, in.x, in.k)) || (r1 && t1.plain && overlaps(in.sum_out, in.k, in.x1, in.k))) { mi355q_set_error("plan_create: sum_out overlaps a plain operand of the same stage"); return MI355Q_ERR_SHAPE; }
        }
        if (in.x_out) {
            VecSrc t0, t1; const bool r0 = resolve(in.x, in.k, in.x_id, t0), r1 = resolve(in.x1, in.k, in.x1_id, t1);
            if ((r0 && t0.plain && overlaps(in.x_out, in.k, in.x, in.k)) || (r1 && t1.plain && overlaps(in.x_out, in.k, in.x1, in.k))) { mi355q_set_error("plan_create: x_out overlaps a plain operand of the same stage"); return MI355Q_ERR_SHAPE; }
        }
        const bool paired = in.y_kind == MI355Q_Y_UNARY_MUL;
        if (paired && (in.n_mats != 2 || in.mats[0].type != in.mats[1].type || in.mats[0].m != in.mats[1].m ||
                       (in.y_unary != MI355Q_UNARY_SILU && in.y_unary != MI355Q_UNARY_RELU && in.y_unary != MI355Q_UNARY_SIGMOID))) { mi355q_set_error("plan_create: Y_UNARY_MUL needs two matrices of one type and size and SILU / RELU / SIGMOID"); return MI355Q_ERR_UNSUPPORTED; }
        if (in.y_kind != MI355Q_Y_ROWS && !paired) { mi355q_set_error("plan_create: unknown y_kind"); return MI355Q_ERR_UNSUPPORTED; }
        VecSrc x0, x1;
        if (!resolve(in.x, in.k, in.x_id, x0) || !resolve(in.x1, in.k, in.x1_id, x1)) { mi355q_set_error("plan_create: an activation operand straddles an earlier output"); return MI355Q_ERR_SHAPE; }
        if ((x0.gran && (((uintptr_t) x0.gran - 1) & 1)) || (x1.gran && (((uintptr_t) x1.gran - 1) & 1))) { mi355q_set_error("plan_create: an operand starts at an odd element of an earlier output"); return MI355Q_ERR_ALIGN; }
        // the matrices of a stage share ONE quantized image of the activations: their types must pair with the same activation format
        for (int i = 1; i < in.n_mats; ++i)
            if (gemv_fast_family(in.mats[i].type) != gemv_fast_family(in.mats[0].type)) { mi355q_set_error("plan_create: the matrices of a stage must share the activation format (Q8_K or Q8_0 family)"); return MI355Q_ERR_UNSUPPORTED; }
        bool done[GEMV_MAX_MATS] = { false, false, false, false };
        bool first = true;
        for (int i = 0; i < in.n_mats; ++i) {
            if (done[i]) continue;
            const int type = in.mats[i].type;
            const int fam = gemv_fast_family(type);
            if (fam < 0 || !(tbit(type) & SET_ANY) || !mi355q_weights_are_planar(type, in.k)) { mi355q_set_error("plan_create: weight type / k has no planar streaming kernel in the plan"); return MI355Q_ERR_UNSUPPORTED; }
            const int64_t row_bytes = mi355q_row_size(type, in.k);
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
                if (m.w_stride != row_bytes && m.m > 1) { mi355q_set_error("plan_create: the rows of a matrix must be contiguous (w_stride == row size): the loader copies whole row ranges"); return MI355Q_ERR_UNSUPPORTED; }
                p.w[n] = (const uint8_t *) m.w; p.y[n] = m.y; p.row_begin[n] = (int) rows;
                if (!paired || n == 0) {   // the stage's granules form one block indexed by concatenated row: matrix n starts at gran_count + rows
                    OutRange r = { m.y, m.m, sub_base + (size_t) rows, p.tag_off, in.y_id[j] };
                    outs.push_back(r);
                }
                rows += m.m; bytes += m.m * row_bytes; ++n;
            }
            if (rows > 0x7FFFFFF0) { mi355q_set_error("plan_create: too many rows"); return MI355Q_ERR_UNSUPPORTED; }
            gran_count += (size_t) ((rows + 3) & ~(int64_t) 3);
            for (int j = n; j < GEMV_MAX_MATS; ++j) p.row_begin[j] = 0x7FFFFFFF;
            p.total_rows = (int) rows; p.n_mats = n; p.type = type; p.k = (int) in.k; p.row_bytes = (int) row_bytes;
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
            if (paired) { rows = in.mats[0].m; p.total_rows = (int) rows; p.flags |= PLAN_F_PAIRED; p.x_unary |= in.y_unary << 8; }   // per-workgroup PAIRS; the granule block holds m elements
            int64_t rpw = (rows + n_cu - 1) / n_cu; if (rpw < 1) rpw = 1;
            p.rows_per_wg = (int) rpw;
            v.push_back(p); attn_of.push_back(-1);
            set |= tbit(type);
            const size_t colb = ((size_t) lds_col_bytes(fam, (int) in.k) + 15) & ~(size_t) 15;
            if (colb > img_max) img_max = colb;
            if ((size_t) row_bytes > row_max) row_max = (size_t) row_bytes;
            first = false;
        }
    }
    if (set == 0) { mi355q_set_error("plan_create: a plan needs at least one GEMV stage"); return MI355Q_ERR_UNSUPPORTED; }
    // LDS: [image | attention scratch][control block][row-slot terms][ring of np pages, a multiple of 4, 1024-aligned]
    const size_t lds_cap = 160 * 1024 - 64;
    const size_t ctl_off = (img_max + 63) & ~(size_t) 63;
    const size_t ring_off = (ctl_off + CB_BYTES + PLAN_TERM_BYTES + 1023) & ~(size_t) 1023;
    int np = ring_off + 16 * 1024 <= lds_cap ? (int) ((lds_cap - ring_off) / 4096) * 4 : 0;
    if (const char * e = getenv("MI355Q_PLAN_RING_PAGES")) { const int want = atoi(e) & ~3; if (want >= 16 && want < np) np = want; }   // (dev: A/B of the ring size)
    // the ring must hold a wave's item -- a row, or for a PAIRED stage both rows of a pair, 2 G rows apart -- with room to spare
    if (np < 16 || (size_t) np * 1024 < 3 * row_max + 8192) { mi355q_set_error("plan_create: k / attention window too large: no room for the weight ring in LDS"); return MI355Q_ERR_UNSUPPORTED; }
    for (auto & p : v) {
        if (p.kind != PLAN_K_GEMV) continue;
        int glog = 0;
        if (p.flags & PLAN_F_PAIRED) while (glog < 5 && (size_t) 2 * ((size_t) 2 << glog) * (size_t) p.row_bytes <= (size_t) np * 1024 / 3) ++glog;   // largest G with 2 G rows <= a third of the ring
        p.glog = glog;
        p.kf = PLAN_K_GEMV | ((p.flags & 0xFF) << 8) | (glog << 24);
    }

    Plan * pl = new Plan();
    pl->device = dev; pl->n_cu = n_cu; pl->n_stages = (int) v.size(); pl->set = set; pl->weight_bytes = bytes;
    pl->ctl_off = (int) ctl_off; pl->ring_off = (int) ring_off; pl->np = np; pl->lds_total = ring_off + (size_t) np * 1024;
    pl->even = (flags & MI355Q_FLAG_ROUND_EVEN) ? 1 : 0;
    pl->gran_count = gran_count + 4;
    pl->timeout_ticks = 100ull * 1000 * 20;                    // 20 ms of the 100 MHz real-time counter per wait
    if (const char * e = getenv("MI355Q_PLAN_TIMEOUT_MS")) pl->timeout_ticks = 100ull * 1000 * (unsigned long long) atoll(e);
    const void * kern = plan_kernel(set);
    if (!kern) { delete pl; mi355q_set_error("plan_create: this mix of weight types has no kernel instantiation"); return MI355Q_ERR_UNSUPPORTED; }
    int per_cu = 0;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_cap) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, GEMV_THREADS, pl->lds_total) != hipSuccess || per_cu < 1) {
        delete pl; mi355q_set_error("plan_create: persistent kernel does not fit a CU"); return MI355Q_ERR_HIP;
    }
    pl->grid = n_cu;                                           // one workgroup per CU, all co-resident
    bool ok = hipMalloc((void **) &pl->d_stages, v.size() * sizeof(PlanStage)) == hipSuccess &&
              hipMalloc((void **) &pl->d_sync, PLAN_SYNC_WORDS * sizeof(unsigned)) == hipSuccess &&
              hipMalloc((void **) &pl->d_gran, pl->gran_count * sizeof(Granule)) == hipSuccess &&
              (va.empty() || hipMalloc((void **) &pl->d_attn, va.size() * sizeof(AttnStage)) == hipSuccess);
    if (ok) {
        // patch the (offset + 1) placeholders into device pointers
        auto fix = [&](const Granule * p) { return p ? pl->d_gran + ((uintptr_t) p - 1) : nullptr; };
        auto fixs = [&](VecSrc & s) { s.gran = fix(s.gran); };
        for (auto & A : va) { fixs(A.q); fixs(A.k); fixs(A.v); A.part = (Granule *) fix(A.part); A.out_gran = (Granule *) fix(A.out_gran); }
        for (size_t i = 0; i < v.size(); ++i) {
            PlanStage & p = v[i];
            fixs(p.x0); fixs(p.x1); p.sum_gran = (Granule *) fix(p.sum_gran);
            p.yg = (Granule *) fix(p.yg);
            p.attn = attn_of[i] >= 0 ? pl->d_attn + attn_of[i] : nullptr;
        }
        ok = hipMemcpy(pl->d_stages, v.data(), v.size() * sizeof(PlanStage), hipMemcpyHostToDevice) == hipSuccess &&
             (va.empty() || hipMemcpy(pl->d_attn, va.data(), va.size() * sizeof(AttnStage), hipMemcpyHostToDevice) == hipSuccess) &&
             hipMemset(pl->d_sync, 0, PLAN_SYNC_WORDS * sizeof(unsigned)) == hipSuccess &&
             hipMemset(pl->d_gran, 0, pl->gran_count * sizeof(Granule)) == hipSuccess;
    }
    if (!ok) {
        if (pl->d_stages) (void) hipFree(pl->d_stages);
        if (pl->d_sync) (void) hipFree(pl->d_sync);
        if (pl->d_gran) (void) hipFree(pl->d_gran);
        if (pl->d_attn) (void) hipFree(pl->d_attn);
        delete pl; mi355q_set_error("plan_create: device allocation failed"); return MI355Q_ERR_HIP;
    }
    *out = (mi355q_plan *) pl;
    return MI355Q_OK;
}

int mi355q_ring_plan_run(mi355q_plan * plan, void * stream) {
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
    int ctl_off = pl->ctl_off, ring_off = pl->ring_off, np = pl->np;
    unsigned ring_magic = (unsigned) (0x100000000ull / ((unsigned long long) np * 1024ull));
    unsigned long long timeout = pl->timeout_ticks;
    { static const bool loader_only = getenv("MI355Q_PLAN_LOADER_ONLY") != nullptr; if (loader_only) even |= 2; }
    void * args[] = { (void *) &st, (void *) &n, (void *) &sync, (void *) &even, (void *) &timeout, (void *) &ctl_off, (void *) &ring_off, (void *) &np, (void *) &ring_magic, (void *) &epoch };
    // A PLAIN launch of one workgroup per CU (checked against the occupancy query at creation).  hipLaunchCooperativeKernel gives the same
    // residency and only adds a launch-time check of the grid size -- at +15-19 us of host time per launch, through a second (cooperative) HSA
    // queue whose teardown at process exit crashed inside the HSA runtime under rocprofv3 (MI355X_MICROARCH.md, coop-launch row).
    // Every wait is bounded, so a workgroup that found no CU (another persistent kernel holding them) ends in status() == 1, not in a hang.
    const hipError_t rc = hipLaunchKernel(plan_kernel(pl->set), dim3((unsigned) pl->grid), dim3(GEMV_THREADS), args, pl->lds_total, (hipStream_t) stream);
    if (rc != hipSuccess) { mi355q_set_error(hipGetErrorString(rc)); return MI355Q_ERR_HIP; }
    return MI355Q_OK;
}

/* dev / test hook: the number of runs the plan believes it has made (the epoch source); setting it close to the wrap exercises the granule reset */
int mi355q_ring_plan_debug_set_runs(mi355q_plan * plan, unsigned long long runs) {
    Plan * pl = (Plan *) plan;
    if (!pl) return MI355Q_ERR_SHAPE;
    pl->runs = runs;
    return MI355Q_OK;
}

/* diagnostics: the plan's 32 sync words (word 0: abort flag; words 1..7 of an aborted plan: which wait gave up -- kind, stage, workgroup, wave, two operands,
 * landed pages).  Synchronizes with the device. */
int mi355q_ring_plan_debug_words(mi355q_plan * plan, unsigned * out32) {
    Plan * pl = (Plan *) plan;
    if (!pl || !out32) return MI355Q_ERR_SHAPE;
    return hipMemcpy(out32, pl->d_sync, PLAN_SYNC_WORDS * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess ? MI355Q_OK : MI355Q_ERR_HIP;
}

/* 0 = healthy; 1 = a wait timed out (the plan is dead: destroy it).  Synchronizes with the device. */
int mi355q_ring_plan_status(mi355q_plan * plan) {
    Plan * pl = (Plan *) plan;
    if (!pl) return MI355Q_ERR_SHAPE;
    unsigned h[PLAN_SYNC_WORDS];
    if (hipMemcpy(h, pl->d_sync, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return MI355Q_ERR_HIP;
    return h[PLAN_SYNC_ABORT] ? 1 : 0;
}

int mi355q_ring_plan_status_async(mi355q_plan * plan, unsigned * host_flag, void * stream) {
    Plan * pl = (Plan *) plan;
    if (!pl || !host_flag) return MI355Q_ERR_SHAPE;
    if (hipMemcpyAsync(host_flag, pl->d_sync + PLAN_SYNC_ABORT, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t) stream) != hipSuccess) return MI355Q_ERR_HIP;
    return MI355Q_OK;
}

int64_t mi355q_ring_plan_weight_bytes(const mi355q_plan * plan) { return plan ? ((const Plan *) plan)->weight_bytes : 0; }
int     mi355q_ring_plan_launch_stages(const mi355q_plan * plan) { return plan ? ((const Plan *) plan)->n_stages : 0; }

int mi355q_ring_plan_destroy(mi355q_plan * plan) {
    Plan * pl = (Plan *) plan;
    if (!pl) return MI355Q_OK;
    (void) hipFree(pl->d_stages); (void) hipFree(pl->d_sync); (void) hipFree(pl->d_gran);
    if (pl->d_attn) (void) hipFree(pl->d_attn);
    delete pl;
    return MI355Q_OK;
}

} // extern "C"
