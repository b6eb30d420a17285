// plan.hip -- the token-generation chain as ONE persistent launch (MI355X-first replacement for the
// reference's "one kernel per node + CUDA graph" decode loop, ggml-cuda.cu:2470-2781).
//
// A PLAN is an ordered list of STAGES; a stage = up to 4 planar weight matrices of one type that share an
// activation vector x (N = 1), e.g. wq|wk of a layer, or ffn_gate|ffn_up.  One cooperative launch of
// #CU workgroups (16 waves each, one per CU) walks the whole list:
//
//   * the weight stream never stops at a matmul boundary.  Weights do not depend on activations, so
//     a wave requests the first chunks of its rows of stage s+1 as soon as it has finished its rows of
//     stage s -- BEFORE the grid barrier that orders the activations -- and they are in flight during
//     the barrier and the activation quantization.  (A per-matmul launch pays kernel boundary +
//     dispatch + ring prime + quantize prologue serially, ~5-8 us)
//   * the dependency between stages (y of one feeds, through the graph's glue ops, x of the next) is a
//     grid barrier in device memory: two-level arrive counters (no cache line is touched by more than
//     ~32 workgroups: one counter hammered by 256 pollers costs 5-20 us per barrier on this chip), relaxed
//     agent-scope atomics, polled by one wave.  Outputs are written with agent-scope (sc1, write-through)
//     stores and acknowledged (vmcnt) before the arrival; activations are read with sc1 loads, so no cache
//     invalidate / write-back fences (which would also drain the in-flight weight loads) are needed.
//   * a stage that reuses the previous stage's x (Q4_K wq|wk followed by Q6_K wv of a Q4_K_M layer)
//     needs neither barrier nor re-quantization: the LDS image stays.
//   * every spin has a wall-clock bound (s_memrealtime): on timeout the workgroup raises the plan's
//     sticky abort flag and returns, so the grid always drains.
//
// Arithmetic per row is EXACTLY gemv_fast.hip's (same chunk loaders / consumers, gemv_stream.cuh):
// bit-identical outputs to mi355q_mul_mat for the same matrices.
//
// Bound: HBM read of W.  Algorithmic bytes per launch = sum over stages, matrices of m * row_size(type, k).
#include <hip/hip_runtime.h>
#include <vector>

#include "gemv_stream.cuh"

namespace mi355q {

constexpr int PLAN_D_MAX = 8;                     // ring depth (1-KiB steps in flight per wave)
__host__ __device__ constexpr int plan_depth(int type) {   // Q5_K / Q6_K slots carry qh too (1.5 KiB per step): 6 steps are the bytes of 8 Q4_K steps
    return (type == MI355Q_TYPE_Q5_K || type == MI355Q_TYPE_Q6_K) ? 6 : PLAN_D_MAX;
}
enum { PLAN_F_BARRIER = 1, PLAN_F_NEW_X = 2 };
// Grid barrier state (u32 words; every counter on its own 128-byte line).  Two levels, so that no line is hammered
// by more than ~32 agents: workgroup i belongs to group i % 8 (the dispatcher deals workgroups round-robin to the 8
// XCDs, so a group is normally one XCD; nothing but speed depends on that).  Arrive = one atomic add on the group's
// counter.  The group's leader (workgroup g) polls the 8 group counters with one 8-lane load and then publishes
// release[g]; the other workgroups of the group poll only release[g].
constexpr int PLAN_GROUPS = 8, PLAN_LINE = 32;
enum { PLAN_SYNC_ARRIVE = 0, PLAN_SYNC_RELEASE = PLAN_GROUPS * PLAN_LINE, PLAN_SYNC_EXIT = 2 * PLAN_GROUPS * PLAN_LINE,
       PLAN_SYNC_ABORT = PLAN_SYNC_EXIT + PLAN_LINE, PLAN_SYNC_WORDS = PLAN_SYNC_ABORT + PLAN_LINE };

struct alignas(64) PlanStage {
    // -- what the streamers need per row, contiguous (arrives with a few scalar loads issued together) --
    const uint8_t * w[GEMV_MAX_MATS];
    int64_t         w_stride[GEMV_MAX_MATS];
    float *         y[GEMV_MAX_MATS];
    int             row_begin[GEMV_MAX_MATS];     // first concatenated row of each matrix (unused entries: INT_MAX)
    int             total_rows, rows_per_wg, k, n_mats;
    // -- the rest --
    const float *   x;
    int             type, flags, prime, pad;      // prime: ring slots requested before the barrier / quantization
};
typedef const __attribute__((address_space(4))) PlanStage * StageC;   // descriptors are read with scalar loads

// type sets a kernel instantiation can stream (register allocation is the max over the set)
constexpr unsigned tbit(int t) { return 1u << t; }
constexpr unsigned SET_K46  = tbit(MI355Q_TYPE_Q4_K) | tbit(MI355Q_TYPE_Q6_K);
constexpr unsigned SET_K456 = SET_K46 | tbit(MI355Q_TYPE_Q5_K);
constexpr unsigned SET_80   = tbit(MI355Q_TYPE_Q8_0) | tbit(MI355Q_TYPE_Q4_0);
constexpr unsigned SET_ALL  = SET_K456 | SET_80;

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

__device__ __forceinline__ bool plan_type_is_q8k(int t) { return t == MI355Q_TYPE_Q4_K || t == MI355Q_TYPE_Q5_K || t == MI355Q_TYPE_Q6_K; }

// A wave's own copy of the per-matrix fields of a stage descriptor, in SGPRs: read once per stage with independent
// scalar loads (one round trip).  Reading them per row instead (matrix index -> base -> stride: a dependent chain of
// scalar loads) costs microseconds whenever the descriptors miss the scalar cache.
struct StageW {
    const uint8_t * w0, * w1, * w2, * w3; int64_t ws0, ws1, ws2, ws3; float * y0, * y1, * y2, * y3; int rb1, rb2, rb3;
    __device__ __forceinline__ void load(StageC st) {
        w0 = st->w[0]; w1 = st->w[1]; w2 = st->w[2]; w3 = st->w[3];
        ws0 = st->w_stride[0]; ws1 = st->w_stride[1]; ws2 = st->w_stride[2]; ws3 = st->w_stride[3];
        y0 = st->y[0]; y1 = st->y[1]; y2 = st->y[2]; y3 = st->y[3];
        rb1 = st->row_begin[1]; rb2 = st->row_begin[2]; rb3 = st->row_begin[3];
    }
    // all fields are read BEFORE the selects (a select between field addresses would keep the struct in scratch memory)
    __device__ __forceinline__ const uint8_t * row_ptr(int r) const {
        const uint64_t a0 = (uint64_t) w0, a1 = (uint64_t) w1, a2 = (uint64_t) w2, a3 = (uint64_t) w3;
        const int64_t s0 = ws0, s1 = ws1, s2 = ws2, s3 = ws3;
        const int b1 = rb1, b2 = rb2, b3 = rb3;
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
    g.nb = (T == MI355Q_TYPE_Q8_0 || T == MI355Q_TYPE_Q4_0) ? k >> 5 : k >> 8;
    g.nchunks = row_chunks(T, k); g.steps = (g.nchunks + 63) >> 6;
    return g;
}

template <int T>
__device__ __forceinline__ void plan_issue(Chunk & slot, PlanCursor & ld, const StageW & st, const StageGeom & g, int lane) {
    if (ld.gr < g.r_hi) {                                     // wave-uniform
        if (64 * ld.s + lane < g.nchunks) chunk_load<T>(slot, ld.row, g.nb, ld.s, lane);
        if (++ld.s == g.steps) { ld.s = 0; ld.gr += GEMV_WAVES; if (ld.gr < g.r_hi) ld.row = st.row_ptr(ld.gr); }
    }
}

// request slots [from, to) of the ring for stage st (its loader cursor continues where it stands)
template <int T, int D>
__device__ __forceinline__ void plan_fill(Chunk (&ring)[D], int from, int to, PlanCursor & ld, const StageW & st, const StageGeom & g, int lane) {
#pragma unroll
    for (int d = 0; d < D; ++d) if (d >= from && d < to) plan_issue<T>(ring[d], ld, st, g, lane);
}

// consume this wave's rows of the stage; slot d holds item d, d+D, ... ; refills keep D items in flight
template <int T, int D>
__device__ __forceinline__ void plan_run(Chunk (&ring)[D], PlanCursor & ld, const StageW & st, const StageGeom & g, int r_lo, int wave, int lane, const ActView * av) {
    // the consumers' lane-invariant state (LDS offsets, shifts) is derived from an opaque copy of the lane id HERE, so
    // that it cannot be computed (and kept live, and spilled) before the barrier / quantization phase
    int lane_c = lane; asm volatile("" : "+v"(lane_c));
    int cs_gr = r_lo + wave, cs_s = 0;
    float acc[1] = { 0.0f };
    while (cs_gr < g.r_hi) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (cs_gr < g.r_hi) {                             // wave-uniform
                if (64 * cs_s + lane_c < g.nchunks) Consume<T, 1>::run(ring[d], cs_s, lane_c, av, acc);
                if (++cs_s == g.steps) {                      // row finished: reduce, store (agent-coherent), next row
                    const float t = wave_sum(acc[0]);
                    if (lane_c == 0) __hip_atomic_store(st.y_ptr(cs_gr), t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    acc[0] = 0.0f; cs_s = 0; cs_gr += GEMV_WAVES;
                }
                plan_issue<T>(ring[d], ld, st, g, lane);      // refill the slot just consumed
            }
        }
    }
}

// ---- activations: agent-coherent fetch (another XCD wrote them in this same launch) -------------
// Buffer loads with the sc1 (agent scope) cache policy: 16 bytes per lane, tracked by the compiler's vmcnt bookkeeping
// (so weight loads issued AFTER them can stay in flight while they are waited for), and out-of-range lanes read 0.
typedef unsigned int plan_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 plan_act_fetch(__amdgpu_buffer_rsrc_t xr, int span, int lane) {
    const plan_u4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, span * 1024 + 16 * lane, 0, 16 /* sc1 */);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// LDS visibility + workgroup barrier WITHOUT draining vmcnt: __syncthreads() fences every address space and so waits
// for the weight loads in flight; these fences name the LDS only (lgkmcnt), and -- unlike a bare s_barrier, which is
// IntrNoMem for the compiler -- they also keep the LDS accesses on their side of the barrier at compile time.
__device__ __forceinline__ void plan_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int FAM>
__device__ __forceinline__ void plan_quantize_span(const float4 v, int span, uint8_t * lds, int k, bool even, int lane) {
    if (span * 256 >= k) return;                               // wave-uniform
    if constexpr (FAM == FAM_Q8K) quantize_span_to_lds<FAM_Q8K, false>(v, span, lds, k, lane);
    else if (even)                quantize_span_to_lds<FAM_Q80, true>(v, span, lds, k, lane);
    else                          quantize_span_to_lds<FAM_Q80, false>(v, span, lds, k, lane);
}

// Called by wave 0 (all 64 lanes) of a workgroup: wait until every workgroup has arrived `arrivals` times.
// false on timeout / abort.
__device__ __forceinline__ bool plan_grid_wait(unsigned * sync, unsigned arrivals, unsigned grid, unsigned long long timeout_ticks, int lane) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned g = blockIdx.x % PLAN_GROUPS;
    const bool leader = blockIdx.x < PLAN_GROUPS;
    // leader: lane l < 8 watches group l (target = arrivals * size of group l); follower: lane 0 watches release[g]
    const unsigned l = (unsigned) lane < (unsigned) PLAN_GROUPS ? (unsigned) lane : 0u;
    const unsigned gsize = grid > l ? (grid - l + PLAN_GROUPS - 1) / PLAN_GROUPS : 0u;
    const unsigned * addr = leader ? sync + PLAN_SYNC_ARRIVE + l * PLAN_LINE : sync + PLAN_SYNC_RELEASE + g * PLAN_LINE;
    const unsigned target = leader ? arrivals * gsize : arrivals;
    const bool watch = leader ? lane < PLAN_GROUPS : lane == 0;
    unsigned spins = 0;
    for (;;) {
        unsigned v = target;
        if (watch) v = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__ballot(v < target) == 0ull) break;
        if ((++spins & 31u) == 0u) {
            unsigned ab = 0;
            if (lane == 0) ab = __hip_atomic_load(sync + PLAN_SYNC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ab = (unsigned) __builtin_amdgcn_readfirstlane((int) ab);
            if (ab != 0u) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
                if (lane == 0) __hip_atomic_store(sync + PLAN_SYNC_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(2);
    }
    if (leader && lane == 0) __hip_atomic_store(sync + PLAN_SYNC_RELEASE + g * PLAN_LINE, arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// One stage, start to finish, for weight type T.  Deliberately NOT inlined: each type's streamer gets its own
// register allocation (inlined side by side, the lane-invariant state of both types stays live and spills).
//   prime (weights in flight) -> [grid barrier] -> [quantize x -> LDS] -> stream rows -> [acknowledge y, arrive]
struct StageCtx {
    uint8_t * lds; uint8_t * stage_lds; int * ctl; unsigned * sync; unsigned long long timeout;
    unsigned grid, arrivals; int even, next_barrier, stage;
};
enum { CTL_OK = 0, CTL_WAVES = 1 };

template <int T>
static __device__ __forceinline__ bool plan_stage(StageC st, const StageCtx & c) {
    // an opaque copy of the lane id per stage: everything derived from it is recomputed here (a few VALU ops) instead
    // of being hoisted out of the stage loop for BOTH types' loaders, quantizers and pollers and kept live (and spilled)
    int lane = lane_id(); asm volatile("" : "+v"(lane));
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int k = st->k, flags = st->flags;
    int r_lo = (int) blockIdx.x * st->rows_per_wg, r_hi = r_lo + st->rows_per_wg;
    if (r_hi > st->total_rows) r_hi = st->total_rows;
    const StageGeom g = plan_geom<T>(k, r_hi);

    constexpr int PLAN_D = plan_depth(T);
    Chunk ring[PLAN_D];
    PlanCursor ld;
    ld.gr = r_lo + wave; ld.s = 0; ld.row = nullptr;
    StageW sw; sw.load(st);
    if (ld.gr < r_hi) ld.row = sw.row_ptr(ld.gr);
    PLAN_STAMP(0);
    plan_fill<T, PLAN_D>(ring, 0, st->prime, ld, sw, g, lane);                 // weights start flowing before anything else
    PLAN_STAMP(1);

    if (flags & PLAN_F_BARRIER) {                             // every workgroup has finished (and made visible) all earlier stages
        if (wave == 0) {
            const bool ok = plan_grid_wait(c.sync, c.arrivals, c.grid, c.timeout, lane);
            if (lane == 0) c.ctl[CTL_OK] = ok ? 1 : 0;
        }
        plan_lds_barrier();
        if (!c.ctl[CTL_OK]) return false;
    } else if (flags & PLAN_F_NEW_X) {
        plan_lds_barrier();                                   // all waves are done with the previous LDS image
    }
    PLAN_STAMP(2);
    if (flags & PLAN_F_NEW_X) {
        // Spans of 256 activations are dealt round-robin to the waves, four in flight per wave (one memory round trip for
        // k <= 16384).  Agent-scope (sc1) buffer loads: another XCD wrote them during this launch; lanes beyond k read 0.
        constexpr int FAM = (T == MI355Q_TYPE_Q8_0 || T == MI355Q_TYPE_Q4_0) ? FAM_Q80 : FAM_Q8K;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *) st->x, 0, k * 4, 0x00020000);
        const int spans = (k + 255) >> 8;
        // The raw f32 spans are parked in a per-wave LDS staging area right after they arrive: holding them in VGPRs
        // through the quantizer (on top of the ring) does not fit 128 registers.
        float4 * stg = (float4 *) (c.stage_lds + wave * (4 * 1024)) + lane;
        auto quantize4 = [&](int span0) {
#pragma unroll 1
            for (int i = 0; i < 4; ++i) {                      // same-lane round trip through LDS: no barrier needed
                const float4 v = stg[64 * i];
                plan_quantize_span<FAM>(v, span0 + i * GEMV_WAVES, c.lds, k, c.even != 0, lane);
            }
        };
        {   // first pass (all of k <= 16384)
            const float4 x0 = plan_act_fetch(xr, wave, lane), x1 = plan_act_fetch(xr, wave + GEMV_WAVES, lane);
            const float4 x2 = plan_act_fetch(xr, wave + 2 * GEMV_WAVES, lane), x3 = plan_act_fetch(xr, wave + 3 * GEMV_WAVES, lane);
            stg[0] = x0; stg[64] = x1; stg[128] = x2; stg[192] = x3;
        }
        quantize4(wave);
#pragma unroll 1
        for (int span = wave + 4 * GEMV_WAVES; span < spans; span += 4 * GEMV_WAVES) {
            const float4 x0 = plan_act_fetch(xr, span, lane), x1 = plan_act_fetch(xr, span + GEMV_WAVES, lane);
            const float4 x2 = plan_act_fetch(xr, span + 2 * GEMV_WAVES, lane), x3 = plan_act_fetch(xr, span + 3 * GEMV_WAVES, lane);
            stg[0] = x0; stg[64] = x1; stg[128] = x2; stg[192] = x3;
            quantize4(span);
        }
        plan_lds_barrier();
    }
    plan_fill<T, PLAN_D>(ring, st->prime, PLAN_D, ld, sw, g, lane);                // top the ring up (a no-op when prime == depth)
    PLAN_STAMP(3);
    ActView av[1];
    av[0].base = c.lds; av[0].k = k;
    plan_run<T, PLAN_D>(ring, ld, sw, g, r_lo, wave, lane, av);
    PLAN_STAMP(4);

    if (c.next_barrier) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's y stores have reached the coherence point (its ring is empty)
        PLAN_STAMP(5);
        if (lane == 0) {                                      // the last of the 16 waves arrives for the workgroup; nobody waits here
            const int done = atomicAdd(&c.ctl[CTL_WAVES], 1) + 1;
            if (done == GEMV_WAVES * (int) (c.arrivals + 1))
                __hip_atomic_fetch_add(c.sync + PLAN_SYNC_ARRIVE + (blockIdx.x % PLAN_GROUPS) * PLAN_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    return true;
}

template <unsigned SET>
__global__ void __launch_bounds__(GEMV_THREADS)
k_plan(const PlanStage * stages_g, int n_stages, unsigned * sync, int even, unsigned long long timeout_ticks, int lds_image_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    StageC stages = (StageC) stages_g;
    if (__hip_atomic_load(sync + PLAN_SYNC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;   // sticky: a plan that timed out stays dead

    StageCtx c;
    c.lds = lds; c.ctl = (int *) (lds + lds_image_bytes); c.stage_lds = lds + lds_image_bytes + 64; c.sync = sync; c.timeout = timeout_ticks;
    c.grid = gridDim.x; c.arrivals = 0; c.even = even;
    if (threadIdx.x == 0) { c.ctl[CTL_OK] = 1; c.ctl[CTL_WAVES] = 0; }
    plan_lds_barrier();

#pragma unroll 1
    for (int s = 0; s < n_stages; ++s) {
        StageC st = stages + s;
        c.stage = s;
        c.next_barrier = (s + 1 < n_stages && ((st + 1)->flags & PLAN_F_BARRIER)) ? 1 : 0;
        bool ok = true;
        switch (st->type) {
        case MI355Q_TYPE_Q4_K: if constexpr ((SET & tbit(MI355Q_TYPE_Q4_K)) != 0) ok = plan_stage<MI355Q_TYPE_Q4_K>(st, c); break;
        case MI355Q_TYPE_Q5_K: if constexpr ((SET & tbit(MI355Q_TYPE_Q5_K)) != 0) ok = plan_stage<MI355Q_TYPE_Q5_K>(st, c); break;
        case MI355Q_TYPE_Q6_K: if constexpr ((SET & tbit(MI355Q_TYPE_Q6_K)) != 0) ok = plan_stage<MI355Q_TYPE_Q6_K>(st, c); break;
        case MI355Q_TYPE_Q8_0: if constexpr ((SET & tbit(MI355Q_TYPE_Q8_0)) != 0) ok = plan_stage<MI355Q_TYPE_Q8_0>(st, c); break;
        case MI355Q_TYPE_Q4_0: if constexpr ((SET & tbit(MI355Q_TYPE_Q4_0)) != 0) ok = plan_stage<MI355Q_TYPE_Q4_0>(st, c); break;
        default: break;
        }
        if (!ok) return;
        c.arrivals += (unsigned) c.next_barrier;
    }
    // the last workgroup out re-arms the counters for the next launch
    asm volatile("s_waitcnt vmcnt(0)");
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(sync + PLAN_SYNC_EXIT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == c.grid - 1) {
            for (int g = 0; g < PLAN_GROUPS; ++g) {
                __hip_atomic_store(sync + PLAN_SYNC_ARRIVE + g * PLAN_LINE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sync + PLAN_SYNC_RELEASE + g * PLAN_LINE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __hip_atomic_store(sync + PLAN_SYNC_EXIT, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct Plan {
    int           device = 0, n_cu = 0, grid = 0, n_stages = 0, even = 0;
    unsigned      set = 0;
    size_t        lds_bytes = 0;
    int64_t       weight_bytes = 0;
    PlanStage *   d_stages = nullptr;
    unsigned *    d_sync = nullptr;
};

int gemv_fast_family(int type);

static const void * plan_kernel(unsigned set) {
    if ((set & ~SET_K46) == 0)  return (const void *) k_plan<SET_K46>;
    if ((set & ~SET_K456) == 0) return (const void *) k_plan<SET_K456>;
    if ((set & ~SET_80) == 0)   return (const void *) k_plan<SET_80>;
    return (const void *) k_plan<SET_ALL>;
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

int mi355q_plan_create(mi355q_plan ** out, const mi355q_stage * stages, int n_stages, int flags) {
    if (!out || !stages || n_stages < 1) { mi355q_set_error("plan_create: null argument / no stages"); return MI355Q_ERR_SHAPE; }
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { mi355q_set_error("plan_create: no device"); return MI355Q_ERR_HIP; }
    if (!prop.cooperativeLaunch) { mi355q_set_error("plan_create: device lacks cooperative launch"); return MI355Q_ERR_UNSUPPORTED; }
    const int n_cu = prop.multiProcessorCount;

    // split every stage per weight type (a stage of mixed types becomes consecutive sub-stages on the same x)
    std::vector<PlanStage> v;
    unsigned set = 0; size_t lds_max = 0; int64_t bytes = 0;
    for (int s = 0; s < n_stages; ++s) {
        const mi355q_stage & in = stages[s];
        if (in.n_mats < 1 || in.n_mats > GEMV_MAX_MATS || !in.x || in.k <= 0 || ((uintptr_t) in.x & 3)) { mi355q_set_error("plan_create: bad stage"); return MI355Q_ERR_SHAPE; }
        bool done[GEMV_MAX_MATS] = { false, false, false, false };
        bool first = true;
        for (int i = 0; i < in.n_mats; ++i) {
            if (done[i]) continue;
            const int type = in.mats[i].type;
            const int fam = gemv_fast_family(type);
            if (fam < 0 || !mi355q_weights_are_planar(type, in.k)) { mi355q_set_error("plan_create: weight type / k has no planar streaming kernel"); return MI355Q_ERR_UNSUPPORTED; }
            PlanStage p = {};
            int64_t rows = 0; int n = 0;
            for (int j = i; j < in.n_mats; ++j) {
                if (done[j] || in.mats[j].type != type) continue;
                done[j] = true;
                const mi355q_mat & m = in.mats[j];
                if (!m.w || !m.y || m.m < 0) { mi355q_set_error("plan_create: null matrix pointer"); return MI355Q_ERR_SHAPE; }
                if (((uintptr_t) m.w | (uintptr_t) m.w_stride) & 15) { mi355q_set_error("plan_create: planar rows must be 16-byte aligned"); return MI355Q_ERR_ALIGN; }
                if (m.w_stride < mi355q_row_size(type, in.k)) { mi355q_set_error("plan_create: w_stride smaller than a row"); return MI355Q_ERR_SHAPE; }
                p.w[n] = (const uint8_t *) m.w; p.y[n] = m.y; p.w_stride[n] = m.w_stride; p.row_begin[n] = (int) rows;
                rows += m.m; bytes += m.m * mi355q_row_size(type, in.k); ++n;
            }
            if (rows > 0x7FFFFFF0) { mi355q_set_error("plan_create: too many rows"); return MI355Q_ERR_UNSUPPORTED; }
            for (int j = n; j < GEMV_MAX_MATS; ++j) p.row_begin[j] = 0x7FFFFFFF;
            p.x = in.x; p.total_rows = (int) rows; p.n_mats = n; p.type = type; p.k = (int) in.k;
            // barrier + fresh activations at the head of a dependent stage; same-x continuation otherwise
            const bool depends = (in.flags & MI355Q_STAGE_DEPENDS) && !v.empty();
            const bool reuse = !first || (!depends && !v.empty() && v.back().x == in.x && v.back().k == (int) in.k &&
                                          gemv_fast_family(v.back().type) == fam);         // the LDS image of x is still valid
            p.flags = reuse ? 0 : (PLAN_F_NEW_X | (depends ? PLAN_F_BARRIER : 0));
            int64_t rpw = (rows + n_cu - 1) / n_cu; if (rpw < 1) rpw = 1;
            p.rows_per_wg = (int) rpw;
            p.prime = plan_depth(type);
            v.push_back(p);
            set |= tbit(type);
            const size_t colb = ((size_t) lds_col_bytes(fam, (int) in.k) + 15) & ~(size_t) 15;
            if (colb > lds_max) lds_max = colb;
            first = false;
        }
    }
    if (lds_max + 64 + GEMV_WAVES * 4096 > 159 * 1024) { mi355q_set_error("plan_create: k too large for the LDS activation image"); return MI355Q_ERR_UNSUPPORTED; }

    Plan * pl = new Plan();
    pl->device = dev; pl->n_cu = n_cu; pl->n_stages = (int) v.size(); pl->set = set; pl->lds_bytes = lds_max; pl->weight_bytes = bytes;
    pl->even = (flags & MI355Q_FLAG_ROUND_EVEN) ? 1 : 0;
    const void * kern = plan_kernel(set);
    int per_cu = 0;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, GEMV_THREADS, lds_max + 64 + GEMV_WAVES * 4096) != hipSuccess || per_cu < 1) {
        delete pl; mi355q_set_error("plan_create: persistent kernel does not fit a CU"); return MI355Q_ERR_HIP;
    }
    pl->grid = n_cu;                                           // one workgroup per CU, all co-resident (checked again by the cooperative launch)
    if (hipMalloc((void **) &pl->d_stages, v.size() * sizeof(PlanStage)) != hipSuccess ||
        hipMalloc((void **) &pl->d_sync, PLAN_SYNC_WORDS * sizeof(unsigned)) != hipSuccess ||
        hipMemcpy(pl->d_stages, v.data(), v.size() * sizeof(PlanStage), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(pl->d_sync, 0, PLAN_SYNC_WORDS * sizeof(unsigned)) != hipSuccess) {
        if (pl->d_stages) (void) hipFree(pl->d_stages);
        if (pl->d_sync) (void) hipFree(pl->d_sync);
        delete pl; mi355q_set_error("plan_create: device allocation failed"); return MI355Q_ERR_HIP;
    }
    *out = (mi355q_plan *) pl;
    return MI355Q_OK;
}

int mi355q_plan_run(mi355q_plan * plan, void * stream) {
    Plan * pl = (Plan *) plan;
    if (!pl) { mi355q_set_error("plan_run: null plan"); return MI355Q_ERR_SHAPE; }
    const PlanStage * st = pl->d_stages; int n = pl->n_stages; unsigned * sync = pl->d_sync; int even = pl->even;
    int image = (int) pl->lds_bytes;
    unsigned long long timeout = 100ull * 1000 * 20;          // 20 ms of the 100 MHz real-time counter per barrier
    if (const char * e = getenv("MI355Q_PLAN_TIMEOUT_MS")) timeout = 100ull * 1000 * (unsigned long long) atoll(e);
    void * args[] = { (void *) &st, (void *) &n, (void *) &sync, (void *) &even, (void *) &timeout, (void *) &image };
    const hipError_t rc = hipLaunchCooperativeKernel(plan_kernel(pl->set), dim3((unsigned) pl->grid), dim3(GEMV_THREADS), args,
                                                     pl->lds_bytes + 64 + GEMV_WAVES * 4096, (hipStream_t) stream);
    if (rc != hipSuccess) { mi355q_set_error(hipGetErrorString(rc)); return MI355Q_ERR_HIP; }
    return MI355Q_OK;
}

/* 0 = healthy; 1 = a grid barrier timed out (the plan is dead: destroy it).  Synchronizes with the device. */
int mi355q_plan_status(mi355q_plan * plan) {
    Plan * pl = (Plan *) plan;
    if (!pl) return MI355Q_ERR_SHAPE;
    static unsigned h[PLAN_SYNC_WORDS];
    if (hipMemcpy(h, pl->d_sync, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return MI355Q_ERR_HIP;
    return h[PLAN_SYNC_ABORT] ? 1 : 0;
}

int64_t mi355q_plan_weight_bytes(const mi355q_plan * plan) { return plan ? ((const Plan *) plan)->weight_bytes : 0; }
int     mi355q_plan_launch_stages(const mi355q_plan * plan) { return plan ? ((const Plan *) plan)->n_stages : 0; }

int mi355q_plan_destroy(mi355q_plan * plan) {
    Plan * pl = (Plan *) plan;
    if (!pl) return MI355Q_OK;
    (void) hipFree(pl->d_stages); (void) hipFree(pl->d_sync);
    delete pl;
    return MI355Q_OK;
}

} // extern "C"
