// gemv_fast.hip -- the token-generation hot loop: y[n][m] = W_q[m][k] . x[n][k], n <= 8, for
// PLANAR device rows (see mi355q_common.h).  Replaces the reference's quantize_q8_1 + mul_mat_vec_q
// pair (ggml-cuda/quantize.cu:4-38, mmvq.cu:130-288) with ONE launch, designed for gfx950:
//
//  * fused prologue: every workgroup quantizes the activation columns itself (Q8_K / Q8_0, exactly
//    the CPU's arithmetic, act_quant.cuh) into LDS -- no separate quantize launch, no HBM round trip;
//    the f32 activations (<= 8 x 56 KiB) come out of L2.  The first weight loads are already in
//    flight while this runs.
//  * one wave = one weight row at a time; a row is cut into STEPS of 64 payload chunks of 16 bytes:
//    in a step lane l owns chunk 64*s + l, so a wave-wide global_load_dwordx4 reads 1 KiB of contiguous
//    packed quants (non-temporal: weights are streamed exactly once); the small per-block fields come
//    from their own dense planes.  Lane roles inside a block (which sub-block, which nibble half) are
//    loop-invariant because 64 is a multiple of the 8 chunks of a block.
//  * a ring of D steps per wave is kept in flight across row boundaries (the (row, step) items of a
//    wave form one flat stream): consume item i, then immediately re-issue the load of item i+D.
//    With <=128 VGPRs there are 16 waves per CU, i.e. ~16*D KiB of weight reads in flight per CU.
//  * nibble / 6-bit unpack with full-dword bit ops, v_dot4_i32_i8 against int8 activations from LDS
//    (ds_read_b128), exact int32 block sums, f32 scale, 64-lane shuffle reduction per row.
//  * several matrices that share the activations (wq/wk/wv, gate/up) run in ONE launch (row ranges
//    are concatenated; the type switch is wave-uniform), removing launch gaps from the token loop.
//
// Bound: HBM read of W.  Algorithmic bytes per launch = sum_i m_i * row_size(type_i, k).
#include "gemv_stream.cuh"

namespace mi355q {

struct GemvMat {
    const uint8_t * w;
    float *         y;
    int64_t         w_stride;
    int64_t         y_stride;   // bytes between activation columns in y
    int64_t         row_begin;  // first concatenated row index of this matrix
    int             type;
    int             pad;
};

#ifdef MI355Q_STAMPS
// Diagnostic build only (libmi355q_dbg.so, -DMI355Q_STAMPS): lane 0 of every wave records 100 MHz wall-clock
// stamps at fixed points into g_stamps[(block*WAVES + wave)*8 + i].  The product library contains none of this.
__device__ unsigned long long * g_stamps = nullptr;
#define MI355Q_STAMP(i) do { if (g_stamps && lane == 0) g_stamps[((size_t) blockIdx.x * GEMV_WAVES + wave) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MI355Q_STAMP(i) do { } while (0)
#endif

struct GemvArgs {
    GemvMat       mats[GEMV_MAX_MATS];
    const float * x;
    int64_t       x_stride;     // bytes
    int64_t       total_rows;
    int64_t       rows_per_wg;
    int           n_mats;
    int           k;
    int           x_vec;        // x rows 16-byte aligned
    int           pad;
    // MUL_MAT_ID mode (ids != nullptr): blockIdx.y = (token t, slot u) pair; one matrix, one column.
    //   W = mats[0].w + ids[t][u]*expert_stride ; x = x + t*x_stride2 + (u % x_ne1)*x_stride ; y = mats[0].y + pair*m*4
    const int32_t * ids;
    int64_t       ids_stride;   // bytes between token rows of ids
    int64_t       expert_stride;
    int64_t       x_stride2;
    int           n_used;
    int           x_ne1;
    int           n_expert;
    int           pad2;
};


// ------------------------------------------------------------------------------------------------
// Kernel.  WT = the single weight type of the launch (compile time), or WT_MIXED.
//
//  The rows of all matrices of the launch (all of type WT) form one concatenated range; each wave walks its
//  rows (r_begin+wave, +WAVES, ...) as one flat stream of (row, step) items with a ring of D items in
//  flight.  The first D items are requested BEFORE the activation quantization, so they fly during it.
//  (A launch of mixed types, e.g. Q4_K wq/wk + Q6_K wv of a Q4_K_M layer, is split per type by the host.)
// All cursor state is wave-uniform (SGPRs); descriptor fields are read with constant kernarg indices only,
// so they arrive in one batch of scalar loads at kernel entry.
// ------------------------------------------------------------------------------------------------

struct MatSel { const uint8_t * w; float * y; int64_t w_stride, y_stride; int rb, re, type; };

// descriptor of matrix i (i wave-uniform) through constant-index selects
template <bool MULTI>
__device__ __forceinline__ MatSel select_mat(const GemvArgs & a, int i) {
    MatSel m = { a.mats[0].w, a.mats[0].y, a.mats[0].w_stride, a.mats[0].y_stride, 0, (int) a.total_rows, a.mats[0].type };
    if constexpr (MULTI) {
        if (a.n_mats > 1) m.re = (int) a.mats[1].row_begin;
#pragma unroll
        for (int j = 1; j < GEMV_MAX_MATS; ++j) {
            const bool is = i == j;
            m.w = is ? a.mats[j].w : m.w; m.y = is ? a.mats[j].y : m.y;
            m.w_stride = is ? a.mats[j].w_stride : m.w_stride; m.y_stride = is ? a.mats[j].y_stride : m.y_stride;
            m.rb = is ? (int) a.mats[j].row_begin : m.rb; m.type = is ? a.mats[j].type : m.type;
            const int nxt = (j + 1 < GEMV_MAX_MATS && j + 1 < a.n_mats) ? (int) a.mats[j + 1 < GEMV_MAX_MATS ? j + 1 : j].row_begin : (int) a.total_rows;
            m.re = is ? nxt : m.re;
        }
    }
    return m;
}
// index of the matrix that owns concatenated row gr
template <bool MULTI>
__device__ __forceinline__ int mat_of_row(const GemvArgs & a, int gr) {
    int mi = 0;
    if constexpr (MULTI) {
#pragma unroll
        for (int j = 1; j < GEMV_MAX_MATS; ++j) if (j < a.n_mats && gr >= (int) a.mats[j].row_begin) mi = j;
    }
    return mi;
}

// Stream the concatenated rows [r_lo, r_hi) restricted to this wave (r_lo + wave, + WAVES, ...), all of type T.
// PRIME_ONLY / RUN_ONLY let the caller put the activation prologue between the two halves.
template <int T, int NCOLS, int D, bool MULTI>
struct Streamer {
    int ld_gr, ld_s, cs_gr, cs_s, r_hi, nb, nchunks, steps;
    const uint8_t * ld_row;
    float * cs_y; int64_t cs_ys;
    Chunk ring[D];

    __device__ __forceinline__ void resolve_ld(const GemvArgs & a, int64_t w_off) {
        const MatSel m = select_mat<MULTI>(a, mat_of_row<MULTI>(a, ld_gr));
        ld_row = m.w + w_off + (int64_t) (ld_gr - m.rb) * m.w_stride;
    }
    __device__ __forceinline__ void resolve_cs(const GemvArgs & a, int64_t y_off) {
        const MatSel m = select_mat<MULTI>(a, mat_of_row<MULTI>(a, cs_gr));
        cs_y = (float *) ((char *) m.y + y_off) + (cs_gr - m.rb); cs_ys = m.y_stride;
    }
    __device__ __forceinline__ void issue(Chunk & slot, const GemvArgs & a, int64_t w_off, int lane) {
        if (ld_gr < r_hi) {                                   // wave-uniform
            if (64 * ld_s + lane < nchunks) chunk_load<T>(slot, ld_row, nb, ld_s, lane);
            if (++ld_s == steps) { ld_s = 0; ld_gr += GEMV_WAVES; if (ld_gr < r_hi) resolve_ld(a, w_off); }
        }
    }
    __device__ __forceinline__ void prime(const GemvArgs & a, int r_lo, int r_hi_, int k, int wave, int lane, int64_t w_off, int64_t y_off) {
        r_hi = r_hi_;
        nb = (T == MI355Q_TYPE_Q8_0 || T == MI355Q_TYPE_Q4_0 || T == MI355Q_TYPE_IQ4_NL) ? k >> 5 : k >> 8;
        nchunks = row_chunks(T, k); steps = (nchunks + 63) >> 6;
        ld_gr = cs_gr = r_lo + wave; ld_s = cs_s = 0;
        if (ld_gr < r_hi) { resolve_ld(a, w_off); resolve_cs(a, y_off); }
#pragma unroll
        for (int d = 0; d < D; ++d) issue(ring[d], a, w_off, lane);
    }
    __device__ __forceinline__ void run(const GemvArgs & a, const ActView * av, int lane, int64_t w_off, int64_t y_off) {
        float acc[NCOLS];
#pragma unroll
        for (int n = 0; n < NCOLS; ++n) acc[n] = 0.0f;
        while (cs_gr < r_hi) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (cs_gr < r_hi) {                           // wave-uniform
                    if (64 * cs_s + lane < nchunks) Consume<T, NCOLS>::run(ring[d], cs_s, lane, av, acc);
                    if (++cs_s == steps) {                    // row finished: reduce, store, next row
#pragma unroll
                        for (int n = 0; n < NCOLS; ++n) {
                            const float t = wave_sum(acc[n]);
                            if (lane == 0) *(float *) ((char *) cs_y + (int64_t) n * cs_ys) = t;
                            acc[n] = 0.0f;
                        }
                        cs_s = 0; cs_gr += GEMV_WAVES;
                        if (cs_gr < r_hi) resolve_cs(a, y_off);
                    }
                    issue(ring[d], a, w_off, lane);           // refill the slot just consumed
                }
            }
        }
    }
};

template <int FAM, int WT, int NCOLS, int D, bool ROUND_EVEN, bool MULTI>
__global__ void __launch_bounds__(GEMV_THREADS)
k_gemv_fast(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));   // wave-uniform -> SGPR
    const int k    = a.k;
    const int colb = (lds_col_bytes(FAM, k) + 15) & ~15;
    MI355Q_STAMP(0);

    const char * xbase = (const char *) a.x;
    int64_t w_off = 0, y_off = 0;
    if (a.ids) {                                              // MoE: resolve this pair's expert on the device
        const int pair = (int) blockIdx.y, t = pair / a.n_used, u = pair - t * a.n_used;
        const int e = *(const int32_t *) ((const char *) a.ids + (int64_t) t * a.ids_stride + 4 * u);
        if (e < 0 || e >= a.n_expert) {                       // the reference asserts on such an id; on every device path the pair's row becomes NaN
            const int rb = (int) blockIdx.x * (int) a.rows_per_wg, re = min(rb + (int) a.rows_per_wg, (int) a.total_rows);
            float * yp = (float *) ((char *) a.mats[0].y + (int64_t) pair * a.total_rows * 4);
            for (int r = rb + (int) threadIdx.x; r < re; r += GEMV_THREADS) yp[r] = __int_as_float(0x7FC00000);
            return;
        }
        xbase += (int64_t) t * a.x_stride2 + (int64_t) (u % a.x_ne1) * a.x_stride;
        w_off = (int64_t) e * a.expert_stride;
        y_off = (int64_t) pair * a.total_rows * 4;
    }

    // this workgroup's contiguous range of the concatenated rows
    const int r_begin = (int) blockIdx.x * (int) a.rows_per_wg;
    int       r_end   = r_begin + (int) a.rows_per_wg;
    if (r_end > (int) a.total_rows) r_end = (int) a.total_rows;

    ActView av[NCOLS];
#pragma unroll
    for (int n = 0; n < NCOLS; ++n) { av[n].base = lds + n * colb; av[n].k = k; }

    Streamer<WT, NCOLS, D, MULTI> st;
    MI355Q_STAMP(1);
    // order matters (vmcnt retires in issue order): activation loads first, then the weight ring
    const float4 c0 = act_fetch((const float *) xbase, wave, k, a.x_vec, lane);
    const float4 c1 = act_fetch((const float *) xbase, wave + GEMV_WAVES, k, a.x_vec, lane);
    st.prime(a, r_begin, r_end, k, wave, lane, w_off, y_off);              // first D items in flight ...
    MI355Q_STAMP(2);
    quantize_columns_to_lds<FAM, NCOLS, ROUND_EVEN>(lds, colb, xbase, a.x_stride, k, a.x_vec, wave, lane, c0, c1);   // ... while the activations are quantized
    MI355Q_STAMP(3);
    st.run(a, av, lane, w_off, y_off);
    MI355Q_STAMP(5);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int family_of(int type) {
    switch (type) {
    case MI355Q_TYPE_Q4_K: case MI355Q_TYPE_Q5_K: case MI355Q_TYPE_Q6_K: case MI355Q_TYPE_IQ4_XS: return FAM_Q8K;
    case MI355Q_TYPE_Q8_0: case MI355Q_TYPE_Q4_0: case MI355Q_TYPE_IQ4_NL: return FAM_Q80;
    default: return -1;
    }
}

int gemv_fast_family(int type) { return family_of(type); }

// largest number of activation columns whose LDS image fits in the CU's 160 KiB (one workgroup per CU), <= 8
int gemv_fast_max_cols(int type, int64_t k) {
    const int fam = family_of(type);
    if (fam < 0) return 0;
    const int colb = (lds_col_bytes(fam, (int) k) + 15) & ~15;
    const int n = (160 * 1024 - 1024) / colb;
    return n > 8 ? 8 : n;
}

template <int FAM, int WT, int NCOLS, int D, bool EVEN, bool MULTI>
static int launch_one_m(const GemvArgs & a, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    auto kern = k_gemv_fast<FAM, WT, NCOLS, D, EVEN, MULTI>;
    static bool lds_enabled[64] = {};                      // per kernel instantiation AND per device (the plugin drives every GPU from one process)
    if (lds_bytes > 48 * 1024) {
        int dev = 0; (void) hipGetDevice(&dev); dev = dev >= 0 && dev < 64 ? dev : 0;
        if (!lds_enabled[dev]) {
            if (hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return MI355Q_ERR_HIP;
            lds_enabled[dev] = true;
        }
    }
    hipLaunchKernelGGL(kern, grid, dim3(GEMV_THREADS), lds_bytes, stream, a);
    return hipGetLastError() == hipSuccess ? MI355Q_OK : MI355Q_ERR_HIP;
}

template <int FAM, int WT, int NCOLS, int D, bool EVEN>
static int launch_one(const GemvArgs & a, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    if (a.n_mats > 1) return launch_one_m<FAM, WT, NCOLS, D, EVEN, true>(a, grid, lds_bytes, stream);
    return launch_one_m<FAM, WT, NCOLS, D, EVEN, false>(a, grid, lds_bytes, stream);
}

// ring depth D (1-KiB steps in flight per wave): 8 for one column (a whole K=14336 row), shallower as columns eat VGPRs.
// Every N=1 launch of a token of one weight type runs the SAME kernel, so its code stays hot in the
// instruction cache across the ~130 launches of a token.
template <int FAM, int WT, bool EVEN>
static int launch_cols(const GemvArgs & a, int ncols, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    switch (ncols) {
    // (a Q5_K ring slot is 12 registers -- quants, qh bytes, header: 8 of them in flight spill inside the streaming loop, 1.9 TB/s)
    case 1: return launch_one<FAM, WT, 1, WT == MI355Q_TYPE_Q5_K ? 6 : 8, EVEN>(a, grid, lds_bytes, stream);
    case 2: return launch_one<FAM, WT, 2, WT == MI355Q_TYPE_Q5_K ? 4 : 6, EVEN>(a, grid, lds_bytes, stream);
    case 3: return launch_one<FAM, WT, 3, 3, EVEN>(a, grid, lds_bytes, stream);
    case 4: return launch_one<FAM, WT, 4, 3, EVEN>(a, grid, lds_bytes, stream);
    case 5: return launch_one<FAM, WT, 5, 2, EVEN>(a, grid, lds_bytes, stream);
    case 6: return launch_one<FAM, WT, 6, 2, EVEN>(a, grid, lds_bytes, stream);
    case 7: return launch_one<FAM, WT, 7, 2, EVEN>(a, grid, lds_bytes, stream);
    case 8: return launch_one<FAM, WT, 8, 2, EVEN>(a, grid, lds_bytes, stream);
    default: return MI355Q_ERR_UNSUPPORTED;
    }
}

struct MoeArgs {
    const int32_t * ids; int64_t ids_stride; int64_t expert_stride; int64_t x_stride2;
    int n_used; int x_ne1; int n_expert; int n_pairs;
};

static int launch_gemv_fast_typed(const mi355q_mat * mats, int n_mats, const float * x, int64_t x_stride,
                                  int ncols, int64_t k, int flags, int n_cu, hipStream_t stream, const MoeArgs * moe);

// mats: planar device rows of one activation family, same k.  ncols <= gemv_fast_max_cols().
// Matrices of different weight types are grouped per type (one launch per type).
int launch_gemv_fast(const mi355q_mat * mats, int n_mats, const float * x, int64_t x_stride,
                     int ncols, int64_t k, int flags, int n_cu, hipStream_t stream, const MoeArgs * moe = nullptr) {
    if (n_mats < 1 || n_mats > GEMV_MAX_MATS) return MI355Q_ERR_UNSUPPORTED;
    bool done[GEMV_MAX_MATS] = { false, false, false, false };
    for (int i = 0; i < n_mats; ++i) {
        if (done[i]) continue;
        mi355q_mat grp[GEMV_MAX_MATS]; int ng = 0;
        for (int j = i; j < n_mats; ++j) if (!done[j] && mats[j].type == mats[i].type) { grp[ng++] = mats[j]; done[j] = true; }
        const int rc = launch_gemv_fast_typed(grp, ng, x, x_stride, ncols, k, flags, n_cu, stream, moe);
        if (rc != MI355Q_OK) return rc;
    }
    return MI355Q_OK;
}

static int launch_gemv_fast_typed(const mi355q_mat * mats, int n_mats, const float * x, int64_t x_stride,
                                  int ncols, int64_t k, int flags, int n_cu, hipStream_t stream, const MoeArgs * moe) {
    if (n_mats < 1 || n_mats > GEMV_MAX_MATS) return MI355Q_ERR_UNSUPPORTED;
    const int fam = family_of(mats[0].type);
    if (fam < 0) return MI355Q_ERR_UNSUPPORTED;
    GemvArgs a = {};
    int64_t rows = 0;
    int max_steps = 0;
    for (int i = 0; i < n_mats; ++i) {
        if (family_of(mats[i].type) != fam) return MI355Q_ERR_UNSUPPORTED;
        if (((uintptr_t) mats[i].w | (uintptr_t) mats[i].w_stride) & 15) return MI355Q_ERR_ALIGN;
        a.mats[i].w = (const uint8_t *) mats[i].w; a.mats[i].y = mats[i].y;
        a.mats[i].w_stride = mats[i].w_stride; a.mats[i].y_stride = mats[i].y_stride;
        a.mats[i].row_begin = rows; a.mats[i].type = mats[i].type;
        rows += mats[i].m;
        const int st = (row_chunks(mats[i].type, (int) k) + 63) / 64;
        if (st > max_steps) max_steps = st;
    }
    if (rows == 0) return MI355Q_OK;
    if (rows > 0x7FFFFFF0) return MI355Q_ERR_UNSUPPORTED;
    a.x = x; a.x_stride = x_stride; a.total_rows = rows; a.n_mats = n_mats; a.k = (int) k;
    a.x_vec = (((uintptr_t) x | (uintptr_t) x_stride) & 15) == 0 ? 1 : 0;
    int pairs = 1;
    if (moe) {
        if (n_mats != 1 || ncols != 1) return MI355Q_ERR_UNSUPPORTED;
        a.ids = moe->ids; a.ids_stride = moe->ids_stride; a.expert_stride = moe->expert_stride; a.x_stride2 = moe->x_stride2;
        a.n_used = moe->n_used; a.x_ne1 = moe->x_ne1; a.n_expert = moe->n_expert;
        pairs = moe->n_pairs;
        if (pairs <= 0) return MI355Q_OK;
        if (pairs > 65535) return MI355Q_ERR_UNSUPPORTED;
    }
    // grid: one 16-wave workgroup per CU, each a contiguous row range
    int64_t grid = (int64_t) n_cu / pairs;
    if (grid < 1) grid = 1;
    int64_t rpw = (rows + grid - 1) / grid;
    if (rpw < 1) rpw = 1;
    grid = (rows + rpw - 1) / rpw;
    a.rows_per_wg = rpw;
    const int colb = (lds_col_bytes(fam, (int) k) + 15) & ~15;
    const size_t lds_bytes = (size_t) colb * ncols;
    const bool even = (flags & MI355Q_FLAG_ROUND_EVEN) != 0;
    const dim3 g((unsigned) grid, (unsigned) pairs);
    const int wt = mats[0].type;                               // the caller groups by type
    if (fam == FAM_Q8K) {                                      // Q8_K activations have one rounding rule
        switch (wt) {
        case MI355Q_TYPE_Q4_K: return launch_cols<FAM_Q8K, MI355Q_TYPE_Q4_K, false>(a, ncols, g, lds_bytes, stream);
        case MI355Q_TYPE_Q6_K: return launch_cols<FAM_Q8K, MI355Q_TYPE_Q6_K, false>(a, ncols, g, lds_bytes, stream);
        case MI355Q_TYPE_IQ4_XS: return launch_cols<FAM_Q8K, MI355Q_TYPE_IQ4_XS, false>(a, ncols, g, lds_bytes, stream);
        default:               return launch_cols<FAM_Q8K, MI355Q_TYPE_Q5_K, false>(a, ncols, g, lds_bytes, stream);
        }
    }
    if (wt == MI355Q_TYPE_IQ4_NL) return even ? launch_cols<FAM_Q80, MI355Q_TYPE_IQ4_NL, true>(a, ncols, g, lds_bytes, stream)
                                              : launch_cols<FAM_Q80, MI355Q_TYPE_IQ4_NL, false>(a, ncols, g, lds_bytes, stream);
    if (even) return wt == MI355Q_TYPE_Q8_0 ? launch_cols<FAM_Q80, MI355Q_TYPE_Q8_0, true>(a, ncols, g, lds_bytes, stream)
                                            : launch_cols<FAM_Q80, MI355Q_TYPE_Q4_0, true>(a, ncols, g, lds_bytes, stream);
    return wt == MI355Q_TYPE_Q8_0 ? launch_cols<FAM_Q80, MI355Q_TYPE_Q8_0, false>(a, ncols, g, lds_bytes, stream)
                                  : launch_cols<FAM_Q80, MI355Q_TYPE_Q4_0, false>(a, ncols, g, lds_bytes, stream);
}

#ifdef MI355Q_STAMPS
extern "C" int mi355q_debug_set_stamps(void * dev_buf) {
    unsigned long long * p = (unsigned long long *) dev_buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -4;
}
#endif

} // namespace mi355q
