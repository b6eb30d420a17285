// gemv_fast.hip -- the token-generation hot loop: y[n][m] = W_q[m][k] . x[n][k], n <= 8, for
// PLANAR device rows (see mi355q_common.h).  Replaces the reference's quantize_q8_1 + mul_mat_vec_q
// pair (ggml-cuda/quantize.cu:4-38, mmvq.cu:130-288) with ONE launch, designed for gfx950:
//
//  * fused prologue: every workgroup quantizes the activation columns itself (Q8_K / Q8_0, exactly
//    the CPU's arithmetic, act_quant.cuh) into LDS -- no separate quantize launch, no HBM round trip;
//    the f32 activations (<= 8 x 56 KiB) come out of L2.
//  * one wave = one weight row at a time; lane l owns the 16-byte payload chunks l, l+64, ... of the
//    row: a wave-wide global_load_dwordx4 reads 1 KiB of contiguous packed quants (non-temporal, the
//    weights are streamed exactly once), the small per-block fields come from their own dense planes.
//  * all chunk loads of a step are issued before the first use (U chunks per lane in flight);
//    nibble/6-bit unpack with full-dword bit ops, v_dot4_i32_i8 against int8 activations from LDS
//    (ds_read_b128), exact int32 block sums, f32 scale, 64-lane shuffle reduction.
//  * several matrices that share the activations (wq/wk/wv, gate/up) run in ONE launch (row ranges
//    are concatenated; the type switch is wave-uniform), removing launch gaps from the token loop.
//
// Bound: HBM read of W.  Algorithmic bytes per launch = sum_i m_i * row_size(type_i, k).
#include "act_quant.cuh"

namespace mi355q {

constexpr int GEMV_THREADS = 512;            // 8 waves
constexpr int GEMV_WAVES   = GEMV_THREADS / WAVE;
constexpr int GEMV_MAX_MATS = 4;

struct GemvMat {
    const uint8_t * w;
    float *         y;
    int64_t         w_stride;
    int64_t         y_stride;   // bytes between activation columns in y
    int64_t         row_begin;  // first concatenated row index of this matrix
    int             type;
    int             pad;
};

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
// LDS image of the quantized activations, per column n (all offsets in bytes from the column base)
//   Q8_K family: q8[k] | d f32 [k/256] | bsums i16 [k/16]
//   Q8_0 family: q8[k] | d f32 [k/32] (already f16-rounded) | sums i32 [k/32]
// ------------------------------------------------------------------------------------------------
enum { FAM_Q8K = 0, FAM_Q80 = 1 };

__host__ __device__ __forceinline__ int lds_col_bytes(int fam, int k) {
    return fam == FAM_Q8K ? k + (k / 256) * 4 + (k / 16) * 2 : k + (k / 32) * 8;
}

struct ActView {
    const uint8_t * base;   // column base in LDS
    int             k;
    __device__ __forceinline__ uint4 q16(int e) const { return *(const uint4 *) (base + e); }       // 16 int8, e % 16 == 0
    // Q8_K
    __device__ __forceinline__ float dK(int b) const { return *(const float *) (base + k + 4 * b); }
    __device__ __forceinline__ int   bsum(int g16) const { return *(const int16_t *) (base + k + (k >> 6) + 2 * g16); }
    // Q8_0
    __device__ __forceinline__ float d0(int b) const { return *(const float *) (base + k + 4 * b); }
    __device__ __forceinline__ int   sum0(int b) const { return *(const int *) (base + k + (k >> 3) + 4 * b); }
};

__device__ __forceinline__ int dot16(const uint32_t w[4], const uint4 a) {
    int s = dot4((int) w[0], (int) a.x, 0);
    s = dot4((int) w[1], (int) a.y, s);
    s = dot4((int) w[2], (int) a.z, s);
    s = dot4((int) w[3], (int) a.w, s);
    return s;
}

// ------------------------------------------------------------------------------------------------
// per-type chunk processing.  Each struct: load(row, nb, c) issues the global loads of chunk c,
// then accumulate<NCOLS>(...) consumes them.  c = index of the 16-byte payload chunk in the row.
// ------------------------------------------------------------------------------------------------
struct ChunkQ4K {          // block_q4_K planar: [qs 128*nb][hdr(d,dmin,scales) 16*nb]      ggml-common.h:285-296
    uint4 q, h;
    static constexpr int CHUNKS_PER_BLOCK = 8;
    __device__ __forceinline__ void load(const uint8_t * row, int nb, int c) {
        q = ldg16_nt(row + 16 * (int64_t) c);
        h = ldg16(row + 128 * (int64_t) nb + 16 * (c >> 3));
    }
    template <int NCOLS>
    __device__ __forceinline__ void accumulate(int c, const ActView * av, float * acc) const {
        const int b = c >> 3, g = (c >> 1) & 3, half = c & 1;
        uint32_t lo[4] = { q.x & 0x0F0F0F0Fu, q.y & 0x0F0F0F0Fu, q.z & 0x0F0F0F0Fu, q.w & 0x0F0F0F0Fu };
        uint32_t hi[4] = { (q.x >> 4) & 0x0F0F0F0Fu, (q.y >> 4) & 0x0F0F0F0Fu, (q.z >> 4) & 0x0F0F0F0Fu, (q.w >> 4) & 0x0F0F0F0Fu };
        const float d = h2f(h.x & 0xFFFFu), dmin = h2f(h.x >> 16);
        int sc0, mn0, sc1, mn1;
        k4_scale_min(h.y, h.z, h.w, 2 * g, sc0, mn0);
        k4_scale_min(h.y, h.z, h.w, 2 * g + 1, sc1, mn1);
        const int e = 256 * b + 64 * g + 16 * half;          // low nibbles -> e.., high nibbles -> e+32..
#pragma unroll
        for (int n = 0; n < NCOLS; ++n) {
            const int s0 = dot16(lo, av[n].q16(e)), s1 = dot16(hi, av[n].q16(e + 32));
            const int m  = mn0 * av[n].bsum(16 * b + 4 * g + half) + mn1 * av[n].bsum(16 * b + 4 * g + 2 + half);
            const float yd = av[n].dK(b);
            acc[n] += (d * yd) * (float) (sc0 * s0 + sc1 * s1) - (dmin * yd) * (float) m;
        }
    }
};

struct ChunkQ5K {          // block_q5_K planar: [qs 128*nb][qh 32*nb][hdr 16*nb]               ggml-common.h:302-314
    uint4 q, hb, h;
    static constexpr int CHUNKS_PER_BLOCK = 8;
    __device__ __forceinline__ void load(const uint8_t * row, int nb, int c) {
        q  = ldg16_nt(row + 16 * (int64_t) c);
        hb = ldg16(row + 128 * (int64_t) nb + 32 * (c >> 3) + 16 * (c & 1));
        h  = ldg16(row + 160 * (int64_t) nb + 16 * (c >> 3));
    }
    template <int NCOLS>
    __device__ __forceinline__ void accumulate(int c, const ActView * av, float * acc) const {
        const int b = c >> 3, g = (c >> 1) & 3, half = c & 1;
        const uint32_t qw[4] = { q.x, q.y, q.z, q.w }, hw[4] = { hb.x, hb.y, hb.z, hb.w };
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[i] = (qw[i] & 0x0F0F0F0Fu)        | (((hw[i] >> (2 * g))     & 0x01010101u) << 4);
            hi[i] = ((qw[i] >> 4) & 0x0F0F0F0Fu) | (((hw[i] >> (2 * g + 1)) & 0x01010101u) << 4);
        }
        const float d = h2f(h.x & 0xFFFFu), dmin = h2f(h.x >> 16);
        int sc0, mn0, sc1, mn1;
        k4_scale_min(h.y, h.z, h.w, 2 * g, sc0, mn0);
        k4_scale_min(h.y, h.z, h.w, 2 * g + 1, sc1, mn1);
        const int e = 256 * b + 64 * g + 16 * half;
#pragma unroll
        for (int n = 0; n < NCOLS; ++n) {
            const int s0 = dot16(lo, av[n].q16(e)), s1 = dot16(hi, av[n].q16(e + 32));
            const int m  = mn0 * av[n].bsum(16 * b + 4 * g + half) + mn1 * av[n].bsum(16 * b + 4 * g + 2 + half);
            const float yd = av[n].dK(b);
            acc[n] += (d * yd) * (float) (sc0 * s0 + sc1 * s1) - (dmin * yd) * (float) m;
        }
    }
};

struct ChunkQ6K {          // block_q6_K planar: [ql 128*nb][qh 64*nb][scales 16*nb][d 2*nb]    ggml-common.h:320-326
    uint4 ql, qh; int sc0, sc1; uint32_t dh;
    static constexpr int CHUNKS_PER_BLOCK = 8;
    __device__ __forceinline__ void load(const uint8_t * row, int nb, int c) {
        const int b = c >> 3, j = c & 7;                     // j = 4h + 2cc + p : ql bytes 16j..16j+15 of the block
        ql = ldg16_nt(row + 16 * (int64_t) c);
        qh = ldg16_nt(row + 128 * (int64_t) nb + 64 * b + 32 * (j >> 2) + 16 * (j & 1));
        const int8_t * sp = (const int8_t *) row + 192 * (int64_t) nb + 16 * b + 8 * (j >> 2) + 2 * ((j >> 1) & 1) + (j & 1);
        sc0 = sp[0]; sc1 = sp[4];
        dh = *(const uint16_t *) (row + 208 * (int64_t) nb + 2 * b);
    }
    template <int NCOLS>
    __device__ __forceinline__ void accumulate(int c, const ActView * av, float * acc) const {
        const int b = c >> 3, j = c & 7, cc = (j >> 1) & 1;
        const uint32_t lw[4] = { ql.x, ql.y, ql.z, ql.w }, hw[4] = { qh.x, qh.y, qh.z, qh.w };
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[i] = (lw[i] & 0x0F0F0F0Fu)        | (((hw[i] >> (2 * cc))     & 0x03030303u) << 4);
            hi[i] = ((lw[i] >> 4) & 0x0F0F0F0Fu) | (((hw[i] >> (2 * cc + 4)) & 0x03030303u) << 4);
        }
        const float d = h2f(dh);
        const int e = 256 * b + 128 * (j >> 2) + 32 * cc + 16 * (j & 1);   // low -> e.., high -> e+64..
#pragma unroll
        for (int n = 0; n < NCOLS; ++n) {
            // sum (q-32)*y = sum q*y - 32*sum y ; sum y over the 16 elements is exactly a Q8_K bsum
            const int s0 = dot16(lo, av[n].q16(e))      - 32 * av[n].bsum(e >> 4);
            const int s1 = dot16(hi, av[n].q16(e + 64)) - 32 * av[n].bsum((e >> 4) + 4);
            acc[n] += (d * av[n].dK(b)) * (float) (sc0 * s0 + sc1 * s1);
        }
    }
};

struct ChunkQ80 {          // block_q8_0 planar: [qs 32*nb][d 2*nb]; a chunk is HALF a block       ggml-common.h:209-214
    uint4 q; uint32_t dh;
    __device__ __forceinline__ void load(const uint8_t * row, int nb, int c) {
        q  = ldg16_nt(row + 16 * (int64_t) c);
        dh = *(const uint16_t *) (row + 32 * (int64_t) nb + 2 * (c >> 1));
    }
    template <int NCOLS>
    __device__ __forceinline__ void accumulate(int c, const ActView * av, float * acc) const {
        const uint32_t w[4] = { q.x, q.y, q.z, q.w };
        const float dw = h2f(dh);
#pragma unroll
        for (int n = 0; n < NCOLS; ++n) {
            int s = dot16(w, av[n].q16(16 * c));
            s += __shfl_xor(s, 1, 64);                       // exact int32 sum of the whole 32-block, as the CPU forms it
            if ((c & 1) == 0) acc[n] += (float) s * (dw * av[n].d0(c >> 1));
        }
    }
};

struct ChunkQ40 {          // block_q4_0 planar: [qs 16*nb][d 2*nb]; a chunk is one block          ggml-common.h:167-172
    uint4 q; uint32_t dh;
    __device__ __forceinline__ void load(const uint8_t * row, int nb, int c) {
        q  = ldg16_nt(row + 16 * (int64_t) c);
        dh = *(const uint16_t *) (row + 16 * (int64_t) nb + 2 * c);
    }
    template <int NCOLS>
    __device__ __forceinline__ void accumulate(int c, const ActView * av, float * acc) const {
        uint32_t lo[4] = { q.x & 0x0F0F0F0Fu, q.y & 0x0F0F0F0Fu, q.z & 0x0F0F0F0Fu, q.w & 0x0F0F0F0Fu };
        uint32_t hi[4] = { (q.x >> 4) & 0x0F0F0F0Fu, (q.y >> 4) & 0x0F0F0F0Fu, (q.z >> 4) & 0x0F0F0F0Fu, (q.w >> 4) & 0x0F0F0F0Fu };
        const float dw = h2f(dh);
#pragma unroll
        for (int n = 0; n < NCOLS; ++n) {
            // sum (q-8)*y = sum q*y - 8*sum y
            const int s = dot16(lo, av[n].q16(32 * c)) + dot16(hi, av[n].q16(32 * c + 16)) - 8 * av[n].sum0(c);
            acc[n] += (float) s * dw * av[n].d0(c);          // CPU order: sumi*d_x*d_y (ggml-cpu-quants.c:2604)
        }
    }
};

// one weight row against NCOLS activation columns: U chunks per lane in flight per step
template <typename CH, int NCOLS, int U>
__device__ __forceinline__ void row_dot(const uint8_t * row, int nb, int nchunks, const ActView * av, float * acc) {
    const int lane = lane_id();
    for (int c0 = 0; c0 < nchunks; c0 += 64 * U) {
        CH ch[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + 64 * u + lane;
            if (c < nchunks) ch[u].load(row, nb, c);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + 64 * u + lane;
            if (c < nchunks) ch[u].template accumulate<NCOLS>(c, av, acc);
        }
    }
}

template <int FAM, int NCOLS, int U, bool ROUND_EVEN>
__global__ void __launch_bounds__(GEMV_THREADS)
k_gemv_fast(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));   // wave-uniform -> SGPR
    const int k    = a.k;
    const int colb = (lds_col_bytes(FAM, k) + 15) & ~15;

    const char * xbase = (const char *) a.x;
    int64_t w_off = 0, y_off = 0;
    if (a.ids) {                                              // MoE: resolve this pair's expert on the device
        const int pair = (int) blockIdx.y, t = pair / a.n_used, u = pair - t * a.n_used;
        const int e = *(const int32_t *) ((const char *) a.ids + (int64_t) t * a.ids_stride + 4 * u);
        if (e < 0 || e >= a.n_expert) return;                 // (the reference asserts; we leave the row untouched)
        xbase += (int64_t) t * a.x_stride2 + (int64_t) (u % a.x_ne1) * a.x_stride;
        w_off = (int64_t) e * a.expert_stride;
        y_off = (int64_t) pair * a.total_rows * 4;
    }

    // ---------------- prologue: quantize the NCOLS activation columns into LDS ----------------
    const int spans = (k + 255) / 256;
    for (int idx = wave; idx < NCOLS * spans; idx += GEMV_WAVES) {
        const int n = idx / spans, span = idx - n * spans;
        const float * xr = (const float *) (xbase + (int64_t) n * a.x_stride);
        const int     e0 = span * 256 + 4 * lane;
        const float4  v  = load_x4(xr, e0, k, a.x_vec != 0);
        uint8_t * col = lds + n * colb;
        if constexpr (FAM == FAM_Q8K) {
            uint32_t q; float d; int bsum;
            q8k_wave(v, q, d, bsum);
            *(uint32_t *) (col + e0) = q;
            if (lane == 0) *(float *) (col + k + 4 * span) = d;
            if ((lane & 3) == 0) *(int16_t *) (col + k + (k >> 6) + 2 * (16 * span + (lane >> 2))) = (int16_t) bsum;
        } else {
            uint32_t q; float d; int sum;
            q80_group8<ROUND_EVEN>(v, q, d, sum);
            if (e0 < k) {
                *(uint32_t *) (col + e0) = q;
                if ((lane & 7) == 0) {
                    const int b = e0 >> 5;
                    *(float *) (col + k + 4 * b) = __half2float(__float2half_rn(d));
                    *(int *) (col + k + (k >> 3) + 4 * b) = sum;
                }
            }
        }
    }
    __syncthreads();

    ActView av[NCOLS];
#pragma unroll
    for (int n = 0; n < NCOLS; ++n) { av[n].base = lds + n * colb; av[n].k = k; }

    // ---------------- main loop: this workgroup's contiguous row range, waves interleaved ----------------
    const int64_t r_begin = (int64_t) blockIdx.x * a.rows_per_wg;
    int64_t       r_end   = r_begin + a.rows_per_wg;
    if (r_end > a.total_rows) r_end = a.total_rows;
    for (int64_t gr = r_begin + wave; gr < r_end; gr += GEMV_WAVES) {
        int mi = 0;
#pragma unroll
        for (int i = 1; i < GEMV_MAX_MATS; ++i) if (i < a.n_mats && gr >= a.mats[i].row_begin) mi = i;
        const GemvMat & mt  = a.mats[mi];
        const int64_t   r   = gr - mt.row_begin;
        const uint8_t * row = mt.w + w_off + r * mt.w_stride;
        float acc[NCOLS];
#pragma unroll
        for (int n = 0; n < NCOLS; ++n) acc[n] = 0.0f;
        if constexpr (FAM == FAM_Q8K) {
            const int nb = k >> 8;
            switch (mt.type) {                                   // wave-uniform
            case MI355Q_TYPE_Q4_K: row_dot<ChunkQ4K, NCOLS, U>(row, nb, 8 * nb, av, acc); break;
            case MI355Q_TYPE_Q5_K: row_dot<ChunkQ5K, NCOLS, U>(row, nb, 8 * nb, av, acc); break;
            default:               row_dot<ChunkQ6K, NCOLS, U>(row, nb, 8 * nb, av, acc); break;
            }
        } else {
            const int nb = k >> 5;
            if (mt.type == MI355Q_TYPE_Q8_0) row_dot<ChunkQ80, NCOLS, U>(row, nb, 2 * nb, av, acc);
            else                             row_dot<ChunkQ40, NCOLS, U>(row, nb, nb, av, acc);
        }
#pragma unroll
        for (int n = 0; n < NCOLS; ++n) {
            const float s = wave_sum(acc[n]);
            if (lane == 0) *(float *) ((char *) mt.y + y_off + (int64_t) n * mt.y_stride + 4 * r) = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int family_of(int type) {
    switch (type) {
    case MI355Q_TYPE_Q4_K: case MI355Q_TYPE_Q5_K: case MI355Q_TYPE_Q6_K: return FAM_Q8K;
    case MI355Q_TYPE_Q8_0: case MI355Q_TYPE_Q4_0: return FAM_Q80;
    default: return -1;
    }
}

int gemv_fast_family(int type) { return family_of(type); }

// largest number of activation columns whose LDS image fits (<= 8)
int gemv_fast_max_cols(int type, int64_t k) {
    const int fam = family_of(type);
    if (fam < 0) return 0;
    const int colb = (lds_col_bytes(fam, (int) k) + 15) & ~15;
    int n = (160 * 1024 - 1024) / colb;
    return n > 8 ? 8 : n;
}

template <int FAM, int NCOLS, int U, bool EVEN>
static int launch_one(const GemvArgs & a, dim3 grid, size_t lds_bytes, hipStream_t stream) {
    auto kern = k_gemv_fast<FAM, NCOLS, U, EVEN>;
    static size_t lds_enabled = 48 * 1024;                 // per kernel instantiation
    if (lds_bytes > lds_enabled) {
        if (hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return MI355Q_ERR_HIP;
        lds_enabled = 160 * 1024;
    }
    hipLaunchKernelGGL(kern, grid, dim3(GEMV_THREADS), lds_bytes, stream, a);
    return hipGetLastError() == hipSuccess ? MI355Q_OK : MI355Q_ERR_HIP;
}

// Instantiated (NCOLS, U) pairs: U = chunk loads in flight per lane.
//   NCOLS 1    : U in {1,2,4,8}      NCOLS 2..4 : U in {1,2,4}      NCOLS 5..8 : U in {1,2}
template <int FAM, bool EVEN>
static int launch_nu(const GemvArgs & a, int ncols, int u, dim3 grid, size_t lds_bytes, hipStream_t stream) {
#define MI355Q_GEMV_CASE(N, UU) if (ncols == N && u == UU) return launch_one<FAM, N, UU, EVEN>(a, grid, lds_bytes, stream);
    MI355Q_GEMV_CASE(1, 1) MI355Q_GEMV_CASE(1, 2) MI355Q_GEMV_CASE(1, 4) MI355Q_GEMV_CASE(1, 8)
    MI355Q_GEMV_CASE(2, 1) MI355Q_GEMV_CASE(2, 2) MI355Q_GEMV_CASE(2, 4)
    MI355Q_GEMV_CASE(3, 1) MI355Q_GEMV_CASE(3, 2) MI355Q_GEMV_CASE(3, 4)
    MI355Q_GEMV_CASE(4, 1) MI355Q_GEMV_CASE(4, 2) MI355Q_GEMV_CASE(4, 4)
    MI355Q_GEMV_CASE(5, 1) MI355Q_GEMV_CASE(5, 2)
    MI355Q_GEMV_CASE(6, 1) MI355Q_GEMV_CASE(6, 2)
    MI355Q_GEMV_CASE(7, 1) MI355Q_GEMV_CASE(7, 2)
    MI355Q_GEMV_CASE(8, 1) MI355Q_GEMV_CASE(8, 2)
#undef MI355Q_GEMV_CASE
    return MI355Q_ERR_UNSUPPORTED;
}

// mats: planar device rows, all of one activation family, same k.  ncols <= gemv_fast_max_cols().
struct MoeArgs {
    const int32_t * ids; int64_t ids_stride; int64_t expert_stride; int64_t x_stride2;
    int n_used; int x_ne1; int n_expert; int n_pairs;
};

int launch_gemv_fast(const mi355q_mat * mats, int n_mats, const float * x, int64_t x_stride,
                     int ncols, int64_t k, int flags, int n_cu, hipStream_t stream, const MoeArgs * moe = nullptr) {
    if (n_mats < 1 || n_mats > GEMV_MAX_MATS) return MI355Q_ERR_UNSUPPORTED;
    const int fam = family_of(mats[0].type);
    if (fam < 0) return MI355Q_ERR_UNSUPPORTED;
    GemvArgs a = {};
    int64_t rows = 0;
    int max_chunks = 0;
    for (int i = 0; i < n_mats; ++i) {
        if (family_of(mats[i].type) != fam) return MI355Q_ERR_UNSUPPORTED;
        if (((uintptr_t) mats[i].w | (uintptr_t) mats[i].w_stride) & 15) return MI355Q_ERR_ALIGN;
        a.mats[i].w = (const uint8_t *) mats[i].w; a.mats[i].y = mats[i].y;
        a.mats[i].w_stride = mats[i].w_stride; a.mats[i].y_stride = mats[i].y_stride;
        a.mats[i].row_begin = rows; a.mats[i].type = mats[i].type;
        rows += mats[i].m;
        int chunks;
        switch (mats[i].type) {
        case MI355Q_TYPE_Q8_0: chunks = (int) (k / 16); break;
        case MI355Q_TYPE_Q4_0: chunks = (int) (k / 32); break;
        default:               chunks = (int) (k / 32); break;      // 8 per 256-block
        }
        if (chunks > max_chunks) max_chunks = chunks;
    }
    if (rows == 0) return MI355Q_OK;
    a.x = x; a.x_stride = x_stride; a.total_rows = rows; a.n_mats = n_mats; a.k = (int) k;
    a.x_vec = (((uintptr_t) x | (uintptr_t) x_stride) & 15) == 0 ? 1 : 0;
    // chunks per lane per step: cover the row in as few steps as possible, at most 8 loads in flight per lane
    const int per_lane = (max_chunks + 63) / 64;
    int u = per_lane >= 7 ? 8 : (per_lane >= 3 ? 4 : (per_lane == 2 ? 2 : 1));
    // VGPR budget (-Rpass-analysis): the Q8_K family spills at U=8, and at U=4 beyond 2 columns
    if (fam == FAM_Q8K && u > 4) u = 4;
    if (ncols >= 2 && u > 4) u = 4;
    if (fam == FAM_Q8K && ncols >= 3 && u > 2) u = 2;
    if (ncols >= 5 && u > 2) u = 2;
    if (fam == FAM_Q8K && ncols >= 8) u = 1;
    int pairs = 1;
    if (moe) {
        if (n_mats != 1 || ncols != 1) return MI355Q_ERR_UNSUPPORTED;
        a.ids = moe->ids; a.ids_stride = moe->ids_stride; a.expert_stride = moe->expert_stride; a.x_stride2 = moe->x_stride2;
        a.n_used = moe->n_used; a.x_ne1 = moe->x_ne1; a.n_expert = moe->n_expert;
        pairs = moe->n_pairs;
        if (pairs <= 0) return MI355Q_OK;
        if (pairs > 65535) return MI355Q_ERR_UNSUPPORTED;
    }
    // grid: two workgroups per CU (16 waves/CU), each a contiguous row range
    int64_t grid = 2 * (int64_t) n_cu / pairs;
    if (grid < 1) grid = 1;
    int64_t rpw = (rows + grid - 1) / grid;
    if (rpw < GEMV_WAVES) rpw = GEMV_WAVES;                            // at least one row per wave
    rpw = (rpw + GEMV_WAVES - 1) / GEMV_WAVES * GEMV_WAVES;
    grid = (rows + rpw - 1) / rpw;
    a.rows_per_wg = rpw;
    const int colb = (lds_col_bytes(fam, (int) k) + 15) & ~15;
    const size_t lds_bytes = (size_t) colb * ncols;
    const bool even = (flags & MI355Q_FLAG_ROUND_EVEN) != 0;
    const dim3 g((unsigned) grid, (unsigned) pairs);
    if (fam == FAM_Q8K) return launch_nu<FAM_Q8K, false>(a, ncols, u, g, lds_bytes, stream);   // Q8_K has one rounding rule
    return even ? launch_nu<FAM_Q80, true>(a, ncols, u, g, lds_bytes, stream)
                : launch_nu<FAM_Q80, false>(a, ncols, u, g, lds_bytes, stream);
}

} // namespace mi355q
