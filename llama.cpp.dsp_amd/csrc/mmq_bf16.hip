// mmq_bf16.hip -- the batched (prefill) tier: y[n][m] = W_q[m][k] . x[n][k] for n > 8, on the matrix cores.
//
// Replaces the reference's two CDNA prefill routes -- mul_mat_q (dp4a int8 tiles, ggml-cuda/mmq.cuh:2595-2674,
// only used for n < 64 on CDNA, mmq.cu:152) and dequantize-to-f16 + hipBLAS GEMM (ggml-cuda/convert.cu:189-277 +
// ggml-cuda.cu:1225-1259, which writes and re-reads a full f16 copy of W per call) -- with one kernel:
// packed super-blocks are read ONCE from HBM, dequantized in registers straight into a bf16 LDS tile, and
// multiplied with v_mfma_f32_16x16x32_bf16 (f32 accumulate).  No dequantized copy of W ever touches HBM.
//
// Tiling (wave64, 256 threads = 2x2 waves, each wave a 64x64 output tile = 4x4 MFMA tiles, 64 accumulator VGPRs):
//   BM = 128 weight rows, BN = 128 tokens, BK = 128 elements per step.
//   thread t dequantizes 64 consecutive k of row t/2 (k half t%2) -> 8 x ds_write_b128;
//   the activations are converted to bf16 [n][k] once per call (k_x_to_bf16) and staged with 16-byte loads;
//   LDS rows are padded by 16 B so the ds_read_b128 fragment reads of 16 rows hit 16 distinct bank groups.
//   The packed weights of step i+1 are loaded (global_load_dwordx4) before the MFMAs of step i are issued.
//
// Numerics: w = d*sc*q - dmin*m is formed in f32 exactly as dequantize_row_* does, then rounded to bf16; the
// activations are rounded to bf16 (no Q8 quantization on this tier).  Error vs the CPU backend is dominated
// by the CPU's own Q8_K activation quantization; measured NMSE ~1e-5, bound 5e-4 (tests/test-backend-ops.cpp:1990).
// Bound: MFMA (2*M*N*K flop); HBM traffic = packed W once per 128-token column tile + bf16 activations.
#include "mi355q_common.h"

namespace mi355q {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float  f32x2;
typedef __attribute__((ext_vector_type(4))) float  f32x4;

constexpr int MMQ_BM = 128, MMQ_BK = 128, MMQ_THREADS = 256;   // BN (tokens per tile) is a template parameter: 128 or 64
constexpr int MMQ_LDS_STRIDE = (MMQ_BK + 8) * 2;          // bytes per LDS row (16-byte pad)

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    const f32x2 v = { a, b };
    const bf16x2 r = __builtin_convertvector(v, bf16x2);  // v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN-preserving)
    return __builtin_bit_cast(uint32_t, r);
}

// ---- f32 activations -> bf16 [n][k] (contiguous) -------------------------------------------------
__global__ void __launch_bounds__(256)
k_x_to_bf16(const float * __restrict__ x, int64_t x_stride, uint32_t * __restrict__ out, int64_t n, int64_t k) {
    const int64_t pairs = k / 2;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n * pairs; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t r = i / pairs, c = i - r * pairs;
        const float * xr = (const float *) ((const char *) x + r * x_stride);
        out[i] = pack_bf16(xr[2 * c], xr[2 * c + 1]);
    }
}

// ---- per-type: packed bytes of 64 consecutive k of one row -> 64 f32 weights -----------------------
// `row` = planar device row, nb = blocks per row, ks = 128-element step index, sub = which 64 of the step.
// load() only issues global loads (so the caller can run them ahead of the MFMAs); dequant() consumes them.
template <int T> struct W64;

template <> struct W64<MI355Q_TYPE_Q4_K> {          // planar [qs 128*nb][hdr 16*nb]
    uint4 q0, q1, h;
    __device__ __forceinline__ void load(const uint8_t * row, int nb, int ks, int sub) {
        const int b = ks >> 1, g = 2 * (ks & 1) + sub;                   // 64-group g of block b: qs bytes 32g..32g+31
        const uint8_t * p = row + 128 * (int64_t) b + 32 * g;
        q0 = ldg16_nt(p); q1 = ldg16_nt(p + 16);
        h  = ldg16(row + 128 * (int64_t) nb + 16 * b);
        gsel = g;
    }
    int gsel;
    __device__ __forceinline__ void dequant(float * w) const {          // dequantize_row_q4_K, ggml-quants.c:1280-1302
        const float d = h2f(h.x & 0xFFFFu), dmin = h2f(h.x >> 16);
        int sc0, mn0, sc1, mn1;
        k4_scale_min(h.y, h.z, h.w, 2 * gsel, sc0, mn0);
        k4_scale_min(h.y, h.z, h.w, 2 * gsel + 1, sc1, mn1);
        const float d1 = d * (float) sc0, m1 = dmin * (float) mn0, d2 = d * (float) sc1, m2 = dmin * (float) mn1;
        const uint32_t qw[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t byte = (qw[i] >> (8 * j)) & 0xFFu;
                w[4 * i + j]      = d1 * (float) (byte & 0x0Fu) - m1;
                w[32 + 4 * i + j] = d2 * (float) (byte >> 4) - m2;
            }
    }
};

template <> struct W64<MI355Q_TYPE_Q5_K> {          // planar [qs 128*nb][qh 32*nb][hdr 16*nb]
    uint4 q0, q1, b0, b1, h; int gsel;
    __device__ __forceinline__ void load(const uint8_t * row, int nb, int ks, int sub) {
        const int b = ks >> 1, g = 2 * (ks & 1) + sub;
        const uint8_t * p = row + 128 * (int64_t) b + 32 * g;
        q0 = ldg16_nt(p); q1 = ldg16_nt(p + 16);
        const uint8_t * qh = row + 128 * (int64_t) nb + 32 * b;
        b0 = ldg16(qh); b1 = ldg16(qh + 16);
        h  = ldg16(row + 160 * (int64_t) nb + 16 * b);
        gsel = g;
    }
    __device__ __forceinline__ void dequant(float * w) const {          // dequantize_row_q5_K, ggml-quants.c:1482-1507
        const float d = h2f(h.x & 0xFFFFu), dmin = h2f(h.x >> 16);
        int sc0, mn0, sc1, mn1;
        k4_scale_min(h.y, h.z, h.w, 2 * gsel, sc0, mn0);
        k4_scale_min(h.y, h.z, h.w, 2 * gsel + 1, sc1, mn1);
        const float d1 = d * (float) sc0, m1 = dmin * (float) mn0, d2 = d * (float) sc1, m2 = dmin * (float) mn1;
        const uint32_t qw[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
        const uint32_t hw[8] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t byte = (qw[i] >> (8 * j)) & 0xFFu, hb = (hw[i] >> (8 * j)) & 0xFFu;
                w[4 * i + j]      = d1 * (float) ((byte & 0x0Fu) + (((hb >> (2 * gsel)) & 1u) << 4)) - m1;
                w[32 + 4 * i + j] = d2 * (float) ((byte >> 4) + (((hb >> (2 * gsel + 1)) & 1u) << 4)) - m2;
            }
    }
};

template <> struct W64<MI355Q_TYPE_Q6_K> {          // planar [ql 128*nb][qh 64*nb][scales 16*nb PERMUTED][d 2*nb]
    uint4 l0, l1, l2, l3, h0, h1, sc; uint32_t dh; int hsel, ssel;
    __device__ __forceinline__ void load(const uint8_t * row, int nb, int ks, int sub) {
        const int b = ks >> 1, hh = ks & 1;                              // half hh of block b; sub 0: low nibbles, 1: high nibbles
        const uint8_t * ql = row + 128 * (int64_t) b + 64 * hh;
        l0 = ldg16_nt(ql); l1 = ldg16_nt(ql + 16); l2 = ldg16_nt(ql + 32); l3 = ldg16_nt(ql + 48);
        const uint8_t * qh = row + 128 * (int64_t) nb + 64 * b + 32 * hh;
        h0 = ldg16(qh); h1 = ldg16(qh + 16);
        sc = ldg16(row + 192 * (int64_t) nb + 16 * b);
        dh = *(const uint16_t *) (row + 208 * (int64_t) nb + 2 * b);
        hsel = hh; ssel = sub;
    }
    // scale of sub-block s (0..15) from the permuted 16-byte group (layout.hip: device byte 2j <- 8h+2cc+p, 2j+1 <- +4)
    __device__ __forceinline__ int scale(int s) const {
        const int hh = s >> 3, t = s & 7, hi = t >> 2, j = 4 * hh + (t & 3);
        const int dev = 2 * j + hi;
        const uint32_t wv = dev < 4 ? sc.x : (dev < 8 ? sc.y : (dev < 12 ? sc.z : sc.w));
        return (int) (int8_t) ((wv >> (8 * (dev & 3))) & 0xFFu);
    }
    __device__ __forceinline__ void dequant(float * w) const {          // dequantize_row_q6_K, ggml-quants.c:1690-1719
        const float d = h2f(dh);
        const uint32_t lw[16] = { l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w, l2.x, l2.y, l2.z, l2.w, l3.x, l3.y, l3.z, l3.w };
        const uint32_t hw[8]  = { h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w };
        float dsc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) dsc[i] = d * (float) scale(8 * hsel + 4 * ssel + i);
        // element e (0..63) of this 64-run: ql byte e, qh byte e & 31, 2-bit field 2*(e>>5) (+4 for the high nibbles)
#pragma unroll
        for (int e = 0; e < 64; ++e) {
            const uint32_t lb = (lw[e >> 2] >> (8 * (e & 3))) & 0xFFu;
            const uint32_t hb = (hw[(e & 31) >> 2] >> (8 * (e & 3))) & 0xFFu;
            const uint32_t lo4 = ssel ? (lb >> 4) : (lb & 0x0Fu);
            const uint32_t hi2 = (hb >> (2 * (e >> 5) + 4 * ssel)) & 3u;
            w[e] = dsc[e >> 4] * (float) ((int) (lo4 | (hi2 << 4)) - 32);
        }
    }
};

template <> struct W64<MI355Q_TYPE_Q8_0> {          // planar [qs 32*nb][d 2*nb]; 64 k = 2 blocks
    uint4 q0, q1, q2, q3; uint32_t d01;
    __device__ __forceinline__ void load(const uint8_t * row, int nb, int ks, int sub) {
        const int b = 4 * ks + 2 * sub;
        const uint8_t * p = row + 32 * (int64_t) b;
        q0 = ldg16_nt(p); q1 = ldg16_nt(p + 16); q2 = ldg16_nt(p + 32); q3 = ldg16_nt(p + 48);
        d01 = *(const uint32_t *) (row + 32 * (int64_t) nb + 2 * b);     // two f16 scales (b is even: 4-byte aligned)
    }
    __device__ __forceinline__ void dequant(float * w) const {          // dequantize_row_q8_0, ggml-quants.c:349-363
        const float d0 = h2f(d01 & 0xFFFFu), d1 = h2f(d01 >> 16);
        const uint32_t qw[16] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w };
#pragma unroll
        for (int e = 0; e < 64; ++e) {
            const int q = (int) (int8_t) ((qw[e >> 2] >> (8 * (e & 3))) & 0xFFu);
            w[e] = (float) q * (e < 32 ? d0 : d1);
        }
    }
};

template <> struct W64<MI355Q_TYPE_Q4_0> {          // planar [qs 16*nb][d 2*nb]; 64 k = 2 blocks
    uint4 q0, q1; uint32_t d01;
    __device__ __forceinline__ void load(const uint8_t * row, int nb, int ks, int sub) {
        const int b = 4 * ks + 2 * sub;
        const uint8_t * p = row + 16 * (int64_t) b;
        q0 = ldg16_nt(p); q1 = ldg16_nt(p + 16);
        d01 = *(const uint32_t *) (row + 16 * (int64_t) nb + 2 * b);
    }
    __device__ __forceinline__ void dequant(float * w) const {          // dequantize_row_q4_0, ggml-quants.c:255-273
        const float d[2] = { h2f(d01 & 0xFFFFu), h2f(d01 >> 16) };
        const uint32_t qw[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t byte = (qw[4 * blk + (j >> 2)] >> (8 * (j & 3))) & 0xFFu;
                w[32 * blk + j]      = (float) ((int) (byte & 0x0Fu) - 8) * d[blk];
                w[32 * blk + 16 + j] = (float) ((int) (byte >> 4) - 8) * d[blk];
            }
    }
};

// ---- the kernel ---------------------------------------------------------------------------------
template <int T, int MMQ_BN>
__global__ void __launch_bounds__(MMQ_THREADS, 2)
k_mmq_bf16(const uint8_t * __restrict__ w, int64_t w_stride, const uint16_t * __restrict__ xb /* bf16 [n][k] */,
           float * __restrict__ y, int64_t y_stride, int m, int n, int k) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];      // 2 * 128 * 272 B = 68 KiB (> the 64 KiB static limit)
    uint8_t * Ws = lds;
    uint8_t * Xs = lds + MMQ_BM * MMQ_LDS_STRIDE;                          // BN rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                              // 2 x 2 waves, 64 x (BN/2) each
    constexpr int WN = MMQ_BN / 2, NT = WN / 16;                          // tokens per wave, 16-token MFMA tiles per wave
    const int m0 = blockIdx.x * MMQ_BM, n0 = blockIdx.y * MMQ_BN;
    const int nb = (T == MI355Q_TYPE_Q8_0 || T == MI355Q_TYPE_Q4_0) ? k >> 5 : k >> 8;
    const int steps = k / MMQ_BK;

    // this thread's dequant job: row m0 + tid/2, k half tid%2
    const int  wr = tid >> 1, wsub = tid & 1;
    const bool wvalid = m0 + wr < m;
    const uint8_t * wrow = w + (int64_t) (wvalid ? m0 + wr : 0) * w_stride;
    // this thread's activation staging job: token n0 + tid/2, 64 k (128 B) at k half tid%2
    const int  xr = tid >> 1, xsub = tid & 1;
    const bool xvalid = xr < MMQ_BN && n0 + xr < n;
    const uint16_t * xrow = xb + (int64_t) (xvalid ? n0 + xr : 0) * k + 64 * xsub;

    f32x4 acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4) { 0.f, 0.f, 0.f, 0.f };

    W64<T> wq;
    uint4 xv[8];
    if (wvalid) wq.load(wrow, nb, 0, wsub);
#pragma unroll
    for (int i = 0; i < 8; ++i) xv[i] = xvalid ? *(const uint4 *) (xrow + 8 * i) : make_uint4(0, 0, 0, 0);

    for (int ks = 0; ks < steps; ++ks) {
        // ---- stage step ks into LDS (registers were loaded one step ahead) ----
        {
            float wf[64];
            if (wvalid) wq.dequant(wf);
            else {
#pragma unroll
                for (int e = 0; e < 64; ++e) wf[e] = 0.0f;
            }
            uint8_t * dst = Ws + wr * MMQ_LDS_STRIDE + 128 * wsub;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint4 v;
                v.x = pack_bf16(wf[8 * i], wf[8 * i + 1]); v.y = pack_bf16(wf[8 * i + 2], wf[8 * i + 3]);
                v.z = pack_bf16(wf[8 * i + 4], wf[8 * i + 5]); v.w = pack_bf16(wf[8 * i + 6], wf[8 * i + 7]);
                *(uint4 *) (dst + 16 * i) = v;
            }
            if (xr < MMQ_BN) {
                uint8_t * xdst = Xs + xr * MMQ_LDS_STRIDE + 128 * xsub;
#pragma unroll
                for (int i = 0; i < 8; ++i) *(uint4 *) (xdst + 16 * i) = xv[i];
            }
        }
        __syncthreads();
        // ---- prefetch step ks+1 while the matrix cores work on step ks ----
        if (ks + 1 < steps) {
            if (wvalid) wq.load(wrow, nb, ks + 1, wsub);
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[i] = xvalid ? *(const uint4 *) (xrow + (int64_t) (ks + 1) * MMQ_BK + 8 * i) : make_uint4(0, 0, 0, 0);
        }
        // ---- 4 k-slices of 32: A = W rows (lane: row l&15, k 8*(l>>4)..+7), B = tokens (lane: col l&15, same k) ----
#pragma unroll
        for (int kk = 0; kk < MMQ_BK / 32; ++kk) {
            bf16x8 af[4], bfr[NT];
            const int koff = 2 * (32 * kk + 8 * (lane >> 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i]  = *(const bf16x8 *) (Ws + (64 * wm + 16 * i + (lane & 15)) * MMQ_LDS_STRIDE + koff);
#pragma unroll
            for (int j = 0; j < NT; ++j) bfr[j] = *(const bf16x8 *) (Xs + (WN * wn + 16 * j + (lane & 15)) * MMQ_LDS_STRIDE + koff);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane&15 (token), row = 4*(lane>>4) + reg (weight row): 4 consecutive m per lane ----
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int tok = n0 + WN * wn + 16 * j + (lane & 15);
        if (tok >= n) continue;
        float * yr = (float *) ((char *) y + (int64_t) tok * y_stride);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mr = m0 + 64 * wm + 16 * i + 4 * (lane >> 4);
            if (mr + 3 < m) *(f32x4 *) (yr + mr) = acc[i][j];
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (mr + r < m) yr[mr + r] = acc[i][j][r];
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------
bool mmq_supported(int type, int64_t k) {
    switch (type) {
    case MI355Q_TYPE_Q4_K: case MI355Q_TYPE_Q5_K: case MI355Q_TYPE_Q6_K: return k % 256 == 0;
    case MI355Q_TYPE_Q8_0: case MI355Q_TYPE_Q4_0: return k % 128 == 0;
    default: return false;
    }
}

size_t mmq_workspace(int64_t n, int64_t k) { return (size_t) (n * k * 2 + 255) & ~(size_t) 255; }

// w: planar device rows; x f32 [n][k] (row stride x_stride); workspace >= mmq_workspace(n,k); y f32 [n][m] (y_stride % 16 == 0)
int launch_mmq_bf16(int type, const void * w, int64_t w_stride, const float * x, int64_t x_stride,
                    float * y, int64_t y_stride, int64_t m, int64_t n, int64_t k, void * workspace, int n_cu, hipStream_t stream) {
    if (!mmq_supported(type, k)) return MI355Q_ERR_UNSUPPORTED;
    if (m <= 0 || n <= 0) return MI355Q_OK;
    if ((y_stride & 15) || ((uintptr_t) y & 15)) return MI355Q_ERR_ALIGN;
    const int64_t pairs = n * k / 2;
    const int cgrid = (int) ((pairs + 255) / 256 < 8192 ? (pairs + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_x_to_bf16, dim3(cgrid), dim3(256), 0, stream, x, x_stride, (uint32_t *) workspace, n, k);
    // 128-token tiles amortize the dequantization best; when that leaves CUs idle (M = 4096, N = 512 is only 128 tiles
    // for 256 CUs) use 64-token tiles.
    const int64_t tiles128 = ((m + MMQ_BM - 1) / MMQ_BM) * ((n + 127) / 128);
    const int bn = (tiles128 < (int64_t) n_cu && n > 64) ? 64 : 128;
    const dim3 grid((unsigned) ((m + MMQ_BM - 1) / MMQ_BM), (unsigned) ((n + bn - 1) / bn));
#define MI355Q_MMQ_LAUNCH(T, BN) {                                                                                                 \
        constexpr size_t lds_bytes = (size_t) (MMQ_BM + BN) * MMQ_LDS_STRIDE;                                                      \
        static bool attr_set = false;                                                                                              \
        if (!attr_set) {                                                                                                           \
            if (hipFuncSetAttribute((const void *) k_mmq_bf16<T, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes) != hipSuccess) \
                return MI355Q_ERR_HIP;                                                                                             \
            attr_set = true;                                                                                                       \
        }                                                                                                                          \
        hipLaunchKernelGGL((k_mmq_bf16<T, BN>), grid, dim3(MMQ_THREADS), lds_bytes, stream, (const uint8_t *) w, w_stride,         \
                           (const uint16_t *) workspace, y, y_stride, (int) m, (int) n, (int) k); }
#define MI355Q_MMQ_CASE(T) case T: if (bn == 64) MI355Q_MMQ_LAUNCH(T, 64) else MI355Q_MMQ_LAUNCH(T, 128) break;
    switch (type) {
        MI355Q_MMQ_CASE(MI355Q_TYPE_Q4_K) MI355Q_MMQ_CASE(MI355Q_TYPE_Q5_K) MI355Q_MMQ_CASE(MI355Q_TYPE_Q6_K)
        MI355Q_MMQ_CASE(MI355Q_TYPE_Q8_0) MI355Q_MMQ_CASE(MI355Q_TYPE_Q4_0)
    default: return MI355Q_ERR_UNSUPPORTED;
    }
#undef MI355Q_MMQ_CASE
#undef MI355Q_MMQ_LAUNCH
    return hipGetLastError() == hipSuccess ? MI355Q_OK : MI355Q_ERR_HIP;
}

} // namespace mi355q
