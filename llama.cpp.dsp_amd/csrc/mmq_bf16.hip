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

constexpr int MMQ_BK = 128, MMQ_THREADS = 256;            // BM x BN (rows x tokens per tile) are template parameters: 128x128, 128x64, 64x64
constexpr int MMQ_LDS_STRIDE = (MMQ_BK + 8) * 2;          // bytes per LDS row: 256 + 16 pad -> the 16 rows of a fragment read hit 16 distinct bank groups

typedef __attribute__((ext_vector_type(2))) float f32x2p;

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

// 4 unsigned bytes of a dword -> a*byte + c as 4 floats -> 2 packed bf16 dwords.  v_cvt_f32_ubyteN (one op per weight),
// v_pk_fma_f32 (two weights per op), v_cvt_pk_bf16_f32 (two per op): 2 ops per weight, the rest is per-dword bit twiddling.
__device__ __forceinline__ void bytes4_to_bf16(uint32_t v, float a, float c, uint32_t & lo, uint32_t & hi) {
    const f32x2p q01 = { (float) (v & 0xFFu), (float) ((v >> 8) & 0xFFu) };          // the compiler selects v_cvt_f32_ubyte0..3 for these
    const f32x2p q23 = { (float) ((v >> 16) & 0xFFu), (float) (v >> 24) };
    const f32x2p aa = { a, a }, cc = { c, c };
    const f32x2p w01 = __builtin_elementwise_fma(q01, aa, cc), w23 = __builtin_elementwise_fma(q23, aa, cc);
    lo = pack_bf16(w01.x, w01.y); hi = pack_bf16(w23.x, w23.y);
}

// ---- per type: the packed bytes of 32 consecutive k of one row -> 32 bf16 (16 dwords) ----------------
// `row` = planar device row, nb = blocks per row, ks = 64-element step index, sub = which 32 of the step.
// load() only issues global loads (so the caller can run them ahead of the MFMAs); dequant() consumes them.
// w = d*sc*q - dmin*m etc. exactly as dequantize_row_* (ggml-quants.c), evaluated as one f32 FMA per weight.
template <int T> struct W32;
#define W32_DEQUANT16 \
    __device__ __forceinline__ void dequant16(uint32_t & o0, uint32_t & o1, uint32_t & o2, uint32_t & o3, uint32_t & o4, uint32_t & o5, uint32_t & o6, uint32_t & o7, \
                                              uint32_t & o8, uint32_t & o9, uint32_t & o10, uint32_t & o11, uint32_t & o12, uint32_t & o13, uint32_t & o14, uint32_t & o15) const { \
        uint32_t t[16]; dequant(t); \
        o0 = t[0]; o1 = t[1]; o2 = t[2]; o3 = t[3]; o4 = t[4]; o5 = t[5]; o6 = t[6]; o7 = t[7]; \
        o8 = t[8]; o9 = t[9]; o10 = t[10]; o11 = t[11]; o12 = t[12]; o13 = t[13]; o14 = t[14]; o15 = t[15]; }

template <> struct W32<MI355Q_TYPE_Q4_K> {          // planar [qs 128*nb][hdr 16*nb]; step = 64-group g of block b: low nibbles = first 32, high = last 32
    W32_DEQUANT16
    uint4 q0, q1, h; int j, sh;
    __device__ __forceinline__ void load(const uint8_t * row, int nb, int ks, int sub) {
        const int b = ks >> 2, g = ks & 3;
        const uint8_t * p = row + 128 * (int64_t) b + 32 * g;
        q0 = ldg16_nt(p); q1 = ldg16_nt(p + 16);
        h  = ldg16(row + 128 * (int64_t) nb + 16 * b);
        j = 2 * g + sub; sh = 4 * sub;
    }
    __device__ __forceinline__ void dequant(uint32_t * out) const {     // dequantize_row_q4_K, ggml-quants.c:1280-1302
        int sc, mn; k4_scale_min(h.y, h.z, h.w, j, sc, mn);
        const float a = h2f(h.x & 0xFFFFu) * (float) sc, c = -(h2f(h.x >> 16) * (float) mn);
        const uint32_t qw[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
#pragma unroll
        for (int i = 0; i < 8; ++i) bytes4_to_bf16((qw[i] >> sh) & 0x0F0F0F0Fu, a, c, out[2 * i], out[2 * i + 1]);
    }
};

template <> struct W32<MI355Q_TYPE_Q5_K> {          // planar [qs 128*nb][qh 32*nb][hdr 16*nb]
    W32_DEQUANT16
    uint4 q0, q1, b0, b1, h; int j, sh;
    __device__ __forceinline__ void load(const uint8_t * row, int nb, int ks, int sub) {
        const int b = ks >> 2, g = ks & 3;
        const uint8_t * p = row + 128 * (int64_t) b + 32 * g;
        q0 = ldg16_nt(p); q1 = ldg16_nt(p + 16);
        const uint8_t * qh = row + 128 * (int64_t) nb + 32 * b;
        b0 = ldg16(qh); b1 = ldg16(qh + 16);
        h  = ldg16(row + 160 * (int64_t) nb + 16 * b);
        j = 2 * g + sub; sh = 4 * sub;
    }
    __device__ __forceinline__ void dequant(uint32_t * out) const {     // dequantize_row_q5_K, ggml-quants.c:1482-1507
        int sc, mn; k4_scale_min(h.y, h.z, h.w, j, sc, mn);
        const float a = h2f(h.x & 0xFFFFu) * (float) sc, c = -(h2f(h.x >> 16) * (float) mn);
        const uint32_t qw[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
        const uint32_t hw[8] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
#pragma unroll
        for (int i = 0; i < 8; ++i)
            bytes4_to_bf16(((qw[i] >> sh) & 0x0F0F0F0Fu) | (((hw[i] >> j) & 0x01010101u) << 4), a, c, out[2 * i], out[2 * i + 1]);
    }
};

template <> struct W32<MI355Q_TYPE_Q6_K> {          // planar [ql 128*nb][qh 64*nb][scales 16*nb PERMUTED][d 2*nb]
    W32_DEQUANT16
    uint4 l0, l1, h0, h1, sc; uint32_t dh; int s0, nsh, fsh;
    __device__ __forceinline__ void load(const uint8_t * row, int nb, int ks, int sub) {
        // block b, 128-half hh, nibble nib (0: elements 0..63 of the half, 1: 64..127), 32-run sub:
        // ql bytes 64*hh + 32*sub + l (nibble nib), qh bytes 32*hh + l, 2-bit field 2*nib + sub      (ggml-quants.c:1690-1719)
        const int b = ks >> 2, hh = (ks >> 1) & 1, nib = ks & 1;
        const uint8_t * ql = row + 128 * (int64_t) b + 64 * hh + 32 * sub;
        l0 = ldg16_nt(ql); l1 = ldg16_nt(ql + 16);
        const uint8_t * qh = row + 128 * (int64_t) nb + 64 * b + 32 * hh;
        h0 = ldg16(qh); h1 = ldg16(qh + 16);
        sc = ldg16(row + 192 * (int64_t) nb + 16 * b);
        dh = ldg<uint16_t>(row + 208 * (int64_t) nb + 2 * b);
        s0 = 8 * hh + 4 * nib + 2 * sub; nsh = 4 * nib; fsh = 2 * (2 * nib + sub);
    }
    // scale of sub-block s (0..15) from the permuted 16-byte group (layout.hip: device byte 2j <- 8h+2cc+p, 2j+1 <- +4)
    __device__ __forceinline__ int scale(int s) const {
        const int hh = s >> 3, t = s & 7, hi = t >> 2, jj = 4 * hh + (t & 3);
        const int dev = 2 * jj + hi;
        // 64-bit shifts, not a 4-way select between the fields: the compiler turns that into an indexed load from a scratch copy
        const uint64_t lo = (uint64_t) sc.x | ((uint64_t) sc.y << 32), hi64 = (uint64_t) sc.z | ((uint64_t) sc.w << 32);
        return (int) (int8_t) (((dev < 8 ? lo : hi64) >> (8 * (dev & 7))) & 0xFFu);
    }
    __device__ __forceinline__ void dequant(uint32_t * out) const {
        const float d = h2f(dh);
        const float a0 = d * (float) scale(s0), a1 = d * (float) scale(s0 + 1);      // 16 elements each
        const uint32_t lw[8] = { l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w };
        const uint32_t hw[8] = { h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t q = ((lw[i] >> nsh) & 0x0F0F0F0Fu) | (((hw[i] >> fsh) & 0x03030303u) << 4);
            const float a = i < 4 ? a0 : a1;
            bytes4_to_bf16(q, a, -32.0f * a, out[2 * i], out[2 * i + 1]);
        }
    }
};

template <> struct W32<MI355Q_TYPE_Q8_0> {          // planar [qs 32*nb][d 2*nb]; 32 k = 1 block
    W32_DEQUANT16
    uint4 q0, q1; uint32_t dh;
    __device__ __forceinline__ void load(const uint8_t * row, int nb, int ks, int sub) {
        const int b = 2 * ks + sub;
        const uint8_t * p = row + 32 * (int64_t) b;
        q0 = ldg16_nt(p); q1 = ldg16_nt(p + 16);
        dh = ldg<uint16_t>(row + 32 * (int64_t) nb + 2 * b);
    }
    __device__ __forceinline__ void dequant(uint32_t * out) const {     // dequantize_row_q8_0, ggml-quants.c:349-363
        const float d = h2f(dh);
        const uint32_t qw[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
#pragma unroll
        for (int i = 0; i < 8; ++i) bytes4_to_bf16(qw[i] ^ 0x80808080u, d, -128.0f * d, out[2 * i], out[2 * i + 1]);   // int8 = (uint8 ^ 0x80) - 128
    }
};

template <> struct W32<MI355Q_TYPE_Q4_0> {          // planar [qs 16*nb][d 2*nb]; 32 k = 1 block: low nibbles = first 16, high = last 16
    W32_DEQUANT16
    uint4 q0; uint32_t dh;
    __device__ __forceinline__ void load(const uint8_t * row, int nb, int ks, int sub) {
        const int b = 2 * ks + sub;
        q0 = ldg16_nt(row + 16 * (int64_t) b);
        dh = ldg<uint16_t>(row + 16 * (int64_t) nb + 2 * b);
    }
    __device__ __forceinline__ void dequant(uint32_t * out) const {     // dequantize_row_q4_0, ggml-quants.c:255-273
        const float d = h2f(dh);
        const uint32_t qw[4] = { q0.x, q0.y, q0.z, q0.w };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bytes4_to_bf16(qw[i] & 0x0F0F0F0Fu, d, -8.0f * d, out[2 * i], out[2 * i + 1]);
            bytes4_to_bf16((qw[i] >> 4) & 0x0F0F0F0Fu, d, -8.0f * d, out[8 + 2 * i], out[8 + 2 * i + 1]);
        }
    }
};

// ---- IQ4_NL / IQ4_XS: 4-bit indices into the 16-entry non-linear code book (kvalues_iq4nl) ----
// four nibbles held as bytes of a dword -> four int8 code-book values: two v_perm_b32 + a select on bit 3 (as gemv_stream.cuh iq4_lut4)
__device__ __forceinline__ uint32_t mmq_iq4_lut4(uint32_t nib) {
    const uint32_t sel = nib & 0x07070707u;
    const uint32_t a = __builtin_amdgcn_perm(0xF6EADDCFu, 0xBFAD9881u, sel);      // entries 0..7:  -127,-104,-83,-65,-49,-35,-22,-10
    const uint32_t b = __builtin_amdgcn_perm(0x71594535u, 0x26190D01u, sel);      // entries 8..15: 1,13,25,38,53,69,89,113
    const uint32_t m = ((nib >> 3) & 0x01010101u) * 0xFFu;
    return (a & ~m) | (b & m);
}
// 4 SIGNED bytes of a dword -> a*byte as 4 floats -> 2 packed bf16 dwords
__device__ __forceinline__ void sbytes4_to_bf16(uint32_t v, float a, uint32_t & lo, uint32_t & hi) {
    const f32x2p q01 = { (float) (int8_t) (v & 0xFFu), (float) (int8_t) ((v >> 8) & 0xFFu) };
    const f32x2p q23 = { (float) (int8_t) ((v >> 16) & 0xFFu), (float) (int8_t) (v >> 24) };
    const f32x2p aa = { a, a };
    const f32x2p w01 = q01 * aa, w23 = q23 * aa;
    lo = pack_bf16(w01.x, w01.y); hi = pack_bf16(w23.x, w23.y);
}
template <> struct W32<MI355Q_TYPE_IQ4_NL> {        // planar [qs 16*nb][d 2*nb]; 32 k = 1 block: low nibbles = first 16, high = last 16
    W32_DEQUANT16
    uint4 q0; uint32_t dh;
    __device__ __forceinline__ void load(const uint8_t * row, int nb, int ks, int sub) {
        const int b = 2 * ks + sub;
        q0 = ldg16_nt(row + 16 * (int64_t) b);
        dh = ldg<uint16_t>(row + 16 * (int64_t) nb + 2 * b);
    }
    __device__ __forceinline__ void dequant(uint32_t * out) const {     // dequantize_row_iq4_nl, ggml-quants.c:2436-2452: y = d * kvalues_iq4nl[nibble]
        const float d = h2f(dh);
        const uint32_t qw[4] = { q0.x, q0.y, q0.z, q0.w };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sbytes4_to_bf16(mmq_iq4_lut4(qw[i] & 0x0F0F0F0Fu), d, out[2 * i], out[2 * i + 1]);
            sbytes4_to_bf16(mmq_iq4_lut4((qw[i] >> 4) & 0x0F0F0F0Fu), d, out[8 + 2 * i], out[8 + 2 * i + 1]);
        }
    }
};
template <> struct W32<MI355Q_TYPE_IQ4_XS> {        // planar [qs 128*nb][hdr 8*nb: d, scales_h, scales_l[4]]; 32 k = sub-block ib of block b
    W32_DEQUANT16
    uint4 q0; uint2 h; int ib;
    __device__ __forceinline__ void load(const uint8_t * row, int nb, int ks, int sub) {
        const int b = ks >> 2; ib = 2 * (ks & 3) + sub;
        q0 = ldg16_nt(row + 128 * (int64_t) b + 16 * ib);
        h  = ldg<uint2>(row + 128 * (int64_t) nb + 8 * b);
    }
    __device__ __forceinline__ void dequant(uint32_t * out) const {     // dequantize_row_iq4_xs, ggml-quants.c:2454-2475: dl = d * (ls - 32)
        const uint32_t scales_h = h.x >> 16;
        const int ls = (int) ((h.y >> (8 * (ib >> 1) + 4 * (ib & 1))) & 0x0Fu) | (int) (((scales_h >> (2 * ib)) & 3u) << 4);
        const float dl = h2f(h.x & 0xFFFFu) * (float) (ls - 32);
        const uint32_t qw[4] = { q0.x, q0.y, q0.z, q0.w };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sbytes4_to_bf16(mmq_iq4_lut4(qw[i] & 0x0F0F0F0Fu), dl, out[2 * i], out[2 * i + 1]);
            sbytes4_to_bf16(mmq_iq4_lut4((qw[i] >> 4) & 0x0F0F0F0Fu), dl, out[8 + 2 * i], out[8 + 2 * i + 1]);
        }
    }
};

// ---- the kernel ---------------------------------------------------------------------------------
// Tile BM x BN x 128; 256 threads = 2 x 2 waves, each a (BM/2) x (BN/2) output tile of 16x16x32 MFMAs.  Thread jobs per
// K-step: BM*4/256 dequant units (row, 32-k quarter) and BN*4/256 activation quarters (64 B).  Shapes that would leave
// CUs idle or with a single workgroup (one wave per SIMD hides no latency) get smaller tiles: 64 x 64 tiles need 35 KiB
// of LDS, so four workgroups share a CU.
template <int T, int MMQ_BM, int MMQ_BN>
__global__ void __launch_bounds__(MMQ_BN == 256 ? 512 : MMQ_THREADS, MMQ_BN == 256 ? 1 : (MMQ_BM == 64 && MMQ_BN == 64) ? 4 : 2)
k_mmq_bf16(const uint8_t * __restrict__ w, int64_t w_stride, const uint16_t * __restrict__ xb /* bf16 [n][k] */,
           float * __restrict__ y, int64_t y_stride, int m, int n, int k, int n_split, int64_t split_stride /* floats between the partial outputs */,
           const MoeTiles moe) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t * Ws = lds;
    uint8_t * Xs = lds + MMQ_BM * MMQ_LDS_STRIDE;                          // BN rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 128 x 256 tiles run with 8 waves (2 x 4: the same 64 x 64 per wave as the 128 x 128 tile) -- one dequantized weight tile then serves
    // 256 tokens instead of 128, which is what the dequantization-bound types (Q6_K) pay for
    constexpr int NTHREADS = MMQ_BN == 256 ? 512 : MMQ_THREADS, WAVES_N = MMQ_BN == 256 ? 4 : 2;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    constexpr int WM = MMQ_BM / 2, WN = MMQ_BN / WAVES_N, MT = WM / 16, NT = WN / 16;
    constexpr int UW = MMQ_BM * 4 / NTHREADS, UX = MMQ_BN * 4 / NTHREADS;
    static_assert(UW >= 1 && UW <= 2 && UX >= 1 && UX <= 2, "one or two staging units per thread");
    const int m0 = blockIdx.x * MMQ_BM, n0 = blockIdx.y * MMQ_BN;
    if (moe.tile_expert) {                                    // grouped MUL_MAT_ID: this token tile's expert (uniform per workgroup)
        const int e = moe.tile_expert[n0 / moe.tile_tokens];
        if (e < 0) return;
        w += (int64_t) e * moe.expert_stride;
        n = moe.seg_end[e];
    }
    const int nb = (T == MI355Q_TYPE_Q8_0 || T == MI355Q_TYPE_Q4_0 || T == MI355Q_TYPE_IQ4_NL) ? k >> 5 : k >> 8;
    // split K (blockIdx.z): a shape whose 128 x 128 tiles do not fill the chip is cut along K instead of into smaller tiles (each weight is
    // then dequantized for 128 tokens, not 64); piece z writes its partial sums to y + z * split_stride, k_mmq_reduce adds them in order
    const int steps_all = k / MMQ_BK;
    const int step_lo = (int) ((int64_t) steps_all * blockIdx.z / n_split), steps = (int) ((int64_t) steps_all * (blockIdx.z + 1) / n_split) - step_lo;
    y += (int64_t) blockIdx.z * split_stride;

    // dequant units: (row, quarter) = ((tid + 256 u) >> 2, tid & 3), u < UW; out-of-range rows read row 0 (never stored).
    // (Named variables, not arrays indexed by u: the compiler leaves such small arrays in scratch memory.)
    const int quarter = tid & 3, r0 = tid >> 2, r1 = (tid + NTHREADS) >> 2;
    const uint8_t * wrow0 = w + (int64_t) (m0 + r0 < m ? m0 + r0 : 0) * w_stride;
    const uint8_t * wrow1 = w + (int64_t) (m0 + r1 < m ? m0 + r1 : 0) * w_stride;
    const uint16_t * xrow0 = xb + (int64_t) (n0 + r0 < n ? n0 + r0 : 0) * k + 32 * quarter;
    const uint16_t * xrow1 = xb + (int64_t) (n0 + r1 < n ? n0 + r1 : 0) * k + 32 * quarter;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4) { 0.f, 0.f, 0.f, 0.f };

    W32<T> wq0, wq1;
    uint4 xa0, xa1, xa2, xa3, xb0, xb1, xb2, xb3;
    auto fetch = [&](int kn) {                                // packed weights + bf16 activations of K-step kn -> registers
        wq0.load(wrow0, nb, 2 * kn + (quarter >> 1), quarter & 1);
        if constexpr (UW > 1) wq1.load(wrow1, nb, 2 * kn + (quarter >> 1), quarter & 1);
        const uint16_t * p0 = xrow0 + (int64_t) kn * MMQ_BK;
        xa0 = *(const uint4 *) p0; xa1 = *(const uint4 *) (p0 + 8); xa2 = *(const uint4 *) (p0 + 16); xa3 = *(const uint4 *) (p0 + 24);
        if constexpr (UX > 1) {
            const uint16_t * p1 = xrow1 + (int64_t) kn * MMQ_BK;
            xb0 = *(const uint4 *) p1; xb1 = *(const uint4 *) (p1 + 8); xb2 = *(const uint4 *) (p1 + 16); xb3 = *(const uint4 *) (p1 + 24);
        }
    };
    auto stage_w = [&](const W32<T> & q, int r) {             // dequantize one unit into the bf16 W tile
        uint32_t b0, b1, b2, b3, b4, b5, b6, b7, b8, b9, b10, b11, b12, b13, b14, b15;
        q.dequant16(b0, b1, b2, b3, b4, b5, b6, b7, b8, b9, b10, b11, b12, b13, b14, b15);
        uint8_t * dst = Ws + r * MMQ_LDS_STRIDE + 64 * quarter;
        *(uint4 *) (dst)      = make_uint4(b0, b1, b2, b3);   *(uint4 *) (dst + 16) = make_uint4(b4, b5, b6, b7);
        *(uint4 *) (dst + 32) = make_uint4(b8, b9, b10, b11); *(uint4 *) (dst + 48) = make_uint4(b12, b13, b14, b15);
    };
    fetch(step_lo);

    for (int ks = 0; ks < steps; ++ks) {
        // ---- stage step ks into LDS (registers were loaded one step ahead) ----
        stage_w(wq0, r0);
        if constexpr (UW > 1) stage_w(wq1, r1);
        {
            uint8_t * xdst = Xs + r0 * MMQ_LDS_STRIDE + 64 * quarter;
            *(uint4 *) xdst = xa0; *(uint4 *) (xdst + 16) = xa1; *(uint4 *) (xdst + 32) = xa2; *(uint4 *) (xdst + 48) = xa3;
        }
        if constexpr (UX > 1) {
            uint8_t * xdst = Xs + r1 * MMQ_LDS_STRIDE + 64 * quarter;
            *(uint4 *) xdst = xb0; *(uint4 *) (xdst + 16) = xb1; *(uint4 *) (xdst + 32) = xb2; *(uint4 *) (xdst + 48) = xb3;
        }
        __syncthreads();
        // ---- prefetch step ks+1 while the matrix cores work on step ks (the last step re-reads itself: no branch) ----
        fetch(step_lo + (ks + 1 < steps ? ks + 1 : ks));
        // ---- 4 k-slices of 32: A = W rows (lane: row l&15, k 8*(l>>4)..+7), B = tokens (lane: col l&15, same k) ----
        // (grouped launches: a wave whose token half lies past the end of the expert's segment has helped to stage the tile and skips the MFMAs)
        if (n0 + WN * wn < n)
#pragma unroll
        for (int kk = 0; kk < MMQ_BK / 32; ++kk) {
            bf16x8 af[MT], bfr[NT];
            const int koff = 2 * (32 * kk + 8 * (lane >> 4));
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i]  = *(const bf16x8 *) (Ws + (WM * wm + 16 * i + (lane & 15)) * MMQ_LDS_STRIDE + koff);
#pragma unroll
            for (int j = 0; j < NT; ++j) bfr[j] = *(const bf16x8 *) (Xs + (WN * wn + 16 * j + (lane & 15)) * MMQ_LDS_STRIDE + koff);
#pragma unroll
            for (int i = 0; i < MT; ++i)
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
        const int dr = moe.dst_row ? moe.dst_row[tok] : tok;                  // grouped MUL_MAT_ID: straight to the pair's row of the result
        if (dr < 0) continue;
        float * yr = (float *) ((char *) y + (int64_t) dr * y_stride);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mr = m0 + WM * wm + 16 * i + 4 * (lane >> 4);
            if (mr + 3 < m) *(f32x4 *) (yr + mr) = acc[i][j];
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (mr + r < m) yr[mr + r] = acc[i][j][r];
            }
        }
    }
}

// y[i] = part[0][i] + part[1][i] + ... (fixed order: deterministic)
__global__ void __launch_bounds__(256) k_mmq_reduce(const float * __restrict__ part, int n_split, int64_t split_stride, float * __restrict__ y, int64_t y_stride, int64_t m, int64_t n) {
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n * (m / 4); i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t r = i / (m / 4), c = i - r * (m / 4);
        f32x4 a = *(const f32x4 *) (part + r * m + 4 * c);
        for (int z = 1; z < n_split; ++z) a += *(const f32x4 *) (part + z * split_stride + r * m + 4 * c);
        *(f32x4 *) ((char *) y + r * y_stride + 16 * c) = a;
    }
}

void launch_mmq_reduce(const float * part, int n_split, int64_t split_stride, float * y, int64_t y_stride, int64_t m, int64_t n, hipStream_t stream) {
    const int64_t quads = n * (m / 4);
    hipLaunchKernelGGL(k_mmq_reduce, dim3((unsigned) ((quads + 255) / 256 < 4096 ? (quads + 255) / 256 : 4096)), dim3(256), 0, stream, part, n_split, split_stride, y, y_stride, m, n);
}

// ---- host side ------------------------------------------------------------------------------------
bool mmq_supported(int type, int64_t k) {
    switch (type) {
    case MI355Q_TYPE_Q4_K: case MI355Q_TYPE_Q5_K: case MI355Q_TYPE_Q6_K: case MI355Q_TYPE_IQ4_XS: return k % 256 == 0;
    case MI355Q_TYPE_Q8_0: case MI355Q_TYPE_Q4_0: case MI355Q_TYPE_IQ4_NL: return k % 128 == 0;
    default: return false;
    }
}

size_t mmq_i8_workspace(int64_t n, int64_t k);
size_t mmq_workspace(int64_t n, int64_t k) {                  // one scratch size for both MFMA tiers: bf16 activations, or the int8 image of mmq_i8.hip
    const size_t a = (size_t) (n * k * 2 + 255) & ~(size_t) 255, b = k % 256 == 0 ? mmq_i8_workspace(n, k) : 0;
    return a > b ? a : b;
}
// extra scratch behind mmq_workspace for split-K partial sums of an m-row matrix (0 when the shape is not split)
static int mmq_bf16_splits(int64_t m, int64_t n, int64_t k, int n_cu) {
    const int64_t t128 = ((m + 127) / 128) * ((n + 127) / 128);
    if (2 * t128 >= 3 * (int64_t) n_cu || k < 4096 || m % 4) return 1;      // fills the chip as it is / too little K to cut
    int s = (int) ((2 * (int64_t) n_cu + t128 - 1) / t128);                    // aim at two workgroups per CU
    const int max_s = (int) (k / MMQ_BK / 8);                                  // at least 8 steps per piece
    if (s > max_s) s = max_s;
    if (s > 8) s = 8;
    return s < 2 ? 1 : s;
}
size_t mmq_split_workspace(int64_t m, int64_t n, int64_t k, int n_cu) {
    const int s = mmq_bf16_splits(m, n, k, n_cu);
    return s > 1 ? (size_t) s * (size_t) n * (size_t) m * 4 + 256 : 0;
}

// w: planar device rows; x f32 [n][k] (row stride x_stride); workspace >= mmq_workspace(n,k); y f32 [n][m] (y_stride % 16 == 0)
// `prepare` = convert the activations to bf16 first (matrices multiplied with the same activations share the copy)
int launch_mmq_bf16(int type, const void * w, int64_t w_stride, const float * x, int64_t x_stride,
                    float * y, int64_t y_stride, int64_t m, int64_t n, int64_t k, void * workspace, size_t workspace_bytes, int n_cu, hipStream_t stream, bool prepare,
                    const MoeTiles * moe_p = nullptr) {
    MoeTiles moe = {}; if (moe_p) moe = *moe_p;
    if (!mmq_supported(type, k)) return MI355Q_ERR_UNSUPPORTED;
    if (m <= 0 || n <= 0) return MI355Q_OK;
    if ((y_stride & 15) || ((uintptr_t) y & 15)) return MI355Q_ERR_ALIGN;
    const int64_t pairs = n * k / 2;
    const int cgrid = (int) ((pairs + 255) / 256 < 8192 ? (pairs + 255) / 256 : 8192);
    if (prepare) hipLaunchKernelGGL(k_x_to_bf16, dim3(cgrid), dim3(256), 0, stream, x, x_stride, (uint32_t *) workspace, n, k);
    // The largest tile that still gives every CU about three workgroups: 128 x 128 amortizes the dequantization best, but a
    // pp512 matmul of a 4096-row matrix is only 128 such tiles for 256 CUs; one workgroup per CU (one wave per SIMD) hides no
    // latency at all.
    auto tiles = [&](int bm, int bn) { return ((m + bm - 1) / bm) * ((n + bn - 1) / bn); };
    int bm = 64, bn = 64;                                     // measured on pp512 shapes: 128x128 wins from ~1.5 workgroups per CU on
    static const int wide_env = getenv("MI355Q_MMQ_BF16_WIDE") ? atoi(getenv("MI355Q_MMQ_BF16_WIDE")) : -1;      // dev: 0 = never, 1 = whenever n >= 256
    // (Q6_K, whose dequantization dominates: 4096x14336 at N = 512 369 -> 425 TFLOP/s, 128256x4096 411 -> 496; Q5_K measures the same either way)
    const bool wide_type = type == MI355Q_TYPE_Q6_K;
    if (2 * tiles(128, 128) >= 3 * (int64_t) n_cu) { bm = 128; bn = 128; }
    else if (tiles(128, 64) >= 2 * (int64_t) n_cu) { bm = 128; bn = 64; }
    // ... or 128 x 128 tiles on K pieces (partial sums behind the activation copy in the scratch buffer, added up by k_mmq_reduce)
    if (moe.tile_expert) { bm = 128; bn = moe.tile_tokens; }   // (grouped: the token tile the segments were aligned to)
    int splits = moe.tile_expert ? 1 : mmq_bf16_splits(m, n, k, n_cu);
    const size_t part_off = mmq_workspace(n, k);
    if (splits > 1 && workspace_bytes < part_off + (size_t) splits * (size_t) n * (size_t) m * 4) splits = 1;
    float * yk = y; int64_t yk_stride = y_stride; int64_t split_stride = 0;
    if (splits > 1) { bm = 128; bn = 128; yk = (float *) ((char *) workspace + part_off); yk_stride = 4 * m; split_stride = n * m; }
    // 128 x 256 tiles (8 waves, one workgroup per CU): when the token count fills them and the 128-row tiles (x K pieces) still cover the chip
    if (!moe.tile_expert && bm == 128 && bn == 128 && n >= 256 && wide_env != 0 && (wide_env == 1 || wide_type) &&
        ((m + 127) / 128) * ((n + 255) / 256) * splits >= (int64_t) n_cu) bn = 256;
    const dim3 grid((unsigned) ((m + bm - 1) / bm), (unsigned) ((n + bn - 1) / bn), (unsigned) splits);
#define MI355Q_MMQ_LAUNCH(T, BM, BN) {                                                                                             \
        constexpr size_t lds_bytes = (size_t) (BM + BN) * MMQ_LDS_STRIDE;          /* 68 / 51 / 34 KiB */                           \
        static bool attr_set[64] = {};          /* the attribute is per device: the plugin drives every visible GPU from one process */ \
        int dev_ = 0; (void) hipGetDevice(&dev_); dev_ = dev_ >= 0 && dev_ < 64 ? dev_ : 0;                                          \
        if (!attr_set[dev_]) {                                                                                                     \
            if (hipFuncSetAttribute((const void *) k_mmq_bf16<T, BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes) != hipSuccess) \
                return MI355Q_ERR_HIP;                                                                                             \
            attr_set[dev_] = true;                                                                                                 \
        }                                                                                                                          \
        hipLaunchKernelGGL((k_mmq_bf16<T, BM, BN>), grid, dim3(BN == 256 ? 512 : MMQ_THREADS), lds_bytes, stream, (const uint8_t *) w, w_stride, \
                           (const uint16_t *) workspace, yk, yk_stride, (int) m, (int) n, (int) k, splits, split_stride, moe); }
#define MI355Q_MMQ_CASE(T) case T: if (bm == 128 && bn == 256) MI355Q_MMQ_LAUNCH(T, 128, 256) else if (bm == 128 && bn == 128) MI355Q_MMQ_LAUNCH(T, 128, 128) else if (bm == 128) MI355Q_MMQ_LAUNCH(T, 128, 64) else MI355Q_MMQ_LAUNCH(T, 64, 64) break;
    switch (type) {
        MI355Q_MMQ_CASE(MI355Q_TYPE_Q4_K) MI355Q_MMQ_CASE(MI355Q_TYPE_Q5_K) MI355Q_MMQ_CASE(MI355Q_TYPE_Q6_K)
        MI355Q_MMQ_CASE(MI355Q_TYPE_Q8_0) MI355Q_MMQ_CASE(MI355Q_TYPE_Q4_0) MI355Q_MMQ_CASE(MI355Q_TYPE_IQ4_NL) MI355Q_MMQ_CASE(MI355Q_TYPE_IQ4_XS)
    default: return MI355Q_ERR_UNSUPPORTED;
    }
#undef MI355Q_MMQ_CASE
#undef MI355Q_MMQ_LAUNCH
    if (splits > 1) {
        launch_mmq_reduce(yk, splits, split_stride, y, y_stride, m, n, stream);
    }
    return hipGetLastError() == hipSuccess ? MI355Q_OK : MI355Q_ERR_HIP;
}

} // namespace mi355q
