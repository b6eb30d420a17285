// api.hip -- the C-ABI of libmi355q.so (include/mi355q.h): argument checking, tier dispatch, thin
// HIP memory helpers.  No ggml, no torch.  Every entry point returns an MI355Q_* code.
#include "mi355q_common.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace mi355q {
// kernels (other translation units)
int launch_quantize_act(int act_type, const float * x, int64_t x_stride, void * out, int64_t n, int64_t k, int flags, hipStream_t stream);
struct GenericMoe { const int32_t * ids; int64_t ids_stride; int64_t expert_stride; int n_used; int x_ne1; int n_expert; int pad; };
bool mmq_generic_supported(int type, int64_t k);
size_t mmq_generic_workspace(int64_t n, int64_t k);
int launch_mmq_generic(int type, const void * w, int64_t w_stride, const float * x, int64_t x_stride, float * y, int64_t y_stride,
                       int64_t m, int64_t n, int64_t k, void * workspace, size_t workspace_bytes, hipStream_t stream, const MoeTiles * moe = nullptr);
int launch_gemv_generic(int type, const void * w, int64_t w_stride, const void * act, int64_t act_stride,
                        float * y, int64_t y_stride, int64_t m, int64_t n, int64_t k, hipStream_t stream, const GenericMoe * moe);
struct MoeArgs { const int32_t * ids; int64_t ids_stride; int64_t expert_stride; int64_t x_stride2; int n_used; int x_ne1; int n_expert; int n_pairs; };
int launch_gemv_fast(const mi355q_mat * mats, int n_mats, const float * x, int64_t x_stride, int ncols, int64_t k,
                     int flags, int n_cu, hipStream_t stream, const MoeArgs * moe);
int gemv_fast_family(int type);
int gemv_fast_max_cols(int type, int64_t k);
int launch_pack(int type, void * dst, const void * src, int64_t nrows, int64_t k, bool pack, hipStream_t stream);
bool mmq_supported(int type, int64_t k);
size_t mmq_workspace(int64_t n, int64_t k);
int launch_mmq_bf16(int type, const void * w, int64_t w_stride, const float * x, int64_t x_stride, float * y, int64_t y_stride,
                    int64_t m, int64_t n, int64_t k, void * workspace, size_t workspace_bytes, int n_cu, hipStream_t stream, bool prepare, const MoeTiles * moe = nullptr);
size_t mmq_split_workspace(int64_t m, int64_t n, int64_t k, int n_cu);
bool mmq_i8_supported(int type, int64_t k);
int launch_mmq_i8_multi(const mi355q_mat * mats, int n_mats, const float * x, int64_t x_stride, int64_t n, int64_t k,
                        void * workspace, size_t workspace_bytes, int n_cu, hipStream_t stream, bool prepare, const MoeTiles * moe = nullptr);
size_t mmq_i8_split_workspace(int64_t m, int64_t n, int64_t k, int n_cu);
bool mmq_q80_supported(int type, int64_t k);
int launch_mmq_q80_multi(const mi355q_mat * mats, int n_mats, const float * x, int64_t x_stride, int64_t n, int64_t k,
                         void * workspace, size_t workspace_bytes, int n_cu, hipStream_t stream, bool prepare, bool round_even, const MoeTiles * moe = nullptr);

// tier choice for planar rows: GEMV (exact integer dot, fused quantizer) up to 8 activation rows, MFMA tier above
static bool use_mmq(int type, int64_t n, int64_t k, int flags) {
    if (!mmq_supported(type, k) || (flags & MI355Q_FLAG_FORCE_GEMV)) return false;
    return n > 8 || (flags & MI355Q_FLAG_FORCE_MMQ);
}

static thread_local char t_err[512] = "";
static int fail(int code, const char * fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(t_err, sizeof(t_err), fmt, ap); va_end(ap);
    return code;
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(MI355Q_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)
#define MQ_TRY(expr)  do { int r_ = (expr); if (r_ != MI355Q_OK) { if (!t_err[0]) fail(r_, "%s failed (%d)", #expr, r_); return r_; } } while (0)

static int cu_count() {
    static int cached[64]; static bool have[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!have[dev]) {
        hipDeviceProp_t p;
        cached[dev] = hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 256;
        have[dev] = true;
    }
    return cached[dev];
}

static int64_t moe_align256(int64_t v) { return (v + 255) & ~(int64_t) 255; }
extern "C" size_t mi355q_mul_mat_workspace(int type, int64_t m, int64_t n, int64_t k);

// ---- grouped MUL_MAT_ID (many (token, slot) pairs), entirely on the device -------------------------------------------------------------
// The reference groups the rows of a batch by expert on the HOST (CPU backend: ggml-cpu.c:1628-1643; CUDA backend: ids copied to the host with a
// stream synchronize, ggml-cuda.cu:2008-2011) and multiplies expert by expert.  Here one workgroup counting-sorts the pairs by expert into
// segments aligned to the matrix-core token tile, the rows are gathered in that order, ONE launch of the prefill tier walks all token tiles
// (each tile knows its expert, csrc/mi355q_common.h MoeTiles) and the rows are scattered back: no host round trip, so the graph stays
// capturable, and an expert with few rows no longer costs a launch of its own.
//   order[slot]        pair index of the gathered row, -1 for padding        slot_of_pair[p]  where pair p went, -1 for an invalid expert id
//   seg_end[e]         one past the last slot of expert e                   tile_expert[t]   expert of token tile t, -1 beyond the last segment
__global__ void __launch_bounds__(1024) k_moe_sort(const char * __restrict__ ids, int64_t ids_stride, int n_used, int pairs, int n_expert, int tile, int n_slots,
                                                   int32_t * __restrict__ order, int32_t * __restrict__ slot_of_pair, int32_t * __restrict__ seg_end,
                                                   int32_t * __restrict__ tile_expert) {
    extern __shared__ int moe_lds[];                          // cnt[n_expert], cur[n_expert]
    int * cnt = moe_lds, * cur = moe_lds + n_expert;
    const int tid = threadIdx.x;
    for (int e = tid; e < n_expert; e += 1024) cnt[e] = 0;
    for (int i = tid; i < n_slots; i += 1024) order[i] = -1;
    for (int t = tid; t < n_slots / tile; t += 1024) tile_expert[t] = -1;
    __syncthreads();
    auto expert_of = [&](int p) { const int t = p / n_used, u = p - t * n_used; return *(const int32_t *) (ids + (int64_t) t * ids_stride + 4 * u); };
    for (int p = tid; p < pairs; p += 1024) { const int e = expert_of(p); if (e >= 0 && e < n_expert) atomicAdd(&cnt[e], 1); }
    __syncthreads();
    if (tid == 0) {                                           // segment starts: running sum, every segment aligned to the token tile (n_expert <= a few hundred)
        int begin = 0;
        for (int e = 0; e < n_expert; ++e) { cur[e] = begin; const int end = begin + cnt[e]; seg_end[e] = end; begin = (end + tile - 1) / tile * tile; }
    }
    __syncthreads();
    for (int e = tid; e < n_expert; e += 1024) {
        const int b = cur[e], en = seg_end[e];
        for (int t = b / tile; t * tile < en; ++t) tile_expert[t] = e;
    }
    __syncthreads();                                          // (cur[] is read above as the segment start before it becomes the fill cursor)
    for (int p = tid; p < pairs; p += 1024) {
        const int e = expert_of(p);
        int slot = -1;
        if (e >= 0 && e < n_expert) { slot = atomicAdd(&cur[e], 1); order[slot] = p; }
        slot_of_pair[p] = slot;
    }
}
// xg[slot][:] = x[tok(order[slot])][slot_u(order[slot]) % x_ne1][:]      (padding slots are left as they are: their outputs are never stored)
__global__ void __launch_bounds__(256) k_moe_gather(const char * __restrict__ x, int64_t x_stride1, int64_t x_stride2, int n_used, int x_ne1,
                                                    const int32_t * __restrict__ order, float * __restrict__ xg, int64_t k) {
    const int p = order[blockIdx.x];
    if (p < 0) return;
    const int t = p / n_used, u = p - t * n_used;
    const float4 * src = (const float4 *) (x + (int64_t) t * x_stride2 + (int64_t) (u % x_ne1) * x_stride1);
    float4 * dst = (float4 *) (xg + (int64_t) blockIdx.x * k);
    for (int64_t i = threadIdx.x; i < k / 4; i += 256) dst[i] = src[i];
}
// The matrix-core kernels store every result row straight into y[pair] (MoeTiles::dst_row = order).  What is left for a pass of its own: a pair
// whose expert id is out of range was given no slot -- its row becomes NaN (the reference asserts on such ids; a device kernel cannot)
__global__ void __launch_bounds__(256) k_moe_nan_rows(const int32_t * __restrict__ slot_of_pair, float * __restrict__ y, int64_t m) {
    if (slot_of_pair[blockIdx.x] >= 0) return;
    float * dst = y + (int64_t) blockIdx.x * m;
    for (int64_t i = threadIdx.x; i < m; i += 256) dst[i] = __int_as_float(0x7FC00000);
}
constexpr int64_t MOE_GROUPED_MIN_PAIRS = 17;      // below: one GEMV column per pair with the ids read on the device
// Token tile the expert segments are aligned to.  The integer tiers (Q4_K, Q8_0) run 64-token tiles at full speed.  The bf16 tier dequantizes
// its weight tile once per workgroup, so there a workgroup covers 128 tokens of one expert and the waves whose 64-token half lies past the
// segment's end skip their MFMAs: a half-empty tile costs the dequantization it would have cost anyway, not the matrix work.
static int moe_tile(int type, int64_t k, int64_t pairs, int64_t n_expert) {
    static const int forced = getenv("MI355Q_MOE_TILE") ? atoi(getenv("MI355Q_MOE_TILE")) : 0;      // dev: 64 / 128
    (void) pairs; (void) n_expert;
    if (forced == 64 || forced == 128) return forced;
    static const bool no_i8 = getenv("MI355Q_NO_MMQ_I8") != nullptr, no_q80 = getenv("MI355Q_NO_MMQ_Q80") != nullptr;
    const bool integer_tier = (!no_i8 && mmq_i8_supported(type, k)) || (!no_q80 && mmq_q80_supported(type, k));
    if (!mmq_supported(type, k)) return 64;                     // canonical rows (mmq_generic.hip): 64-token tiles
    return integer_tier ? 64 : 128;
}
static int64_t moe_slots(int type, int64_t k, int64_t pairs, int64_t n_expert) {     // gathered rows incl. the alignment padding of every segment, a multiple of 128
    const int tile = moe_tile(type, k, pairs, n_expert);
    return (pairs + n_expert * (tile - 1) + 127) / 128 * 128;
}
static size_t moe_grouped_workspace(int type, int64_t m, int64_t k, int64_t pairs, int64_t n_expert) {
    // [order: slots][slot_of_pair: pairs][seg_end: E][tile_expert: slots / 64] i32 | xg: slots x k f32 | scratch of the prefill tier
    const int64_t slots = moe_slots(type, k, pairs, n_expert);
    return (size_t) (moe_align256(4 * (slots + pairs + n_expert + slots / 64)) + moe_align256(4 * slots * k)) + mi355q_mul_mat_workspace(type, m, slots, k);
}
static bool moe_grouped_ok(int type, int64_t m, int64_t k, const void * x, int64_t x_stride1, int64_t x_stride2) {
    return mmq_supported(type, k) && m % 4 == 0 && k % 4 == 0 && !(((uintptr_t) x | (uintptr_t) x_stride1 | (uintptr_t) x_stride2) & 15);
}
// canonical rows: the grouped form runs on the batched canonical tier
static bool moe_grouped_canonical_ok(const TypeInfo * t, int type, int64_t k, const void * x, int64_t x_stride1, int64_t x_stride2) {
    return !is_planar(t, k) && mmq_generic_supported(type, k) && k % 4 == 0 && !(((uintptr_t) x | (uintptr_t) x_stride1 | (uintptr_t) x_stride2) & 15);
}
} // namespace mi355q

using namespace mi355q;

extern "C" {

int mi355q_api_version(void) { return MI355Q_API_VERSION; }
const char * mi355q_last_error(void) { return t_err; }
__attribute__((visibility("hidden"))) void mi355q_set_error(const char * msg) { fail(0, "%s", msg); }   // for the other translation units

int mi355q_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok == n ? n : 0;      // all visible devices must be gfx950: the code objects are built for it only
}

int mi355q_set_device(int device) { HIP_TRY(hipSetDevice(device)); return MI355Q_OK; }

int mi355q_device_info(int device, char * name, size_t name_len, size_t * free_bytes, size_t * total_bytes, int * compute_units) {
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name && name_len) snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (free_bytes || total_bytes) {
        int cur = 0; HIP_TRY(hipGetDevice(&cur)); HIP_TRY(hipSetDevice(device));
        size_t f = 0, t = 0; HIP_TRY(hipMemGetInfo(&f, &t)); HIP_TRY(hipSetDevice(cur));
        if (free_bytes) *free_bytes = f;
        if (total_bytes) *total_bytes = t;
    }
    return MI355Q_OK;
}

int     mi355q_type_supported(int type) { const TypeInfo * t = type_info(type); return t && t->act >= 0; }
int64_t mi355q_blck_size(int type) { const TypeInfo * t = type_info(type); return t ? t->blck : 0; }
int64_t mi355q_type_size(int type) { const TypeInfo * t = type_info(type); return t ? t->bsize : 0; }
int64_t mi355q_row_size(int type, int64_t k) { const TypeInfo * t = type_info(type); return t && k % t->blck == 0 ? k / t->blck * t->bsize : 0; }
int     mi355q_act_type(int type) { const TypeInfo * t = type_info(type); return t ? t->act : -1; }
int     mi355q_weights_are_planar(int type, int64_t k) { return is_planar(type_info(type), k) ? 1 : 0; }

// ---- memory helpers ----
int mi355q_malloc(void ** p, size_t bytes) { HIP_TRY(hipMalloc(p, bytes)); return MI355Q_OK; }
int mi355q_free(void * p) { HIP_TRY(hipFree(p)); return MI355Q_OK; }
int mi355q_memset(void * p, int v, size_t bytes, void * stream) { HIP_TRY(hipMemsetAsync(p, v, bytes, (hipStream_t) stream)); return MI355Q_OK; }
static int copy(void * dst, const void * src, size_t bytes, hipMemcpyKind kind, void * stream) {
    if (stream) { HIP_TRY(hipMemcpyAsync(dst, src, bytes, kind, (hipStream_t) stream)); }
    else        { HIP_TRY(hipMemcpy(dst, src, bytes, kind)); }
    return MI355Q_OK;
}
int mi355q_memcpy_h2d(void * d, const void * s, size_t n, void * st) { return copy(d, s, n, hipMemcpyHostToDevice, st); }
int mi355q_memcpy_d2h(void * d, const void * s, size_t n, void * st) { return copy(d, s, n, hipMemcpyDeviceToHost, st); }
int mi355q_memcpy_d2d(void * d, const void * s, size_t n, void * st) { return copy(d, s, n, hipMemcpyDeviceToDevice, st); }
int mi355q_host_malloc(void ** p, size_t bytes) { HIP_TRY(hipHostMalloc(p, bytes, hipHostMallocDefault)); return MI355Q_OK; }
int mi355q_host_free(void * p) { HIP_TRY(hipHostFree(p)); return MI355Q_OK; }
int mi355q_memcpy_peer(void * dst, int dst_device, const void * src, int src_device, size_t bytes, void * stream) {
    if (dst_device == src_device) { HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t) stream)); return MI355Q_OK; }
    // direct xGMI access between the two devices, enabled once per ordered pair (without it the runtime stages peer copies through the host)
    static bool tried[64][64];
    if (dst_device >= 0 && dst_device < 64 && src_device >= 0 && src_device < 64 && !tried[src_device][dst_device]) {
        tried[src_device][dst_device] = tried[dst_device][src_device] = true;
        int cur = 0, can = 0;
        if (hipGetDevice(&cur) == hipSuccess) {
            for (int pass = 0; pass < 2; ++pass) {
                const int a = pass ? dst_device : src_device, b = pass ? src_device : dst_device;
                if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can && hipSetDevice(a) == hipSuccess) {
                    const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void) hipGetLastError();
                }
            }
            (void) hipSetDevice(cur);
        }
    }
    HIP_TRY(hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, (hipStream_t) stream));
    return MI355Q_OK;
}
int mi355q_event_create(void ** e) { hipEvent_t h; HIP_TRY(hipEventCreateWithFlags(&h, hipEventDisableTiming)); *e = h; return MI355Q_OK; }
int mi355q_event_destroy(void * e) { HIP_TRY(hipEventDestroy((hipEvent_t) e)); return MI355Q_OK; }
int mi355q_event_record(void * e, void * s) { HIP_TRY(hipEventRecord((hipEvent_t) e, (hipStream_t) s)); return MI355Q_OK; }
int mi355q_event_wait(void * s, void * e) { HIP_TRY(hipStreamWaitEvent((hipStream_t) s, (hipEvent_t) e, 0)); return MI355Q_OK; }
int mi355q_event_synchronize(void * e) { HIP_TRY(hipEventSynchronize((hipEvent_t) e)); return MI355Q_OK; }
int mi355q_stream_create(void ** s) { hipStream_t h; HIP_TRY(hipStreamCreateWithFlags(&h, hipStreamNonBlocking)); *s = h; return MI355Q_OK; }
int mi355q_stream_destroy(void * s) { HIP_TRY(hipStreamDestroy((hipStream_t) s)); return MI355Q_OK; }
int mi355q_stream_synchronize(void * s) { HIP_TRY(hipStreamSynchronize((hipStream_t) s)); return MI355Q_OK; }
int mi355q_device_synchronize(void) { HIP_TRY(hipDeviceSynchronize()); return MI355Q_OK; }

// ---- weights ----
static int check_rows(int type, int64_t nrows, int64_t k, const void * dev) {
    const TypeInfo * t = type_info(type);
    if (!t || t->act < 0) return fail(MI355Q_ERR_UNSUPPORTED, "type %d is not a supported weight type", type);
    if (nrows < 0 || k <= 0 || k % t->blck) return fail(MI355Q_ERR_SHAPE, "k=%lld is not a multiple of the %d-element block", (long long) k, t->blck);
    if (is_planar(t, k) && ((uintptr_t) dev & 15)) return fail(MI355Q_ERR_ALIGN, "device rows must be 16-byte aligned");
    return MI355Q_OK;
}

int mi355q_weights_pack_d2d(int type, void * dst, const void * src, int64_t nrows, int64_t k, void * stream) {
    MQ_TRY(check_rows(type, nrows, k, dst));
    return launch_pack(type, dst, src, nrows, k, true, (hipStream_t) stream);
}
int mi355q_weights_unpack_d2d(int type, void * dst, const void * src, int64_t nrows, int64_t k, void * stream) {
    MQ_TRY(check_rows(type, nrows, k, src));
    return launch_pack(type, dst, src, nrows, k, false, (hipStream_t) stream);
}

static int staged_copy(int type, void * dev, void * host, int64_t nrows, int64_t k, bool upload, hipStream_t stream) {
    MQ_TRY(check_rows(type, nrows, k, dev));
    const int64_t rb = mi355q_row_size(type, k);
    if (nrows == 0) return MI355Q_OK;
    if (!is_planar(type_info(type), k)) {
        if (upload) { HIP_TRY(hipMemcpyAsync(dev, host, (size_t) (rb * nrows), hipMemcpyHostToDevice, stream)); }
        else        { HIP_TRY(hipMemcpyAsync(host, dev, (size_t) (rb * nrows), hipMemcpyDeviceToHost, stream)); }
        HIP_TRY(hipStreamSynchronize(stream));
        return MI355Q_OK;
    }
    int64_t chunk_rows = (int64_t) (64u << 20) / rb; if (chunk_rows < 1) chunk_rows = 1; if (chunk_rows > nrows) chunk_rows = nrows;
    void * tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, (size_t) (chunk_rows * rb)));
    int rc = MI355Q_OK;
    for (int64_t r0 = 0; r0 < nrows && rc == MI355Q_OK; r0 += chunk_rows) {
        const int64_t nr = nrows - r0 < chunk_rows ? nrows - r0 : chunk_rows;
        char * d = (char *) dev + r0 * rb; char * h = (char *) host + r0 * rb;
        hipError_t e;
        if (upload) {
            e = hipMemcpyAsync(tmp, h, (size_t) (nr * rb), hipMemcpyHostToDevice, stream);
            if (e == hipSuccess) rc = launch_pack(type, d, tmp, nr, k, true, stream);
        } else {
            rc = launch_pack(type, tmp, d, nr, k, false, stream);
            e = rc == MI355Q_OK ? hipMemcpyAsync(h, tmp, (size_t) (nr * rb), hipMemcpyDeviceToHost, stream) : hipSuccess;
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream);       // tmp is reused by the next chunk
        if (e != hipSuccess) rc = fail(MI355Q_ERR_HIP, "staged weight copy: %s", hipGetErrorString(e));
    }
    (void) hipFree(tmp);
    return rc;
}
int mi355q_weights_upload(int type, void * dst_dev, const void * src_host, int64_t nrows, int64_t k, void * stream) {
    return staged_copy(type, dst_dev, (void *) src_host, nrows, k, true, (hipStream_t) stream);
}
int mi355q_weights_download(int type, void * dst_host, const void * src_dev, int64_t nrows, int64_t k, void * stream) {
    return staged_copy(type, (void *) src_dev, dst_host, nrows, k, false, (hipStream_t) stream);
}

// ---- activation quantizer ----
int mi355q_quantize_act(int act_type, const float * x, int64_t x_stride_bytes, void * out, int64_t n, int64_t k, int flags, void * stream) {
    if (x_stride_bytes % 4 || ((uintptr_t) x & 3) || ((uintptr_t) out & 3)) return fail(MI355Q_ERR_ALIGN, "quantize_act: 4-byte alignment required");
    if (n > 65535) return fail(MI355Q_ERR_UNSUPPORTED, "quantize_act: n > 65535");
    const int rc = launch_quantize_act(act_type, x, x_stride_bytes, out, n, k, flags, (hipStream_t) stream);
    return rc == MI355Q_OK ? rc : fail(rc, "quantize_act(act=%d,n=%lld,k=%lld) failed", act_type, (long long) n, (long long) k);
}

// ---- MUL_MAT ----
static int64_t align256(int64_t v) { return (v + 255) & ~(int64_t) 255; }

size_t mi355q_mul_mat_workspace(int type, int64_t m, int64_t n, int64_t k) {
    const TypeInfo * t = type_info(type);
    if (!t || t->act < 0 || k % t->blck) return 0;
    // GEMV tier: fused prologue, no scratch; matrix-core tiers: the prepared activations (+ split-K partial sums for shapes that are cut along K)
    if (is_planar(t, k)) return mmq_supported(type, k) && n > 8 ? mmq_workspace(n, k) + (mmq_i8_supported(type, k) ? mmq_i8_split_workspace(m, n, k, cu_count()) : mmq_split_workspace(m, n, k, cu_count())) : (mmq_supported(type, k) ? mmq_workspace(n, k) : 0);
    // canonical rows: quantized activations for the per-column tier; bf16 activations for the batched tier (mmq_generic.hip)
    const size_t cols = (size_t) align256(mi355q_row_size(t->act, k) * n);
    const size_t batched = n > 8 && mmq_generic_supported(type, k) ? mmq_generic_workspace(n, k) : 0;
    return cols > batched ? cols : batched;
}

static int mul_mat_checks(const TypeInfo * t, const void * w, int64_t w_stride, const float * x, int64_t x_stride,
                          float * y, int64_t y_stride, int64_t m, int64_t n, int64_t k) {
    if (!t || t->act < 0) return fail(MI355Q_ERR_UNSUPPORTED, "mul_mat: unsupported weight type");
    if (m < 0 || n < 0 || k <= 0 || k % t->blck) return fail(MI355Q_ERR_SHAPE, "mul_mat: bad shape m=%lld n=%lld k=%lld", (long long) m, (long long) n, (long long) k);
    if (k > (1 << 20)) return fail(MI355Q_ERR_UNSUPPORTED, "mul_mat: k too large");
    const int64_t rb = k / t->blck * t->bsize;
    if (w_stride < rb) return fail(MI355Q_ERR_SHAPE, "mul_mat: w_stride < row size");
    if (x_stride < 4 * k || x_stride % 4 || y_stride < 4 * m || y_stride % 4) return fail(MI355Q_ERR_SHAPE, "mul_mat: bad x/y stride");
    if (((uintptr_t) x & 3) || ((uintptr_t) y & 3)) return fail(MI355Q_ERR_ALIGN, "mul_mat: x/y must be 4-byte aligned");
    if ((m > 0 && n > 0) && (!w || !x || !y)) return fail(MI355Q_ERR_SHAPE, "mul_mat: null pointer");
    return MI355Q_OK;
}

int mi355q_mul_mat_multi(const mi355q_mat * mats, int n_mats, const float * x, int64_t x_stride, int64_t n, int64_t k,
                         void * workspace, size_t workspace_bytes, int flags, void * stream) {
    t_err[0] = 0;
    if (n_mats <= 0) return MI355Q_OK;
    hipStream_t st = (hipStream_t) stream;
    bool all_fast = n_mats <= 4;
    int fam = -1;
    for (int i = 0; i < n_mats; ++i) {
        const TypeInfo * t = type_info(mats[i].type);
        MQ_TRY(mul_mat_checks(t, mats[i].w, mats[i].w_stride, x, x_stride, mats[i].y, mats[i].y_stride, mats[i].m, n, k));
        const bool planar = is_planar(t, k);
        if (planar && (((uintptr_t) mats[i].w | (uintptr_t) mats[i].w_stride) & 15)) return fail(MI355Q_ERR_ALIGN, "mul_mat: planar rows need 16-byte aligned w / w_stride");
        const int f = planar ? gemv_fast_family(mats[i].type) : -1;
        if (f < 0 || (fam >= 0 && f != fam)) all_fast = false;
        if (fam < 0) fam = f;
    }
    if (n == 0) return MI355Q_OK;
    if (all_fast && use_mmq(mats[0].type, n, k, flags)) {
        // MFMA tier (prefill): one launch per matrix; rows that break its alignment contract drop to the GEMV tier
        bool ok = workspace && workspace_bytes >= mmq_workspace(n, k);
        for (int i = 0; i < n_mats && ok; ++i) ok = mmq_supported(mats[i].type, k) && !(((uintptr_t) mats[i].y | (uintptr_t) mats[i].y_stride) & 15);
        if (ok) {
            // Q4_K: integer matrix cores on Q8_K-quantized activations (the CPU backend's arithmetic); the other planar types: bf16 tier
            // Matrices of one call share the prepared activations: the integer image once for all Q4_K matrices, then (the two forms share
            // the scratch buffer, and the stream orders them) the bf16 copy once for the rest.
            static const bool no_i8 = getenv("MI355Q_NO_MMQ_I8") != nullptr;
            const bool i8_ok = !no_i8 && !(((uintptr_t) x | (uintptr_t) x_stride) & 15);
            mi355q_mat i8m[4]; int n_i8 = 0;                 // ... and ONE grid: their row blocks are concatenated
            for (int i = 0; i < n_mats; ++i) if (i8_ok && mmq_i8_supported(mats[i].type, k) && mats[i].m > 0) i8m[n_i8++] = mats[i];
            if (n_i8 > 0) MQ_TRY(launch_mmq_i8_multi(i8m, n_i8, x, x_stride, n, k, workspace, workspace_bytes, cu_count(), st, true));
            // Q8_0: integer matrix cores on Q8_0-quantized activations, the CPU's two f32 operations per block in its block order (mmq_q80.hip):
            // bit-identical to the scalar CPU backend at every batch size
            static const bool no_q80 = getenv("MI355Q_NO_MMQ_Q80") != nullptr;
            const bool q80_ok = !no_q80 && !(((uintptr_t) x | (uintptr_t) x_stride) & 15);
            mi355q_mat q8m[4]; int n_q8 = 0;
            for (int i = 0; i < n_mats; ++i) if (q80_ok && mmq_q80_supported(mats[i].type, k) && mats[i].m > 0) q8m[n_q8++] = mats[i];
            if (n_q8 > 0) MQ_TRY(launch_mmq_q80_multi(q8m, n_q8, x, x_stride, n, k, workspace, workspace_bytes, cu_count(), st, true, (flags & MI355Q_FLAG_ROUND_EVEN) != 0));
            bool first = true;
            for (int i = 0; i < n_mats; ++i)
                if (!(i8_ok && mmq_i8_supported(mats[i].type, k)) && !(q80_ok && mmq_q80_supported(mats[i].type, k))) {
                    MQ_TRY(launch_mmq_bf16(mats[i].type, mats[i].w, mats[i].w_stride, x, x_stride, mats[i].y, mats[i].y_stride, mats[i].m, n, k, workspace, workspace_bytes, cu_count(), st, first));
                    first = false;
                }
            return MI355Q_OK;
        }
    }
    if (all_fast) {
        // GEMV tier, tiled over activation columns (each tile re-streams W; the MFMA tier takes over for large n)
        const int maxc = gemv_fast_max_cols(mats[0].type, k);
        if (maxc < 1) return fail(MI355Q_ERR_UNSUPPORTED, "mul_mat: k=%lld does not fit the LDS activation image", (long long) k);
        mi355q_mat tile[4];
        for (int64_t n0 = 0; n0 < n; n0 += maxc) {
            const int nc = (int) (n - n0 < maxc ? n - n0 : maxc);
            for (int i = 0; i < n_mats; ++i) { tile[i] = mats[i]; tile[i].y = (float *) ((char *) mats[i].y + n0 * mats[i].y_stride); }
            MQ_TRY(launch_gemv_fast(tile, n_mats, (const float *) ((const char *) x + n0 * x_stride), x_stride, nc, k, flags, cu_count(), st, nullptr));
        }
        return MI355Q_OK;
    }
    // mixed / non-planar: one matrix at a time
    for (int i = 0; i < n_mats; ++i) {
        const TypeInfo * t = type_info(mats[i].type);
        if (is_planar(t, k)) {
            MQ_TRY(mi355q_mul_mat_multi(&mats[i], 1, x, x_stride, n, k, workspace, workspace_bytes, flags, stream));
            continue;
        }
        // canonical rows at batch sizes: every weight decoded once per 64 tokens and multiplied on the matrix cores (bf16), instead of one
        // GEMV column per token (MI355Q_FLAG_FORCE_GEMV / _FORCE_GENERIC keep the per-column tier with the CPU's integer arithmetic)
        if (n > 8 && !(flags & (MI355Q_FLAG_FORCE_GEMV | MI355Q_FLAG_FORCE_GENERIC)) && mmq_generic_supported(mats[i].type, k) && !((uintptr_t) x & 3) && !(x_stride & 3)) {
            if (!workspace || workspace_bytes < mmq_generic_workspace(n, k)) return fail(MI355Q_ERR_WORKSPACE, "mul_mat: workspace %zu < %zu", workspace_bytes, mmq_generic_workspace(n, k));
            MQ_TRY(launch_mmq_generic(mats[i].type, mats[i].w, mats[i].w_stride, x, x_stride, mats[i].y, mats[i].y_stride, mats[i].m, n, k, workspace, workspace_bytes, st));
            continue;
        }
        const int64_t arow = mi355q_row_size(t->act, k);
        if (!workspace || workspace_bytes < (size_t) (arow * n)) return fail(MI355Q_ERR_WORKSPACE, "mul_mat: workspace %zu < %lld", workspace_bytes, (long long) (arow * n));
        if (n > 65535) return fail(MI355Q_ERR_UNSUPPORTED, "mul_mat: n > 65535 on the generic tier");
        MQ_TRY(launch_quantize_act(t->act, x, x_stride, workspace, n, k, flags, st));
        MQ_TRY(launch_gemv_generic(mats[i].type, mats[i].w, mats[i].w_stride, workspace, arow, mats[i].y, mats[i].y_stride, mats[i].m, n, k, st, nullptr));
    }
    return MI355Q_OK;
}

int mi355q_mul_mat(int type, const void * w, int64_t w_stride, const float * x, int64_t x_stride, float * y, int64_t y_stride,
                   int64_t m, int64_t n, int64_t k, void * workspace, size_t workspace_bytes, int flags, void * stream) {
    mi355q_mat mt = { type, w, w_stride, y, y_stride, m };
    return mi355q_mul_mat_multi(&mt, 1, x, x_stride, n, k, workspace, workspace_bytes, flags, stream);
}

// ---- MUL_MAT_ID ----
size_t mi355q_mul_mat_id_workspace(int type, int64_t m, int64_t k, int64_t n_used, int64_t n_tok, int64_t x_ne1, int64_t n_expert) {
    const TypeInfo * t = type_info(type);
    if (!t || t->act < 0 || k % t->blck) return 0;
    const size_t grouped = n_used * n_tok >= MOE_GROUPED_MIN_PAIRS && ((is_planar(t, k) && mmq_supported(type, k) && m % 4 == 0) || (!is_planar(t, k) && mmq_generic_supported(type, k))) ? moe_grouped_workspace(type, m, k, n_used * n_tok, n_expert < 1 ? 1 : (n_expert > 1024 ? 1024 : n_expert)) : 0;
    if (is_planar(t, k)) return grouped;
    const size_t generic = (size_t) align256(mi355q_row_size(t->act, k) * n_tok * x_ne1);
    return grouped > generic ? grouped : generic;
}

int mi355q_mul_mat_id(int type, const void * w, int64_t w_stride, int64_t expert_stride, int64_t n_expert,
                      const float * x, int64_t x_ne1, int64_t x_stride1, int64_t x_stride2,
                      const int32_t * ids, int64_t ids_stride, float * y, int64_t m, int64_t k, int64_t n_used, int64_t n_tok,
                      void * workspace, size_t workspace_bytes, int flags, void * stream) {
    t_err[0] = 0;
    const TypeInfo * t = type_info(type);
    MQ_TRY(mul_mat_checks(t, w, w_stride, x, x_stride1, y, 4 * m, m, n_used * n_tok, k));
    if (!ids || n_expert <= 0 || x_ne1 <= 0 || (x_ne1 != 1 && x_ne1 != n_used && n_used % x_ne1)) return fail(MI355Q_ERR_SHAPE, "mul_mat_id: bad ids / x_ne1");
    if (ids_stride % 4 || x_stride2 % 4) return fail(MI355Q_ERR_ALIGN, "mul_mat_id: strides must be multiples of 4");
    const int64_t pairs = n_used * n_tok;
    if (pairs == 0 || m == 0) return MI355Q_OK;
    if (pairs > 65535) return fail(MI355Q_ERR_UNSUPPORTED, "mul_mat_id: more than 65535 (token,slot) pairs per call");
    hipStream_t st = (hipStream_t) stream;
    const bool grouped_canonical = moe_grouped_canonical_ok(t, type, k, x, x_stride1, x_stride2);
    if (pairs >= MOE_GROUPED_MIN_PAIRS && n_expert <= 1024 &&
        (grouped_canonical || (is_planar(t, k) && moe_grouped_ok(type, m, k, x, x_stride1, x_stride2) && !((uintptr_t) y & 15) &&
                               !(((uintptr_t) w | (uintptr_t) w_stride | (uintptr_t) expert_stride) & 15)))) {
        // Prefill-sized batches: rows grouped by expert ON THE DEVICE, one launch of the matrix-core tier over all experts (see k_moe_sort)
        const size_t need = moe_grouped_workspace(type, m, k, pairs, n_expert);
        if (!workspace || workspace_bytes < need) return fail(MI355Q_ERR_WORKSPACE, "mul_mat_id: workspace %zu < %zu for the grouped form", workspace_bytes, need);
        const int tile = moe_tile(type, k, pairs, n_expert);
        const int64_t slots = moe_slots(type, k, pairs, n_expert);
        char * wsp = (char *) workspace;
        int32_t * d_order = (int32_t *) wsp, * d_slot = d_order + slots, * d_seg_end = d_slot + pairs, * d_tile = d_seg_end + n_expert;
        wsp += moe_align256(4 * (slots + pairs + n_expert + slots / 64));
        float * xg = (float *) wsp;                           wsp += moe_align256(4 * slots * k);
        const size_t ws_left = workspace_bytes - (size_t) (wsp - (char *) workspace);
        hipLaunchKernelGGL(k_moe_sort, dim3(1), dim3(1024), (size_t) (2 * n_expert * 4), st, (const char *) ids, ids_stride, (int) n_used, (int) pairs, (int) n_expert, tile, (int) slots,
                           d_order, d_slot, d_seg_end, d_tile);
        hipLaunchKernelGGL(k_moe_gather, dim3((unsigned) slots), dim3(256), 0, st, (const char *) x, x_stride1, x_stride2, (int) n_used, (int) x_ne1, d_order, xg, k);
        const MoeTiles mt = { d_tile, d_seg_end, expert_stride, tile, 0, d_order };
        static const bool no_i8 = getenv("MI355Q_NO_MMQ_I8") != nullptr;
        static const bool no_q80g = getenv("MI355Q_NO_MMQ_Q80") != nullptr;
        if (grouped_canonical) {
            MQ_TRY(launch_mmq_generic(type, w, w_stride, xg, 4 * k, y, 4 * m, m, slots, k, wsp, ws_left, st, &mt));
        } else if (!no_i8 && mmq_i8_supported(type, k)) {
            const mi355q_mat one = { type, w, w_stride, y, 4 * m, m };
            MQ_TRY(launch_mmq_i8_multi(&one, 1, xg, 4 * k, slots, k, wsp, ws_left, cu_count(), st, true, &mt));
        } else if (!no_q80g && mmq_q80_supported(type, k)) {
            const mi355q_mat one = { type, w, w_stride, y, 4 * m, m };
            MQ_TRY(launch_mmq_q80_multi(&one, 1, xg, 4 * k, slots, k, wsp, ws_left, cu_count(), st, true, (flags & MI355Q_FLAG_ROUND_EVEN) != 0, &mt));
        } else {
            MQ_TRY(launch_mmq_bf16(type, w, w_stride, xg, 4 * k, y, 4 * m, m, slots, k, wsp, ws_left, cu_count(), st, true, &mt));
        }
        hipLaunchKernelGGL(k_moe_nan_rows, dim3((unsigned) pairs), dim3(256), 0, st, d_slot, y, m);
        return hipGetLastError() == hipSuccess ? MI355Q_OK : fail(MI355Q_ERR_HIP, "mul_mat_id: grouped launch failed");
    }
    if (is_planar(t, k) && gemv_fast_family(type) >= 0) {
        if (((uintptr_t) w | (uintptr_t) w_stride | (uintptr_t) expert_stride) & 15) return fail(MI355Q_ERR_ALIGN, "mul_mat_id: planar rows need 16-byte alignment");
        mi355q_mat mt = { type, w, w_stride, y, 4 * m, m };
        MoeArgs moe = { ids, ids_stride, expert_stride, x_stride2, (int) n_used, (int) x_ne1, (int) n_expert, (int) pairs };
        MQ_TRY(launch_gemv_fast(&mt, 1, x, x_stride1, 1, k, flags, cu_count(), st, &moe));
        return MI355Q_OK;
    }
    // generic tier: quantize all n_tok*x_ne1 activation rows, then one wave per (pair, row)
    const int64_t arow = mi355q_row_size(t->act, k);
    if (!workspace || workspace_bytes < (size_t) (arow * n_tok * x_ne1)) return fail(MI355Q_ERR_WORKSPACE, "mul_mat_id: workspace too small");
    for (int64_t tk = 0; tk < n_tok; ++tk) {        // token planes may be strided: quantize plane by plane
        MQ_TRY(launch_quantize_act(t->act, (const float *) ((const char *) x + tk * x_stride2), x_stride1,
                                   (char *) workspace + tk * x_ne1 * arow, x_ne1, k, flags, st));
    }
    GenericMoe gm = { ids, ids_stride, expert_stride, (int) n_used, (int) x_ne1, (int) n_expert, 0 };
    MQ_TRY(launch_gemv_generic(type, w, w_stride, workspace, arow, y, 4 * m, m, pairs, k, st, &gm));
    return MI355Q_OK;
}

} // extern "C"
