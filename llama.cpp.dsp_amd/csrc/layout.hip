// layout.hip -- canonical ggml rows <-> planar device rows (mi355q_common.h).  Runs once per weight at
// load time (ggml_backend_buffer_i.set_tensor, ggml-backend-impl.h:41-66) and on get_tensor.
// Reordering is per ROW: row byte size and row stride are unchanged, so row-granular views and
// MoE expert offsets (data + i02*nb02, ggml-cuda.cu:2050) remain valid.
#include "mi355q_common.h"

namespace mi355q {

struct PackDesc {
    int     nplanes;
    int     bsize;
    int     src_off[4];
    int     bytes[4];
    int64_t start[5];      // byte offset of each plane inside a device row
    int64_t row_bytes;
    int     perm_plane;    // plane whose bytes are additionally permuted inside each block (-1: none)
    uint8_t perm[16];      // device byte i of that plane's per-block group = source byte perm[i]
};

// one thread per 16-bit word of the device row image; PACK: canonical -> device, else device -> canonical
template <bool PACK>
__global__ void __launch_bounds__(256)
k_repack(const uint16_t * __restrict__ src, uint16_t * __restrict__ dst, const PackDesc d, int64_t nrows) {
    const int64_t words_per_row = d.row_bytes / 2;
    const int64_t total = words_per_row * nrows;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t row = i / words_per_row;
        const int64_t o   = (i - row * words_per_row) * 2;           // byte offset inside the DEVICE row
        int p = 0;
#pragma unroll
        for (int t = 1; t < 4; ++t) if (t < d.nplanes && o >= d.start[t]) p = t;
        const int64_t rel = o - d.start[p];
        const int64_t blk = rel / d.bytes[p];
        const int64_t off = rel - blk * d.bytes[p];
        const int64_t rbase = row * words_per_row;
        if (p == d.perm_plane) {                                     // two bytes of this word come from different places
            const int64_t c0 = blk * d.bsize + d.src_off[p] + d.perm[off], c1 = blk * d.bsize + d.src_off[p] + d.perm[off + 1];
            if (PACK) {
                const uint8_t * sb = (const uint8_t *) (src + rbase);
                dst[rbase + o / 2] = (uint16_t) (sb[c0] | (sb[c1] << 8));
            } else {
                uint8_t * db = (uint8_t *) (dst + rbase);
                const uint16_t v = src[rbase + o / 2];
                db[c0] = (uint8_t) (v & 0xFF); db[c1] = (uint8_t) (v >> 8);
            }
            continue;
        }
        const int64_t canon = blk * d.bsize + d.src_off[p] + off;    // byte offset inside the CANONICAL row
        if (PACK) dst[rbase + o / 2] = src[rbase + canon / 2];
        else      dst[rbase + canon / 2] = src[rbase + o / 2];
    }
}

static bool make_desc(int type, int64_t k, PackDesc & d) {
    const TypeInfo * ti = type_info(type);
    if (!is_planar(ti, k)) return false;
    const int64_t nb = k / ti->blck;
    d.nplanes = ti->nplanes; d.bsize = ti->bsize; d.row_bytes = nb * ti->bsize;
    int64_t off = 0;
    for (int p = 0; p < 4; ++p) {
        d.src_off[p] = ti->planes[p].src_off; d.bytes[p] = ti->planes[p].bytes > 0 ? ti->planes[p].bytes : 1;
        d.start[p] = off;
        if (p < ti->nplanes) off += (int64_t) ti->planes[p].bytes * nb;
    }
    d.start[4] = off;
    d.perm_plane = -1;
    for (int i = 0; i < 16; ++i) d.perm[i] = (uint8_t) i;
    if (type == MI355Q_TYPE_Q6_K) {
        // gemv_fast.hip: chunk j = 4h + 2cc + p reads its two int8 scales (sub-blocks 8h+2cc+p and +4) as ONE 16-bit load
        d.perm_plane = 2;
        for (int j = 0; j < 8; ++j) {
            const int s0 = 8 * (j >> 2) + 2 * ((j >> 1) & 1) + (j & 1);
            d.perm[2 * j] = (uint8_t) s0; d.perm[2 * j + 1] = (uint8_t) (s0 + 4);
        }
    }
    return true;
}

// canonical (device) -> device layout.  src and dst must not overlap.
int launch_pack(int type, void * dst, const void * src, int64_t nrows, int64_t k, bool pack, hipStream_t stream) {
    const TypeInfo * ti = type_info(type);
    if (!ti || ti->act < 0 || k % ti->blck) return MI355Q_ERR_SHAPE;
    const int64_t row_bytes = k / ti->blck * ti->bsize;
    if (nrows <= 0) return MI355Q_OK;
    PackDesc d;
    if (!make_desc(type, k, d)) {                                     // canonical on the device too
        if (hipMemcpyAsync(dst, src, (size_t) (row_bytes * nrows), hipMemcpyDeviceToDevice, stream) != hipSuccess) return MI355Q_ERR_HIP;
        return MI355Q_OK;
    }
    const int64_t total = row_bytes / 2 * nrows;
    const int grid = (int) ((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    if (pack) hipLaunchKernelGGL((k_repack<true>),  dim3(grid), dim3(256), 0, stream, (const uint16_t *) src, (uint16_t *) dst, d, nrows);
    else      hipLaunchKernelGGL((k_repack<false>), dim3(grid), dim3(256), 0, stream, (const uint16_t *) src, (uint16_t *) dst, d, nrows);
    return hipGetLastError() == hipSuccess ? MI355Q_OK : MI355Q_ERR_HIP;
}

} // namespace mi355q
