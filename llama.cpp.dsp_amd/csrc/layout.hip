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
        const int64_t canon = blk * d.bsize + d.src_off[p] + off;    // byte offset inside the CANONICAL row
        const int64_t rbase = row * words_per_row;
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
