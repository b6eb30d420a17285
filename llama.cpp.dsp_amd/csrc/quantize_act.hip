// quantize_act.hip -- standalone activation quantizer: f32 rows -> canonical block_q8_0 / block_q8_1 /
// block_q8_K arrays (ggml-common.h:209-227, 329-334) in device memory, bit-identical to the CPU backend.
// Replaces the reference's per-matmul prologue (CPU: ggml-cpu.c:1328-1363; GPU: quantize_q8_1,
// ggml-cuda/quantize.cu:4-38).  Roofline: HBM/L2 bound, 4 B read + ~1.1 B written per element; tiny.
#include "act_quant.cuh"

namespace mi355q {

// grid: (ceil(k/256) spans, n rows); block: 256 threads = 4 waves, each wave one 256-element span
template <int ACT, bool ROUND_EVEN>
__global__ void __launch_bounds__(256)
k_quantize_act(const float * __restrict__ x, int64_t x_stride, uint8_t * __restrict__ out,
               int64_t k, int vec) {
    const int     lane = lane_id();
    const int     wave = (int) (threadIdx.x >> 6);
    const int64_t span = (int64_t) blockIdx.x * 4 + wave;          // 256-element span index in the row
    const int64_t row  = blockIdx.y;
    if (span * 256 >= k) return;                                     // whole wave exits together
    const float * xr = (const float *) ((const char *) x + row * x_stride);
    const int64_t e0 = span * 256 + 4 * lane;
    const float4  v  = load_x4(xr, e0, k, vec != 0);

    if constexpr (ACT == MI355Q_TYPE_Q8_K) {
        uint32_t q; float d; int bsum;
        q8k_wave(v, q, d, bsum);
        uint8_t * blk = out + (row * (k / 256) + span) * 292;
        *(uint32_t *) (blk + 4 + 4 * lane) = q;
        if ((lane & 3) == 0) *(int16_t *) (blk + 260 + 2 * (lane >> 2)) = (int16_t) bsum;
        if (lane == 0) *(float *) blk = d;
    } else {
        uint32_t q; float d; int sum;
        q80_group8<ROUND_EVEN>(v, q, d, sum);
        const int64_t b = (row * k + e0) / 32;                       // block index over the whole output
        if (e0 < k) {
            if constexpr (ACT == MI355Q_TYPE_Q8_0) {
                uint8_t * blk = out + b * 34;
                *(uint16_t *) (blk + 2 + 4 * (lane & 7))     = (uint16_t) (q & 0xFFFFu);   // 2-byte aligned only
                *(uint16_t *) (blk + 2 + 4 * (lane & 7) + 2) = (uint16_t) (q >> 16);
                if ((lane & 7) == 0) *(uint16_t *) blk = __half_as_ushort(__float2half_rn(d));
            } else {                                                  // Q8_1
                uint8_t * blk = out + b * 36;
                *(uint32_t *) (blk + 4 + 4 * (lane & 7)) = q;
                if ((lane & 7) == 0) {
                    *(uint16_t *) blk       = __half_as_ushort(__float2half_rn(d));
                    *(uint16_t *) (blk + 2) = __half_as_ushort(__float2half_rn(__fmul_rn((float) sum, d)));
                }
            }
        }
    }
}

int launch_quantize_act(int act_type, const float * x, int64_t x_stride, void * out,
                        int64_t n, int64_t k, int flags, hipStream_t stream) {
    if (n <= 0 || k <= 0) return MI355Q_OK;
    const bool even = (flags & MI355Q_FLAG_ROUND_EVEN) != 0;
    const int  vec  = (((uintptr_t) x | (uintptr_t) x_stride) & 15) == 0 ? 1 : 0;
    const dim3 grid((unsigned) ((k + 1023) / 1024), (unsigned) n), block(256);
    uint8_t * o = (uint8_t *) out;
    switch (act_type) {
    case MI355Q_TYPE_Q8_K:
        if (k % 256) return MI355Q_ERR_SHAPE;
        hipLaunchKernelGGL((k_quantize_act<MI355Q_TYPE_Q8_K, false>), grid, block, 0, stream, x, x_stride, o, k, vec);
        break;
    case MI355Q_TYPE_Q8_0:
        if (k % 32) return MI355Q_ERR_SHAPE;
        if (even) hipLaunchKernelGGL((k_quantize_act<MI355Q_TYPE_Q8_0, true>),  grid, block, 0, stream, x, x_stride, o, k, vec);
        else      hipLaunchKernelGGL((k_quantize_act<MI355Q_TYPE_Q8_0, false>), grid, block, 0, stream, x, x_stride, o, k, vec);
        break;
    case MI355Q_TYPE_Q8_1:
        if (k % 32) return MI355Q_ERR_SHAPE;
        if (even) hipLaunchKernelGGL((k_quantize_act<MI355Q_TYPE_Q8_1, true>),  grid, block, 0, stream, x, x_stride, o, k, vec);
        else      hipLaunchKernelGGL((k_quantize_act<MI355Q_TYPE_Q8_1, false>), grid, block, 0, stream, x, x_stride, o, k, vec);
        break;
    default:
        return MI355Q_ERR_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? MI355Q_OK : MI355Q_ERR_HIP;
}

} // namespace mi355q
