// mfma_i8_rate.hip -- cycles per MFMA on one SIMD for the integer forms the quantized mat-mat tier can use (gfx950).
// 4 independent accumulator chains per wave, one wave per SIMD, s_memtime (100 MHz) around 4000 issues + wall clock of a full-chip launch.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_i8_rate.hip -o tools/micro/mfma_i8_rate && tools/micro/mfma_i8_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int   v4i  __attribute__((ext_vector_type(4)));
typedef int   v16i __attribute__((ext_vector_type(16)));
typedef float v4f  __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
constexpr int ITERS = 2000;

template <int MODE> __global__ void __launch_bounds__(256) k(int * out, int seed) {
    const int l = threadIdx.x;
    long a8 = seed * 0x0101010101010101L + l, b8 = a8 ^ 0x55;
    v4i a16 = { seed + l, seed, l, 3 }, b16 = { l, 7, seed, 1 };
    bf8 ah, bh; for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(float)(l + i); bh[i] = (__bf16)(float)(seed + i); }
    v4i c0 = {}, c1 = {}, c2 = {}, c3 = {}; v4f f0 = {}, f1 = {}, f2 = {}, f3 = {}; v16i w0 = {}, w1 = {};
    int acc = 0;
    for (int i = 0; i < ITERS; ++i) {
        if constexpr (MODE == 0) { f0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, f0, 0, 0, 0); f1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, f1, 0, 0, 0);
                                   f2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, f2, 0, 0, 0); f3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, f3, 0, 0, 0); }
        if constexpr (MODE == 1) { c0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, c1, 0, 0, 0);
                                   c2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, c3, 0, 0, 0); }
        if constexpr (MODE == 2) { c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, b16, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, b16, c1, 0, 0, 0);
                                   c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, b16, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, b16, c3, 0, 0, 0); }
        if constexpr (MODE == 3) { w0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, w0, 0, 0, 0); w1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, w1, 0, 0, 0);
                                   w0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(b8, a8, w0, 0, 0, 0); w1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(b8, a8, w1, 0, 0, 0); }
        if constexpr (MODE == 4) { w0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a16, b16, w0, 0, 0, 0); w1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a16, b16, w1, 0, 0, 0);
                                   w0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b16, a16, w0, 0, 0, 0); w1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b16, a16, w1, 0, 0, 0); }
        if constexpr (MODE == 5) {     // the K=32 form with the per-group integer scaling the K-quant formats need: fresh accumulator + 4 v_mad per MFMA
            v4i z = {};
            v4i d0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, z, 0, 0, 0), d1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(b8, a8, z, 0, 0, 0);
            v4i d2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, a8, z, 0, 0, 0), d3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(b8, b8, z, 0, 0, 0);
            const int s = (i & 63) + 1;
            for (int r = 0; r < 4; ++r) { c0[r] += d0[r] * s; c1[r] += d1[r] * s; c2[r] += d2[r] * s; c3[r] += d3[r] * s; }
        }
    }
    for (int r = 0; r < 4; ++r) acc += c0[r] + c1[r] + c2[r] + c3[r] + (int) (f0[r] + f1[r] + f2[r] + f3[r]);
    for (int r = 0; r < 16; ++r) acc += w0[r] + w1[r];
    out[blockIdx.x * 256 + l] = acc;
}

template <int MODE> static void run(const char * name, double ops_per_mfma) {
    int * out; hipMalloc(&out, 256 * 4096 * sizeof(int));
    const int grid = 256 * 8;                       // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    k<MODE><<<grid, 256>>>(out, 3); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<MODE><<<grid, 256>>>(out, 3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double) grid * 4 * ITERS * 4;
    printf("%-44s %8.1f us  %7.1f T(FL)OP/s  (%.2f ns per MFMA per SIMD)\n", name, ms * 1e3, n_mfma * ops_per_mfma / (ms * 1e-3) / 1e12,
           ms * 1e6 / (n_mfma / (256.0 * 4)));
    hipFree(out);
}
int main() {
    run<0>("f32_16x16x32_bf16", 2.0 * 16 * 16 * 32);
    run<1>("i32_16x16x32_i8", 2.0 * 16 * 16 * 32);
    run<2>("i32_16x16x64_i8", 2.0 * 16 * 16 * 64);
    run<3>("i32_32x32x16_i8", 2.0 * 32 * 32 * 16);
    run<4>("i32_32x32x32_i8", 2.0 * 32 * 32 * 32);
    run<5>("i32_16x16x32_i8 + 4 v_mad (group scales)", 2.0 * 16 * 16 * 32);
    return 0;
}
