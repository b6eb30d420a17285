// mfma_tile_sched.hip -- how the per-tile MFMA/VALU dependency pattern of mmq_i8 issues at 1/2 waves per SIMD (dev tool).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_tile_sched.hip -o tools/micro/mfma_tile_sched
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int   v4i  __attribute__((ext_vector_type(4)));
typedef float v4f  __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
constexpr int ITERS = 1000;
// MODE 0: one tile at a time (2 chains of 4 + bf16 + dependent VALU).  MODE 1: two tiles interleaved (4 chains), VALU of both after.
// MODE 2: as 0 but the VALU of tile t is issued after the MFMAs of tile t+1 (software pipelined by hand).
template <int MODE> __global__ void __launch_bounds__(256, 2) k(float * out, int seed) {
    const int l = threadIdx.x;
    v4i a[4], pl[4], ph[4];
    for (int s = 0; s < 4; ++s) { a[s] = (v4i) { seed + l, s, l, 3 }; pl[s] = (v4i) { l, 7 + s, seed, 1 }; ph[s] = (v4i) { l ^ s, 5, seed, 2 }; }
    bf8 sf, mf; for (int i = 0; i < 8; ++i) { sf[i] = (__bf16)(float)(l + i); mf[i] = (__bf16)(float)(seed + i); }
    v4f f0 = {}, f1 = {};
    const float d = 0.5f + seed, dm = 0.25f;
    v4f dx = { 1.f, 2.f, 3.f, 4.f };
    auto fin = [&](v4f & f, const v4i & il, const v4i & ih, const v4f & ms) {
        for (int r = 0; r < 4; ++r) f[r] = __builtin_fmaf(dx[r], __builtin_fmaf(d, (float) ((ih[r] << 3) + il[r]), -(dm * ms[r])), f[r]);
    };
    const v4f zf = { 0.f, 0.f, 0.f, 0.f }; const v4i zi = { 0, 0, 0, 0 };
    v4i pil = zi, pih = zi; v4f pms = zf;
    for (int i = 0; i < ITERS; ++i) {
        if constexpr (MODE == 0) {
            for (int t = 0; t < 2; ++t) {
                v4i il = zi, ih = zi;
                for (int s = 0; s < 4; ++s) { il = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], pl[s], il, 0, 0, 0); ih = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], ph[s], ih, 0, 0, 0); }
                v4f ms = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf, mf, zf, 0, 0, 0);
                fin(t ? f1 : f0, il, ih, ms);
                a[0][0] += 1;
            }
        } else if constexpr (MODE == 1) {
            v4i il0 = zi, ih0 = zi, il1 = zi, ih1 = zi;
            for (int s = 0; s < 4; ++s) {
                il0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], pl[s], il0, 0, 0, 0); ih0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], ph[s], ih0, 0, 0, 0);
                il1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], ph[s], il1, 0, 0, 0); ih1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], pl[s], ih1, 0, 0, 0);
            }
            v4f ms0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf, mf, zf, 0, 0, 0), ms1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf, sf, zf, 0, 0, 0);
            fin(f0, il0, ih0, ms0); fin(f1, il1, ih1, ms1);
            a[0][0] += 1;
        } else {
            for (int t = 0; t < 2; ++t) {
                v4i il = zi, ih = zi;
                for (int s = 0; s < 4; ++s) { il = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], pl[s], il, 0, 0, 0); ih = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], ph[s], ih, 0, 0, 0);
                    if (s == 1) fin(t ? f0 : f1, pil, pih, pms); }       // the previous tile's update rides between this tile's MFMAs
                v4f ms = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf, mf, zf, 0, 0, 0);
                pil = il; pih = ih; pms = ms;
                a[0][0] += 1;
            }
        }
    }
    fin(f0, pil, pih, pms);
    out[blockIdx.x * 256 + l] = f0[0] + f0[1] + f0[2] + f0[3] + f1[0] + f1[1] + f1[2] + f1[3];
}
template <int MODE> static void run(const char * name, int wg_per_cu) {
    float * out; (void) hipMalloc(&out, 256 * 4096 * sizeof(float));
    const int grid = 256 * wg_per_cu;
    k<MODE><<<grid, 256>>>(out, 3); (void) hipDeviceSynchronize();
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    (void) hipEventRecord(e0); k<MODE><<<grid, 256>>>(out, 3); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double) wg_per_cu * ITERS * 18;     // per SIMD
    printf("%-60s %d wave/SIMD: %8.1f us  %.2f ns per MFMA per SIMD\n", name, wg_per_cu, ms * 1e3, ms * 1e6 / n_mfma);
    (void) hipFree(out);
}
int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0>("tile by tile (2 chains x 4, bf16, dependent update)", w);
        run<1>("two tiles interleaved (4 chains)", w);
        run<2>("update of tile t-1 between the MFMAs of tile t", w);
    }
    return 0;
}
