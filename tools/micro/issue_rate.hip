// Dev micro-benchmark (GPU box): what one wave pays per instruction when it runs (nearly) alone on its SIMD -- the regime of the decode plan's
// attention stage, where 16 waves of one workgroup execute short serial phases.  Prints ns per instruction for: a dependent f32 fma chain, four
// independent fma chains, a dependent DPP add chain (the wave reduction), dependent LDS read -> use, s_barrier round trips, expf / cosf / sinf.
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip ; run: ./issue_rate [waves per workgroup = 16] [workgroups = 32]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define N_IT 512

__global__ void k(float * out, unsigned long long * t, int mode, float seed) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float) ((i * 7 + 1) & 1023);
    __syncthreads();
    float a = seed + lane, b = 1.0001f, c = 0.5f, a1 = a + 1, a2 = a + 2, a3 = a + 3;
    int idx = lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (mode == 0) {
#pragma unroll 16
        for (int i = 0; i < N_IT; ++i) a = __builtin_fmaf(a, b, c);
    } else if (mode == 1) {
#pragma unroll 4
        for (int i = 0; i < N_IT / 4; ++i) { a = __builtin_fmaf(a, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c); }
        a += a1 + a2 + a3;
    } else if (mode == 2) {
#pragma unroll 16
        for (int i = 0; i < N_IT; ++i) a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true));
    } else if (mode == 3) {
#pragma unroll 8
        for (int i = 0; i < N_IT; ++i) idx = (int) lds[idx & 4095] + lane;
        a = (float) idx;
    } else if (mode == 4) {
        for (int i = 0; i < N_IT; ++i) { __builtin_amdgcn_s_barrier(); }
    } else if (mode == 5) {
        for (int i = 0; i < N_IT / 8; ++i) a = expf(a * 1e-3f) + c;
    } else if (mode == 6) {
        for (int i = 0; i < N_IT / 8; ++i) a = cosf(a * 3.0f + 100.0f) + sinf(a + 50.0f);
    } else if (mode == 7) {
        for (int i = 0; i < N_IT / 8; ++i) a = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a)) * 1.0f + lane;   // VALU -> SALU -> VALU round trip
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) t[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

int main(int argc, char ** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 16, wgs = argc > 2 ? atoi(argv[2]) : 32;
    float * out; unsigned long long * t;
    hipMalloc(&out, sizeof(float) * wgs * waves * 64); hipMalloc(&t, 8 * wgs * waves);
    const char * names[] = { "dependent fma", "4 independent fma chains", "dependent DPP add", "dependent LDS read+cvt+add", "s_barrier", "expf (+mul, add)", "cosf + sinf", "readfirstlane round trip" };
    const int per[] = { N_IT, N_IT, N_IT, N_IT, N_IT, N_IT / 8, N_IT / 8, N_IT / 8 };
    for (int mode = 0; mode < 8; ++mode) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(wgs), dim3(64 * waves), 0, 0, out, t, mode, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(wgs * waves);
        hipMemcpy(h.data(), t, 8 * wgs * waves, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double) v;
        printf("%-32s %7.2f ns per iteration (%d waves/WG, %d WGs)\n", names[mode], s / h.size() * 10.0 / per[mode], waves, wgs);
    }
    return 0;
}
