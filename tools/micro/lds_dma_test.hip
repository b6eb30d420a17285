// Micro-test (dev tool): semantics of global_load_lds_dwordx4 on gfx950 -- where does lane i's 16 bytes land?
//   hipcc --offload-arch=gfx950 -O2 tools/micro/lds_dma_test.hip -o /tmp/lds_dma_test && /tmp/lds_dma_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned * g, unsigned * out, int lanes_active) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096 / 4; i += blockDim.x) ((unsigned *) lds)[i] = 0xDEADBEEFu;
    __syncthreads();
    const unsigned lds_base = (unsigned) (size_t) (lds + 256);            // LDS byte address (low 32 bits of the shared pointer)
    const unsigned char * p = (const unsigned char *) g + 16 * lane;
    if (lane < lanes_active) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\ts_waitcnt vmcnt(0)" :: "v"(p), "s"(lds_base) : "memory");
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4096 / 4; i += blockDim.x) out[i] = ((unsigned *) lds)[i];
}
int main() {
    std::vector<unsigned> h(1024), o(1024);
    for (int i = 0; i < 1024; ++i) h[i] = i;
    unsigned * dg, * dout; hipMalloc(&dg, 4096); hipMalloc(&dout, 4096);
    hipMemcpy(dg, h.data(), 4096, hipMemcpyHostToDevice);
    for (int la : {64, 8}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, dg, dout, la);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
        int first = -1, last = -1, bad = 0;
        for (int i = 0; i < 1024; ++i) if (o[i] != 0xDEADBEEFu) { if (first < 0) first = i; last = i; }
        printf("lanes_active=%d: written dwords [%d..%d]\n", la, first, last);
        for (int i = first; i <= last && i >= 0; ++i) if (o[i] != (unsigned) (i - 64)) ++bad;     // expect lds[256 + 16*lane + j] = g[16*lane + j]
        printf("  contiguous-by-lane layout at base+256: %s (mismatches %d); sample o[64..71] = %u %u %u %u %u %u %u %u\n", bad ? "NO" : "YES", bad,
               o[64], o[65], o[66], o[67], o[68], o[69], o[70], o[71]);
    }
    return 0;
}
