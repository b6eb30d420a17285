// Micro-benchmark (dev tool): the decode engine's data path in isolation -- one LOADER wave per workgroup streams the workgroup's
// slice of a buffer into an LDS ring with global_load_lds_dwordx4 (1 KiB pages, saddr form, optional nt), publishes how many pages
// have landed in an LDS word; 15 CONSUMER waves take rows of `rb` bytes round-robin, wait for their row to have landed, read it
// with ds_read_b128, do WORK extra vector instructions per 16 bytes, and publish per-row checksums; each consumer keeps an LDS
// "head" word (first page it still needs) from which the loader derives the free space.  No hardware barrier after the start.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ring_stream.hip -o tools/micro/ring_stream && tools/micro/ring_stream
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NW = 16;
enum { C_LANDED = 0, C_ABORT = 1, C_LANDED2 = 2, C_HEAD = 16 };       // u32 indices into the control block (256 bytes)

// control words live in LDS and are touched with explicit DS instructions on their LDS byte address: a `volatile` generic pointer
// compiles to flat loads / stores with sc0 sc1 and a vmcnt(0) wait behind each -- which drains the loader's DMA queue on every publish
__device__ __forceinline__ void lds_st(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned lds_ld(unsigned addr) { unsigned v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory"); return v; }
__device__ __forceinline__ unsigned lds_ld_u(unsigned addr) { return (unsigned) __builtin_amdgcn_readfirstlane((int) lds_ld(addr)); }   // wave-uniform result

template <int D, bool NT>
__device__ __forceinline__ void dma_page(const uint8_t * gbase_in, unsigned voff, unsigned lds_addr_in) {
    const unsigned long long gv = (unsigned long long) (uintptr_t) gbase_in;       // wave-uniform by construction: moved to SGPRs for the saddr form
    const unsigned glo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) gv), ghi = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (gv >> 32));
    const uint8_t * gbase = (const uint8_t *) (uintptr_t) (((unsigned long long) ghi << 32) | glo);
    const unsigned lds_addr = (unsigned) __builtin_amdgcn_readfirstlane((int) lds_addr_in);
    if constexpr (NT) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" :: "v"(voff), "s"(gbase), "s"(lds_addr) : "memory");
    else              asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(gbase), "s"(lds_addr) : "memory");
}

// four consecutive pages with ONE M0 write: the instruction offset advances the global and the LDS address alike
template <bool NT>
__device__ __forceinline__ void dma_group4(const uint8_t * gbase_in, unsigned voff, unsigned lds_addr_in) {
    const unsigned long long gv = (unsigned long long) (uintptr_t) gbase_in;
    const unsigned glo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) gv), ghi = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (gv >> 32));
    const uint8_t * gbase = (const uint8_t *) (uintptr_t) (((unsigned long long) ghi << 32) | glo);
    const unsigned lds_addr = (unsigned) __builtin_amdgcn_readfirstlane((int) lds_addr_in);
    if constexpr (NT) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072 nt" :: "v"(voff), "s"(gbase), "s"(lds_addr) : "memory");
    else              asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(voff), "s"(gbase), "s"(lds_addr) : "memory");
}

template <int D, bool NT, int WORK, int NL>
__global__ void __launch_bounds__(1024) k_ring(const uint8_t * buf, unsigned bytes_per_wg, int rb, int np, unsigned * sums, int rows_per_wg, unsigned * fail) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const unsigned ctl = (unsigned) (size_t) lds;           // LDS byte address of the control block
    const unsigned ring_lds = (unsigned) (size_t) (lds + 256);
    const unsigned ring_bytes = (unsigned) np * 1024u;
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)), lane = (int) (threadIdx.x & 63);
    if (threadIdx.x < 64) ((unsigned *) lds)[threadIdx.x] = threadIdx.x >= C_HEAD && threadIdx.x < C_HEAD + (NW - NL) ? ((threadIdx.x - C_HEAD) * (unsigned) rb) >> 10 : (threadIdx.x >= C_HEAD + (NW - NL) && threadIdx.x < C_HEAD + NW ? 0xFFFFFFFFu : 0u);
    __syncthreads();
    const uint8_t * src = buf + (size_t) blockIdx.x * bytes_per_wg;
    const unsigned total_pages = bytes_per_wg >> 10;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    constexpr int NCc = NW - NL;
    if (wave >= NCc) {
        const unsigned me = (unsigned) (wave - NCc);          // loader index: takes the 4-page groups g with g % NL == me
        __builtin_amdgcn_s_setprio(3);
        // `issued` counts pages of the whole stream up to which THIS loader has issued all of ITS groups; total_pages % 4 == 0, np % 4 == 0
        unsigned issued = 4u * me, tail = 0, landed = 0;
        const unsigned voff = 16u * (unsigned) lane;
        const unsigned my_landed = ctl + 4 * (me ? C_LANDED2 : C_LANDED);
        unsigned spins = 0, mine_out = 0;                          // mine_out: my groups issued so far
        while (issued < total_pages) {
            issued = (unsigned) __builtin_amdgcn_readfirstlane((int) issued); tail = (unsigned) __builtin_amdgcn_readfirstlane((int) tail);
            landed = (unsigned) __builtin_amdgcn_readfirstlane((int) landed); mine_out = (unsigned) __builtin_amdgcn_readfirstlane((int) mine_out);
            {
                unsigned h = lds_ld(ctl + 4 * (C_HEAD + (lane & 15)));
                h = min(h, (unsigned) __builtin_amdgcn_update_dpp(0, (int) h, 0xB1, 0xF, 0xF, true)); h = min(h, (unsigned) __builtin_amdgcn_update_dpp(0, (int) h, 0x4E, 0xF, 0xF, true));
                h = min(h, (unsigned) __builtin_amdgcn_update_dpp(0, (int) h, 0x141, 0xF, 0xF, true)); h = min(h, (unsigned) __builtin_amdgcn_update_dpp(0, (int) h, 0x140, 0xF, 0xF, true));
                tail = (unsigned) __builtin_amdgcn_readfirstlane((int) h);
            }
            // groups I may issue now: group at page `issued` needs issued + 4 - tail <= np
            unsigned n = 0;
            { unsigned p = issued; while (n < 2 && p < total_pages && (int) (p + 4 - tail) <= np) { ++n; p += 4u * NL; } }
            n = (unsigned) __builtin_amdgcn_readfirstlane((int) n);
            if (n == 0) {
                if (landed != issued) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); landed = issued; if (lane == 0) lds_st(my_landed, landed); }
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 4095u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > 100ull * 1000 * 500) { if (lane == 0) { lds_st(ctl + 4 * C_ABORT, 1u); atomicAdd(fail, 1u); } break; }
                continue;
            }
            for (unsigned i = 0; i < n; ++i) {
                const unsigned pg = issued % (unsigned) np;
                dma_group4<NT>(src + (size_t) issued * 1024u, voff, ring_lds + pg * 1024u);
                issued += 4u * NL;
            }
            mine_out += n;
            asm volatile("s_waitcnt vmcnt(%0)" :: "i"(D) : "memory");
            // at most D of my page loads are outstanding = at most D / 4 (rounded up) of my groups: every earlier group of mine has landed.
            // my frontier: the first page of my oldest group that may still be in flight
            { const unsigned safe = mine_out > (unsigned) ((D + 3) / 4) ? mine_out - (unsigned) ((D + 3) / 4) : 0u;      // my groups known landed
              const unsigned fr = safe * 4u * NL + 4u * me;          // = first page of my group number `safe`
              if (fr > landed) { landed = fr; if (lane == 0) lds_st(my_landed, landed); } }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) lds_st(my_landed, 0x7FFFFFFFu);
    } else if (WORK < 0) {                                       // the loader alone: consumers leave, the ring is never full
        if (lane == 0) lds_st(ctl + 4 * (C_HEAD + wave), 0x7FFFFFFFu);
    } else {
        constexpr int NC = NCc;
        unsigned ring_off = (unsigned) wave * (unsigned) rb;   // (row * rb) % ring_bytes
        while (ring_off >= ring_bytes) ring_off -= ring_bytes;
        for (int row = wave; row < rows_per_wg; row += NC) {
            const unsigned off = (unsigned) row * (unsigned) rb;
            const unsigned end_page = (off + (unsigned) rb + 1023u) >> 10;
            for (unsigned spins = 0; (int) (lds_ld_u(ctl + 4 * C_LANDED) - end_page) < 0 || (NL > 1 && (int) (lds_ld_u(ctl + 4 * C_LANDED2) - end_page) < 0);) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 1023u) == 0 && (lds_ld_u(ctl + 4 * C_ABORT) || __builtin_amdgcn_s_memrealtime() - t0 > 100ull * 1000 * 500)) { if (lane == 0) { lds_st(ctl + 4 * C_ABORT, 1u); atomicAdd(fail, 1u); } return; }
            }
            unsigned acc = 0;
            const bool wraps = ring_off + (unsigned) rb > ring_bytes;
            for (unsigned s = 0; s * 1024u < (unsigned) rb; ++s) {
                const unsigned o = s * 1024u + 16u * (unsigned) lane;
                if (o < (unsigned) rb) {
                    unsigned a = ring_off + o;
                    if (wraps && a >= ring_bytes) a -= ring_bytes;
                    const uint4 v = *(const uint4 *) (lds + 256 + a);
                    const unsigned base = v.x ^ v.y ^ v.z ^ v.w;
                    unsigned x = base, y = v.z;
#pragma unroll
                    for (int w = 0; w < WORK / 6; ++w) { x = ((x >> 1) ^ x) + v.y; y = ((y >> 3) ^ y) + v.w; }      // stand-in for the dequant + dot work: 6 full-rate VALU ops, two chains
                    if (WORK) asm volatile("" :: "v"(x), "v"(y));
                    acc ^= base;
                }
            }
            for (int o = 32; o > 0; o >>= 1) acc ^= (unsigned) __shfl_xor((int) acc, o);
            if (lane == 0) sums[(size_t) blockIdx.x * rows_per_wg + row] = acc;
            const int next = row + NC;
            if (lane == 0) lds_st(ctl + 4 * (C_HEAD + wave), next < rows_per_wg ? ((unsigned) next * (unsigned) rb) >> 10 : total_pages);
            ring_off += (unsigned) NC * (unsigned) rb;
            while (ring_off >= ring_bytes) ring_off -= ring_bytes;
        }
    }
}

// reference: plain streaming read with 16 waves, 8 x 16-byte nt loads in flight per lane (the usual "measured HBM read peak" kernel)
__global__ void __launch_bounds__(1024) k_plain(const uint4 * buf, size_t n16_per_wg, unsigned * sink) {
    const uint4 * p = buf + (size_t) blockIdx.x * n16_per_wg;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i + 7 * 1024 < n16_per_wg; i += 8 * 1024) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const uint4 * q = p + i + u * 1024; v[u].x = __builtin_nontemporal_load(&q->x); v[u].y = __builtin_nontemporal_load(&q->y); v[u].z = __builtin_nontemporal_load(&q->z); v[u].w = __builtin_nontemporal_load(&q->w); }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int D, bool NT, int WORK, int NL>
static void run(const char * name, const uint8_t * d_buf, unsigned bytes_per_wg, int rb, int np, unsigned * d_sums, unsigned * d_fail, const std::vector<unsigned> & ref, int grid) {
    const int rows = (int) (bytes_per_wg / (unsigned) rb);
    const size_t lds = 256 + (size_t) np * 1024;
    hipFuncSetAttribute((const void *) k_ring<D, NT, WORK, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    hipMemset(d_sums, 0, (size_t) grid * rows * 4); hipMemset(d_fail, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_ring<D, NT, WORK, NL>), dim3(grid), dim3(1024), lds, 0, d_buf, bytes_per_wg, rb, np, d_sums, rows, d_fail);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    std::vector<unsigned> got((size_t) grid * rows); unsigned fail = 0;
    hipMemcpy(got.data(), d_sums, got.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(&fail, d_fail, 4, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
    printf("%-34s rb %5d np %3d : %8.3f ms  %7.1f GB/s  %s (bad rows %zu, aborts %u)\n", name, rb, np, best, (double) grid * bytes_per_wg / best * 1e-6, WORK >= 0 && (bad || fail) ? "WRONG" : "ok", bad, fail);
    fflush(stdout);
}

int main() {
    const int grid = 256;
    const unsigned rbs[] = { 2304 };
    for (unsigned rb : rbs) {
        // bytes per workgroup: a multiple of rb and of 1024, ~8 MB
        unsigned rows = (8u << 20) / rb; while ((rows * rb) % 4096u) --rows;
        const unsigned bpw = rows * rb;
        const size_t total = (size_t) grid * bpw;
        std::vector<uint8_t> h(total);
        uint64_t s = 88172645463325252ull;
        for (size_t i = 0; i < total; i += 8) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; *(uint64_t *) (h.data() + i) = s; }
        std::vector<unsigned> ref((size_t) grid * rows);
        for (size_t r = 0; r < ref.size(); ++r) { unsigned a = 0; const unsigned * p = (const unsigned *) (h.data() + r * rb); for (unsigned i = 0; i < rb / 4; ++i) a ^= p[i]; ref[r] = a; }
        uint8_t * d_buf; unsigned * d_sums, * d_fail;
        hipMalloc(&d_buf, total); hipMalloc(&d_sums, ref.size() * 4); hipMalloc(&d_fail, 4);
        hipMemcpy(d_buf, h.data(), total, hipMemcpyHostToDevice);
        {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float best = 1e30f;
            for (int it = 0; it < 4; ++it) { hipEventRecord(e0); hipLaunchKernelGGL(k_plain, dim3(grid), dim3(1024), 0, 0, (const uint4 *) d_buf, (size_t) bpw / 16, d_fail); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
            printf("plain 16-wave nt stream (%.0f MB)               : %8.3f ms  %7.1f GB/s\n", total * 1e-6, best, total / best * 1e-6);
        }
        run<32, true, 0, 1>("1 loader  D32 nt work0", d_buf, bpw, rb, 120, d_sums, d_fail, ref, grid);
        run<32, true, -1, 1>("1 loader  D32 nt ALONE", d_buf, bpw, rb, 120, d_sums, d_fail, ref, grid);
        run<56, true, -1, 1>("1 loader  D56 nt ALONE", d_buf, bpw, rb, 120, d_sums, d_fail, ref, grid);
        run<32, true, 48, 1>("1 loader  D32 nt work48", d_buf, bpw, rb, 120, d_sums, d_fail, ref, grid);
        hipFree(d_buf); hipFree(d_sums); hipFree(d_fail);
    }
    return 0;
}
