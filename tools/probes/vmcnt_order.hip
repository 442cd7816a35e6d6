// Probe: do LDS-DMA loads (global_load_lds_dwordx4) and register loads (global_load_dwordx4) retire through
// vmcnt in ISSUE order?  Each wave issues one COLD register load (fresh HBM line), then four HOT LDS-DMA loads
// (L2-resident), then `s_waitcnt vmcnt(4)` — if retirement is in order the register load must have landed.
// The destination is pre-set to a sentinel and snapshotted right after the wait, all inside one asm block.
//   hipcc --offload-arch=gfx950 -O3 vmcnt_order.hip -o vmcnt_order && ./vmcnt_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>


__global__ void __launch_bounds__(256) probe(const uint32_t* __restrict__ cold, const uint32_t* __restrict__ hot, size_t cold_words,
                                             int iters, unsigned long long* __restrict__ violations, unsigned long long* __restrict__ trials) {
    __shared__ __attribute__((aligned(16))) char lds[4 * 4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long bad = 0, n = 0;
    for (int it = 0; it < iters; ++it) {
        // a line nobody touched before: stride the 4 GiB buffer with a large odd step
        const size_t idx = ((gid * 2654435761ull + (size_t)it * 40503ull * 64) % (cold_words / 4)) * 4;
        const uint32_t* cp = cold + idx;
        const uint32_t* hp = hot + (lane * 4) + wave * 256;
        const uint32_t lds_base = __builtin_amdgcn_readfirstlane(
            (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + wave * 4096));   // wave-uniform LDS offset for M0
        // single block: sentinel -> cold register load -> 4 hot LDS-DMA loads -> counted wait -> snapshot
        uint32_t s0;
        asm volatile(
            "v_mov_b32 %0, 0xdeadbeef\n"
            "s_mov_b32 m0, %3\n"
            "global_load_dword %0, %1, off\n"
            "global_load_lds_dwordx4 %2, off\n"
            "global_load_lds_dwordx4 %2, off offset:1024\n"
            "global_load_lds_dwordx4 %2, off offset:2048\n"
            "global_load_lds_dwordx4 %2, off offset:3072\n"
            "s_waitcnt vmcnt(4)\n"
            "v_mov_b32 %0, %0\n"
            : "=&v"(s0)
            : "v"(cp), "v"(hp), "s"(lds_base)
            : "memory", "m0");
        // s0 now holds whatever was in the destination when the wait was satisfied (the v_mov above is a no-op
        // read; a late return would still overwrite it afterwards, so snapshot through a second register)
        uint32_t snap0;
        asm volatile("v_mov_b32 %0, %1\n s_waitcnt vmcnt(0)\n" : "=v"(snap0) : "v"(s0) : "memory");
        ++n;
        if (snap0 == 0xdeadbeefu && cp[0] != 0xdeadbeefu) ++bad;
    }
    atomicAdd(violations, bad);
    atomicAdd(trials, n);
}

int main() {
    const size_t cold_bytes = 4ull << 30, hot_bytes = 1 << 16;
    uint32_t *cold, *hot;
    unsigned long long *cnt;
    hipMalloc(&cold, cold_bytes);
    hipMalloc(&hot, hot_bytes);
    hipMalloc(&cnt, 16);
    hipMemset(cold, 0x11, cold_bytes);
    hipMemset(hot, 0x22, hot_bytes);
    hipMemset(cnt, 0, 16);
    hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, 0, cold, hot, cold_bytes / 4, 64, cnt, cnt + 1);
    hipDeviceSynchronize();
    unsigned long long h[2];
    hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost);
    printf("register load still pending after `vmcnt(4)` with 4 newer LDS-DMA loads: %llu of %llu trials\n", h[0], h[1]);
    printf("%s\n", h[0] ? "=> LDS-DMA and register loads do NOT retire through vmcnt in issue order"
                        : "=> in-order retirement held in every trial");
    return 0;
}
