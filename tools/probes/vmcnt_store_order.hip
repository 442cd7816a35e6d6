// Probe (round 5): does vmcnt retire LOADS and STORES in one issue-ordered queue on gfx950?
//   test 1 (old load, young stores): one COLD register load, then four stores to an L2-hot line, then `s_waitcnt vmcnt(4)`.  If stores could
//           retire ahead of the older load the count would drop to 1 <= 4 with the load still in flight (sentinel seen).
//   test 2 (old stores, young load — the case the rolling GEMM epilogue needs): NS stores of 1 KB per wave to COLD lines (a burst the
//           memory system drains slowly), then one L2-hot load, then `s_waitcnt vmcnt(0)`; and the other order — hot load FIRST, then the NS
//           stores, then `s_waitcnt vmcnt(NS)`.  Times of both by s_memtime: in-order retirement means the second form returns as soon as
//           the load has landed, without waiting for the burst.
//   hipcc --offload-arch=gfx950 -O3 vmcnt_store_order.hip -o vmcnt_store_order && ./vmcnt_store_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__global__ void __launch_bounds__(256) probe1(const uint32_t* __restrict__ cold, uint32_t* __restrict__ hot, size_t cold_words, int iters,
                                              unsigned long long* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long bad = 0, n = 0;
    for (int it = 0; it < iters; ++it) {
        const size_t idx = ((gid * 2654435761ull + (size_t)it * 40503ull * 64) % (cold_words / 4)) * 4;
        const uint32_t* cp = cold + idx;
        uint32_t* hp = hot + ((size_t)blockIdx.x * 4 + wave) * 256 + lane * 4;
        uint32_t s0;
        asm volatile(
            "v_mov_b32 %0, 0xdeadbeef\n"
            "global_load_dword %0, %1, off\n"
            "global_store_dword %2, %3, off\n"
            "global_store_dword %2, %3, off offset:4\n"
            "global_store_dword %2, %3, off offset:8\n"
            "global_store_dword %2, %3, off offset:12\n"
            "s_waitcnt vmcnt(4)\n"
            "v_mov_b32 %0, %0\n"
            : "=&v"(s0)
            : "v"(cp), "v"(hp), "v"(lane)
            : "memory");
        uint32_t snap0;
        asm volatile("v_mov_b32 %0, %1\n s_waitcnt vmcnt(0)\n" : "=v"(snap0) : "v"(s0) : "memory");
        ++n;
        if (snap0 == 0xdeadbeefu && cp[0] != 0xdeadbeefu) ++bad;
    }
    atomicAdd(out, bad);
    atomicAdd(out + 1, n);
}

// every workgroup of the chip bursts NS x 1 KB stores per wave at once (256 WGs x 4 waves x 32 KB = 32 MB), with one hot load before or after
template <int ORDER>
__global__ void __launch_bounds__(256) probe2(uint32_t* __restrict__ big, const uint32_t* __restrict__ hot, unsigned long long* __restrict__ out,
                                              int rounds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(big, 0, 0x7fffffffu, 0x00020000);
    const uint32_t* hp = hot + threadIdx.x;
    unsigned long long tsum = 0;
    for (int r = 0; r < rounds; ++r) {
        const uint32_t base = (uint32_t)((((size_t)r * gridDim.x + blockIdx.x) * 4 + wave) * 32768 + lane * 16);
        const u32x4 v = {(uint32_t)r, (uint32_t)lane, 0u, 0u};
        __syncthreads();
        uint32_t x;
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (ORDER == 0) {                                    // stores first, then the load, wait for everything
#pragma unroll
            for (int s = 0; s < 32; ++s) __builtin_amdgcn_raw_buffer_store_b128(v, rs, base + s * 1024, 0, 0);
            asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(x) : "v"(hp) : "memory");
        } else {                                             // the load first, then the stores, wait for all but the 32 youngest
            asm volatile("global_load_dword %0, %1, off" : "=v"(x) : "v"(hp) : "memory");
#pragma unroll
            for (int s = 0; s < 32; ++s) __builtin_amdgcn_raw_buffer_store_b128(v, rs, base + s * 1024, 0, 0);
            asm volatile("s_waitcnt vmcnt(32)" : "+v"(x)::"memory");
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        tsum += t1 - t0 + (x & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (lane == 0) atomicAdd(out, tsum);
}

int main() {
    const size_t cold_bytes = 4ull << 30;
    uint32_t *cold, *hot, *big;
    unsigned long long* cnt;
    (void)hipMalloc(&cold, cold_bytes);
    (void)hipMalloc(&hot, 1 << 22);
    (void)hipMalloc(&big, (size_t)2 << 30);
    (void)hipMalloc(&cnt, 64);
    (void)hipMemset(cold, 0x11, cold_bytes);
    (void)hipMemset(hot, 0x22, 1 << 22);
    (void)hipMemset(cnt, 0, 64);
    hipLaunchKernelGGL(probe1, dim3(1024), dim3(256), 0, 0, cold, hot, cold_bytes / 4, 64, cnt);
    (void)hipDeviceSynchronize();
    unsigned long long h[8];
    (void)hipMemcpy(h, cnt, 64, hipMemcpyDeviceToHost);
    printf("test 1: cold load still pending after `vmcnt(4)` with 4 younger hot stores: %llu of %llu trials -> %s\n", h[0], h[1],
           h[0] ? "stores retire AHEAD of older loads" : "in-order retirement held in every trial");
    const int rounds = 16;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(cnt, 0, 64);
        hipLaunchKernelGGL(probe2<0>, dim3(256), dim3(256), 0, 0, big, hot, cnt + 2, rounds);
        hipLaunchKernelGGL(probe2<1>, dim3(256), dim3(256), 0, 0, big, hot, cnt + 3, rounds);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, cnt, 64, hipMemcpyDeviceToHost);
        printf("test 2: 32 x 1 KB stores per wave on every CU; avg cycles (s_memtime) until the wait returns: stores-then-load + vmcnt(0): %.0f   "
               "load-then-stores + vmcnt(32): %.0f\n", (double)h[2] / (256.0 * 4 * rounds), (double)h[3] / (256.0 * 4 * rounds));
    }
    return 0;
}
