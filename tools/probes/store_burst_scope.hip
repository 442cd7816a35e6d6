// Probe (round 5): is the ~6 TB/s at which the chip absorbs a store burst a CHIP-wide limit (HBM) or a per-XCD one (L2 -> fabric)?
// 256 workgroups (one per CU; workgroup b sits on XCD b & 7), but only XCDs < nx and, inside an XCD, only workgroups (b >> 3) < ncu
// store: each active workgroup writes `rounds` x 128 KB (4 waves x 32 buffer_store_dwordx4 of 1 KB), as the GEMM epilogue does.
//   hipcc --offload-arch=gfx950 -O3 store_burst_scope.hip -o store_burst_scope && ./store_burst_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int PAT>       // 0: 16 rows x 64 B per instruction (the round-4 epilogue), 1: 8 rows x 128 B (whole lines)
__global__ void __launch_bounds__(256) burst(uint16_t* C, int rounds, int nx, int ncu, unsigned long long* cyc) {
    const int b = blockIdx.x, xcd = b & 7, cu = b >> 3;
    if (xcd >= nx || cu >= ncu) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(C, 0, 0xffffffffu, 0x00020000);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
        // a 256 x 256 bf16 tile of a [*, 4096] matrix: row stride 8 KB, this wave's 128 x 128 quarter, 16 rows x 64 B per instruction
        const uint32_t tile = (uint32_t)(r * 256 + b);
        const uint32_t base = (tile >> 4) * (256u * 8192u) + (tile & 15) * 512u + (wave >> 1) * (128u * 8192u) + (wave & 1) * 256u;
        const u32x4 v = {(uint32_t)r, (uint32_t)lane, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (PAT == 0) __builtin_amdgcn_raw_buffer_store_b128(v, rs, base + (uint32_t)(i * 16 + (lane & 15)) * 8192u + (uint32_t)(s * 64 + (lane >> 4) * 16), 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(v, rs, base + (uint32_t)(i * 16 + (s >> 1) * 8 + (lane >> 3)) * 8192u + (uint32_t)((s & 1) * 128 + (lane & 7) * 16), 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) atomicMax(cyc, t1 - t0);
}

int main() {
    const int rounds = 8;
    uint16_t* C;
    unsigned long long* cyc;
    (void)hipMalloc(&C, (size_t)rounds * 256 * 131072);
    (void)hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int cfg[][2] = {{8, 32}, {4, 32}, {2, 32}, {1, 32}, {8, 16}, {8, 8}, {8, 4}, {1, 8}, {1, 1}};
    for (int rep = 0; rep < 2; ++rep)
      for (int pat = 0; pat < 2; ++pat)
        for (auto& c : cfg) {
            (void)hipMemset(cyc, 0, 8);
            (void)hipEventRecord(e0, 0);
            if (pat == 0) hipLaunchKernelGGL(burst<0>, dim3(256), dim3(256), 0, 0, C, rounds, c[0], c[1], cyc);
            else hipLaunchKernelGGL(burst<1>, dim3(256), dim3(256), 0, 0, C, rounds, c[0], c[1], cyc);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h;
            (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const double bytes = (double)c[0] * c[1] * rounds * 131072;
            if (rep) printf("%s XCDs %d x CUs %2d: %7.1f us  %5.2f TB/s  = %6.1f GB/s per XCD, %5.1f B/clk/CU (%.0f clk per 128 KB tile)\n", pat ? "8 x 128 B" : "16 x 64 B", c[0], c[1], ms * 1e3,
                            bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / c[0], (double)rounds * 131072 / (double)h, (double)h / rounds);
        }
    return 0;
}
