// Probe (round 5): the store BURST of the persistent GEMM's epilogue.  256 workgroups x 4 waves, every wave writes its 128 x 128
// bf16 block of a 256 x 256 tile (32 KB = 32 buffer_store_dwordx4 per wave), round after round with nothing in between — the
// bandwidth the memory system absorbs such a burst with, per lane -> address pattern and cache policy:
//   pat 0: the epilogue's pattern, 16 rows x 64 B per instruction (row = lane & 15, 16-byte chunk = lane >> 4)
//   pat 1: 8 rows x 128 B per instruction (row = (lane & 15) >> 1, chunk = ((lane & 1) << 2) | (lane >> 4))
//   pat 2: lane-linear, 4 rows x 256 B per instruction (row = lane >> 4, chunk = lane & 15) — what an LDS transpose would give
//   pat 3: pat 0 with the nt (streaming) hint, pat 4: pat 0 with sc1 (write-through)
// hipcc --offload-arch=gfx950 -O3 store_burst.hip -o store_burst && ./store_burst [N] [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int PAT>
__global__ void __launch_bounds__(256) burst(uint16_t* C, int M, int N, int tiles_n, int rounds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * 128, wn0 = (wave & 1) * 128;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(C, 0, (uint32_t)((size_t)M * N * 2), 0x00020000);
    for (int r = 0; r < rounds; ++r) {
        const int t = r * (int)gridDim.x + (int)blockIdx.x;
        const int m0 = (t / tiles_n) * 256 + wm0, n0 = (t % tiles_n) * 256 + wn0;
        const u32x4 v = {(uint32_t)t, (uint32_t)lane, 0u, 0u};
        if (PAT == 0 || PAT >= 3) {
            // 8 fragment rows x 4 stores: 16 rows x 64 B each
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const uint32_t off = (uint32_t)(m0 + i * 16 + (lane & 15)) * (uint32_t)N * 2u + (uint32_t)(n0 + s * 32 + (lane >> 4) * 8) * 2u;
                    if (PAT == 3) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 2);
                    else if (PAT == 4) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
                    else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
                }
        } else if (PAT == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int row = (lane & 15) >> 1, ch = ((lane & 1) << 2) | (lane >> 4);
                    const uint32_t off = (uint32_t)(m0 + i * 8 + row) * (uint32_t)N * 2u + (uint32_t)(n0 + s * 64 + ch * 8) * 2u;
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
                }
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const uint32_t off = (uint32_t)(m0 + i * 4 + (lane >> 4)) * (uint32_t)N * 2u + (uint32_t)(n0 + (lane & 15) * 8) * 2u;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
            }
        }
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096, rounds = argc > 2 ? atoi(argv[2]) : 8;
    const int G = 256, tiles_n = N / 256, M = (G * rounds + tiles_n - 1) / tiles_n * 256;
    uint16_t* C;
    hipMalloc(&C, (size_t)M * N * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[5] = {"16 rows x 64 B", "8 rows x 128 B", "4 rows x 256 B", "16 x 64 B nt", "16 x 64 B sc1"};
    for (int rep = 0; rep < 4; ++rep)
        for (int pat = 0; pat < 5; ++pat) {
            hipEventRecord(e0, 0);
            switch (pat) {
                case 0: hipLaunchKernelGGL(burst<0>, dim3(G), dim3(256), 0, 0, C, M, N, tiles_n, rounds); break;
                case 1: hipLaunchKernelGGL(burst<1>, dim3(G), dim3(256), 0, 0, C, M, N, tiles_n, rounds); break;
                case 2: hipLaunchKernelGGL(burst<2>, dim3(G), dim3(256), 0, 0, C, M, N, tiles_n, rounds); break;
                case 3: hipLaunchKernelGGL(burst<3>, dim3(G), dim3(256), 0, 0, C, M, N, tiles_n, rounds); break;
                default: hipLaunchKernelGGL(burst<4>, dim3(G), dim3(256), 0, 0, C, M, N, tiles_n, rounds); break;
            }
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("N %d rounds %d  %-16s %.1f us  %.2f TB/s  (%.2f us per 32 MB round)\n", N, rounds, names[pat], ms * 1e3,
                            (double)G * rounds * 131072 / (ms * 1e-3) / 1e12, ms * 1e3 / rounds);
        }
    return 0;
}
