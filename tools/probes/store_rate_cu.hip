// Probe (round 5): the store throughput of ONE CU (and of every CU at once) by lane -> address pattern.  tools/probes/store_burst_scope
// showed that the GEMM epilogue's pattern moves ~16 B/clk/CU even when a single CU stores: the limit is in the CU's store path, not in HBM.
// Each active workgroup (4 waves) writes `rounds` x 128 KB; patterns (per buffer_store_dwordx4 = 1 KB):
//   0: 16 rows x 64 B   (the epilogue: row = lane & 15, chunk = lane >> 4; row stride 8 KB)
//   1:  8 rows x 128 B  2: 4 rows x 256 B   3: 2 rows x 512 B   4: 1 KB contiguous
//   5: pattern 0 with row stride 2 KB (N = 1024)   6: pattern 0 as dwordx2 (16 rows x 32 B, twice the instructions)
//   hipcc --offload-arch=gfx950 -O3 store_rate_cu.hip -o store_rate_cu && ./store_rate_cu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

template <int PAT>
__global__ void __launch_bounds__(256) burst(uint16_t* C, int rounds, int nactive, unsigned long long* cyc) {
    const int b = blockIdx.x;
    if (b >= nactive) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(C, 0, 0xffffffffu, 0x00020000);
    const uint32_t stride = PAT == 5 ? 2048u : 8192u;
    const int rows_per = PAT == 1 ? 8 : PAT == 2 ? 4 : PAT == 3 ? 2 : PAT == 4 ? 1 : 16;       // rows one instruction touches
    const uint32_t row = PAT == 1 ? (lane >> 3) : PAT == 2 ? (lane >> 4) : PAT == 3 ? (lane >> 5) : PAT == 4 ? 0 : (lane & 15);
    const uint32_t col = PAT == 1 ? (lane & 7) * 16 : PAT == 2 ? (lane & 15) * 16 : PAT == 3 ? (lane & 31) * 16 : PAT == 4 ? lane * 16 : (lane >> 4) * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
        const uint32_t tile = (uint32_t)(r * 256 + b);
        const uint32_t base = tile * (256u * 512u * (stride / 2048u)) ;     // tiles far apart; inside: 256 rows x 512 B at `stride`
        const u32x4 v = {(uint32_t)r, (uint32_t)lane, 0u, 0u};
        // this wave's quarter: 128 rows x 256 B = 32 instructions of 1 KB
        const uint32_t wbase = base + (wave >> 1) * (128u * stride) + (wave & 1) * 256u;
        if (PAT == 6) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{v[0], v[1]}, rs, wbase + (uint32_t)(i * 16 + (lane & 15)) * stride + (uint32_t)(s * 32 + (lane >> 4) * 8), 0, 0);
        } else {
            // cover 128 rows x 256 B with instructions of rows_per rows x (1024 / rows_per) B
            const int per_row_instr = 256 / (1024 / rows_per) > 0 ? 256 / (1024 / rows_per) : 1;     // instructions side by side in a 256-B row span
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                uint32_t off;
                if (PAT == 3) off = wbase + (uint32_t)(k * 2 + row) * stride * 2u / 2u + col - (col >= 256 ? 256u - stride : 0u) * 0u;   // 512 B rows: spill into the neighbour quarter (same bytes total)
                else if (PAT == 4) off = base + (uint32_t)(wave * 32 + k) * 1024u + col;                                             // purely linear
                else off = wbase + (uint32_t)((k / per_row_instr) * rows_per + row) * stride + (uint32_t)(k % per_row_instr) * (1024u / rows_per) + col;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) atomicMax(cyc, t1 - t0);
}

int main() {
    const int rounds = 8;
    uint16_t* C;
    unsigned long long* cyc;
    (void)hipMalloc(&C, (size_t)3 << 30);
    (void)hipMalloc(&cyc, 8);
    const char* names[7] = {"16 x 64 B", "8 x 128 B", "4 x 256 B", "2 x 512 B", "1 KB linear", "16 x 64 B, 2 KB stride", "16 x 32 B (dwordx2)"};
    const int actives[3] = {1, 8, 256};
    for (int rep = 0; rep < 2; ++rep)
        for (int a = 0; a < 3; ++a)
            for (int pat = 0; pat < 7; ++pat) {
                (void)hipMemset(cyc, 0, 8);
                const int na = actives[a];
                switch (pat) {
                    case 0: hipLaunchKernelGGL(burst<0>, dim3(256), dim3(256), 0, 0, C, rounds, na, cyc); break;
                    case 1: hipLaunchKernelGGL(burst<1>, dim3(256), dim3(256), 0, 0, C, rounds, na, cyc); break;
                    case 2: hipLaunchKernelGGL(burst<2>, dim3(256), dim3(256), 0, 0, C, rounds, na, cyc); break;
                    case 3: hipLaunchKernelGGL(burst<3>, dim3(256), dim3(256), 0, 0, C, rounds, na, cyc); break;
                    case 4: hipLaunchKernelGGL(burst<4>, dim3(256), dim3(256), 0, 0, C, rounds, na, cyc); break;
                    case 5: hipLaunchKernelGGL(burst<5>, dim3(256), dim3(256), 0, 0, C, rounds, na, cyc); break;
                    default: hipLaunchKernelGGL(burst<6>, dim3(256), dim3(256), 0, 0, C, rounds, na, cyc); break;
                }
                (void)hipDeviceSynchronize();
                unsigned long long h;
                (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
                if (rep) printf("active workgroups %3d  %-24s %6.0f clk per 128 KB tile = %5.1f B/clk/CU\n", na, names[pat], (double)h / rounds, (double)rounds * 131072 / (double)h);
            }
    return 0;
}
