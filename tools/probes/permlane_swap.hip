// What v_permlane16_swap_b32 / v_permlane32_swap_b32 do on gfx950 (lane -> which value ends up where).
// x = lane id, y = 1000 + lane id; prints the four results per lane for lanes 0, 15, 16, 31, 32, 47, 48, 63.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void k(uint32_t* out) {
    uint32_t x = threadIdx.x, y = threadIdx.x + 1000;
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    auto q = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    out[threadIdx.x * 4] = r[0]; out[threadIdx.x * 4 + 1] = r[1];
    out[threadIdx.x * 4 + 2] = q[0]; out[threadIdx.x * 4 + 3] = q[1];
}
int main() {
    uint32_t* d; hipMalloc(&d, 64 * 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    uint32_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 1, 15, 16, 17, 31, 32, 47, 48, 63})
        printf("lane %2d: swap16 -> (%u, %u)   swap32 -> (%u, %u)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
