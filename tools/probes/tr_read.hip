// Probe: what does ds_read_b64_tr_b16 deliver?  LDS holds element index e at position e (uint16); every lane reads the
// 8 bytes at element offset 4*lane (lane-linear) and prints its four values; then a second layout: lane (4a+t) of each
// 16-lane block reads row t, columns 4a..4a+3 of a [4][16] block (the MFMA B-fragment recipe).
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read.hip -o tools/probes/tr_read
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(uint16_t* out) {
    __shared__ uint16_t lds[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    // (1) lane-linear: lane l reads elements 4l..4l+3
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + 4 * lane));
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = (uint16_t)v[r];
    // (2) block layout [g][t = key 0..3][16 d]: lane (g, j): a = j >> 2 ... reads row (j & 3), columns 4*(j >> 2)..+3
    const int g = lane >> 4, j = lane & 15;
    const int off = g * 64 + (j & 3) * 16 + (j >> 2) * 4;
    s16x4 w = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off));
    for (int r = 0; r < 4; ++r) out[256 + lane * 4 + r] = (uint16_t)w[r];
}
int main() {
    uint16_t* d;
    hipMalloc(&d, 1024 * 2);
    probe<<<1, 64>>>(d);
    uint16_t h[512];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("(1) lane-linear: lane l reads elements 4l..4l+3\n");
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    printf("(2) lane (g, j) reads [g][row j&3][cols 4*(j>>2)..+3] of a [4][16] block (element = g*64 + row*16 + col)\n");
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[256 + l * 4], h[256 + l * 4 + 1], h[256 + l * 4 + 2], h[256 + l * 4 + 3]);
    return 0;
}
