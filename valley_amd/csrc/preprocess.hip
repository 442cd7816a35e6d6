// Frame preprocessing kernels: Pillow's 8-bit separable bilinear resample, restricted to
// the centre crop, + /255 + CLIP normalisation.  Integer arithmetic is bit-exact with
// libImaging/Resample.c: acc = 2^21 + sum(pixel * tap), out = clip8(acc >> 22), uint8 between the passes.
// HBM-bound: per frame H*W*3 bytes read once, 224*224*3 outputs.
#include "common.hpp"
#include "../../include/valley_hip.h"

namespace {

constexpr int PBITS = 22;

VLY_DEVICE uint8_t clip8(int v) { v >>= PBITS; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// horizontal pass: in u8 [T,H,W,3] -> out u8 [T,H,OW,3]; bounds/taps hold only the OW kept columns.
__global__ void __launch_bounds__(256) resize_h_kernel(const uint8_t* __restrict__ in, const int* __restrict__ bounds,
                                                       const int* __restrict__ taps, uint8_t* __restrict__ out,
                                                       long rows, int W, int OW, int ksize) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // (row, ox)
    if (idx >= rows * OW) return;
    const int ox = (int)(idx % OW);
    const long row = idx / OW;
    const int x0 = bounds[2 * ox], n = bounds[2 * ox + 1];
    const uint8_t* p = in + (row * W + x0) * 3;
    const int* k = taps + (size_t)ox * ksize;
    int a0 = 1 << (PBITS - 1), a1 = a0, a2 = a0;
    for (int x = 0; x < n; ++x) {
        const int w = k[x];
        a0 += p[3 * x] * w; a1 += p[3 * x + 1] * w; a2 += p[3 * x + 2] * w;
    }
    uint8_t* o = out + idx * 3;
    o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
}

// vertical pass on the kept rows + crop offset in x + normalise: in u8 [T,H,W,3] -> out [T,3,OS,OS].
template <typename OT>
__global__ void __launch_bounds__(256) resize_v_norm_kernel(const uint8_t* __restrict__ in, const int* __restrict__ bounds,
                                                            const int* __restrict__ taps, const float* __restrict__ mean,
                                                            const float* __restrict__ stdv, OT* __restrict__ out,
                                                            int T, int H, int W, int x_off, int OS, int ksize) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // (t, oy, ox)
    if (idx >= (long)T * OS * OS) return;
    const int ox = (int)(idx % OS);
    const int oy = (int)((idx / OS) % OS);
    const int t = (int)(idx / ((long)OS * OS));
    const int y0 = bounds[2 * oy], n = bounds[2 * oy + 1];
    const uint8_t* p = in + (((size_t)t * H + y0) * W + x_off + ox) * 3;
    const int* k = taps + (size_t)oy * ksize;
    int a[3] = {1 << (PBITS - 1), 1 << (PBITS - 1), 1 << (PBITS - 1)};
    for (int y = 0; y < n; ++y) {
        const int w = k[y];
        const uint8_t* q = p + (size_t)y * W * 3;
        a[0] += q[0] * w; a[1] += q[1] * w; a[2] += q[2] * w;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = ((float)clip8(a[c]) / 255.0f - mean[c]) / stdv[c];
        const size_t o = (((size_t)t * 3 + c) * OS + oy) * OS + ox;
        if constexpr (sizeof(OT) == 2) out[o] = f2h(v);
        else out[o] = v;
    }
}

}  // namespace

extern "C" int vly_resize_h_u8(const uint8_t* in, const int32_t* bounds, const int32_t* taps, uint8_t* out, int T, int H, int W,
                               int OW, int ksize, void* stream) {
    if (T <= 0 || H <= 0 || W <= 0 || OW <= 0 || ksize <= 0) { vly_set_error("vly_resize_h_u8: bad args"); return -22; }
    const long rows = (long)T * H, n = rows * OW;
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, bounds, taps, out,
                       rows, W, OW, ksize);
    return vly_check_launch("vly_resize_h_u8");
}

extern "C" int vly_resize_v_norm(const uint8_t* in, const int32_t* bounds, const int32_t* taps, const float* mean,
                                 const float* stdv, void* out, int T, int H, int W, int x_off, int OS, int ksize, int out_f32,
                                 void* stream) {
    if (T <= 0 || H <= 0 || W <= 0 || OS <= 0 || ksize <= 0 || x_off < 0 || x_off + OS > W) {
        vly_set_error("vly_resize_v_norm: bad args");
        return -22;
    }
    const long n = (long)T * OS * OS;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (out_f32)
        hipLaunchKernelGGL((resize_v_norm_kernel<float>), grid, block, 0, (hipStream_t)stream, in, bounds, taps, mean, stdv,
                           (float*)out, T, H, W, x_off, OS, ksize);
    else
        hipLaunchKernelGGL((resize_v_norm_kernel<uint16_t>), grid, block, 0, (hipStream_t)stream, in, bounds, taps, mean, stdv,
                           (uint16_t*)out, T, H, W, x_off, OS, ksize);
    return vly_check_launch("vly_resize_v_norm");
}
