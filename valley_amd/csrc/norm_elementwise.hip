// HBM-bound row kernels of the hot path: LayerNorm / RMSNorm, patch im2col, ViT embedding
// assembly, temporal pooling, embedding gather + visual splice, RoPE + KV-cache append, argmax.
// All are one-wave-per-row (or per contiguous run) with 16-byte accesses; the roofline that bounds
// them is HBM bandwidth, algorithmic bytes = bytes read + bytes written stated per kernel.
#include "common.hpp"
#include "../../include/valley_hip.h"

namespace {

constexpr int ROWS_PER_BLOCK = 2;           // 2 waves per block, one row each (more blocks for small M)

// ---------------------------------------------------------------------------------------------
// LayerNorm / RMSNorm: fp32 [M,D] -> bf16 (and optionally fp32).  One wave per row, the row is
// read once into registers (NV float4 per lane), statistics in fp32 (two-pass variance on the
// register copy, like torch), written as 8-byte bf16x4.  bytes/row = 4D read + 2D (+4D) written.
// ---------------------------------------------------------------------------------------------
// ADD: x is the fp32 residual stream, `delta` the bf16 output of the sub-layer GEMM: the kernel first
// does x += delta (written back), then normalises the sum — the residual update rides on the
// streaming norm kernel (5-6 TB/s) instead of the GEMM epilogue (64-byte row pieces, exposed).
template <int NV, bool RMS, bool ADD>
__global__ void __launch_bounds__(128) norm_kernel(float* __restrict__ x, const uint16_t* __restrict__ delta,
                                                   const uint16_t* __restrict__ delta2, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, uint16_t* __restrict__ y16,
                                                   float* __restrict__ y32, int M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= M) return;
    float4* xr = (float4*)(x + (size_t)row * D);
    const int nvec = D >> 2;                 // float4 per row; lane handles v = lane + 64*i
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        v[i] = (c < nvec) ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (ADD) {
            if (c < nvec) {
                const u32x2 dk = *(const u32x2*)(delta + (size_t)row * D + 4 * c);
                v[i].x += h_lo(dk[0]); v[i].y += h_hi(dk[0]);
                v[i].z += h_lo(dk[1]); v[i].w += h_hi(dk[1]);
                if (delta2) {                                   // second split-K partial of the sub-layer GEMM
                    const u32x2 d2 = *(const u32x2*)(delta2 + (size_t)row * D + 4 * c);
                    v[i].x += h_lo(d2[0]); v[i].y += h_hi(d2[0]);
                    v[i].z += h_lo(d2[1]); v[i].w += h_hi(d2[1]);
                }
                xr[c] = v[i];
            }
        }
        if constexpr (RMS) s += vly_sumsq4(v[i].x, v[i].y, v[i].z, v[i].w);
        else s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    if (gamma == nullptr) return;            // ADD-only call (last residual update of the stack)
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if constexpr (RMS) {
        rstd = rsqrtf(s / (float)D + eps);
    } else {
        mean = s / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nvec) {
                const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += a * a + b * b + cc * cc + d * d;
            }
        }
        q = wave_sum(q);
        rstd = rsqrtf(q / (float)D + eps);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c >= nvec) continue;
        const float4 gm = ((const float4*)gamma)[c];
        float4 o;
        if constexpr (RMS) {
            // hf: weight * (x * rsqrt(var + eps))
            o.x = gm.x * (v[i].x * rstd); o.y = gm.y * (v[i].y * rstd);
            o.z = gm.z * (v[i].z * rstd); o.w = gm.w * (v[i].w * rstd);
        } else {
            const float4 bt = ((const float4*)beta)[c];
            o.x = (v[i].x - mean) * rstd * gm.x + bt.x; o.y = (v[i].y - mean) * rstd * gm.y + bt.y;
            o.z = (v[i].z - mean) * rstd * gm.z + bt.z; o.w = (v[i].w - mean) * rstd * gm.w + bt.w;
        }
        if (y16) {
            u32x2 pk;
            pk[0] = pack_h2(o.x, o.y);
            pk[1] = pack_h2(o.z, o.w);
            *(u32x2*)(y16 + (size_t)row * D + 4 * c) = pk;
        }
        if (y32) ((float4*)(y32 + (size_t)row * D))[c] = o;
    }
}

// Few-rows variant: one 256-thread workgroup per row so the row's loads are spread over 4 waves instead
// of queued in one; block reduction through LDS.  Same ADD semantics as norm_kernel.
template <bool RMS, bool ADD>
__global__ void __launch_bounds__(256) norm_row_kernel(float* __restrict__ x, const uint16_t* __restrict__ delta,
                                                       const uint16_t* __restrict__ delta2, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, uint16_t* __restrict__ y16,
                                                       float* __restrict__ y32, int D, float eps) {
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    float4* xr = (float4*)(x + (size_t)row * D);
    const int nvec = D >> 2;
    float4 v[8];                                               // D <= 8192
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + 256 * i;
        v[i] = (c < nvec) ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (ADD) {
            if (c < nvec) {
                const u32x2 dk = *(const u32x2*)(delta + (size_t)row * D + 4 * c);
                v[i].x += h_lo(dk[0]); v[i].y += h_hi(dk[0]);
                v[i].z += h_lo(dk[1]); v[i].w += h_hi(dk[1]);
                if (delta2) {
                    const u32x2 d2 = *(const u32x2*)(delta2 + (size_t)row * D + 4 * c);
                    v[i].x += h_lo(d2[0]); v[i].y += h_hi(d2[0]);
                    v[i].z += h_lo(d2[1]); v[i].w += h_hi(d2[1]);
                }
                xr[c] = v[i];
            }
        }
        if constexpr (RMS) s += vly_sumsq4(v[i].x, v[i].y, v[i].z, v[i].w);
        else s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    if (gamma == nullptr) return;                              // ADD-only call
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = red[0] + red[1] + red[2] + red[3];
    float mean = 0.f, rstd;
    if constexpr (RMS) {
        rstd = rsqrtf(s / (float)D + eps);
    } else {
        mean = s / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + 256 * i;
            if (c < nvec) {
                const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += a * a + b * b + cc * cc + d * d;
            }
        }
        q = wave_sum(q);
        if (lane == 0) red[4 + wave] = q;
        __syncthreads();
        rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)D + eps);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + 256 * i;
        if (c >= nvec) continue;
        const float4 gm = ((const float4*)gamma)[c];
        float4 o;
        if constexpr (RMS) {
            o.x = gm.x * (v[i].x * rstd); o.y = gm.y * (v[i].y * rstd);
            o.z = gm.z * (v[i].z * rstd); o.w = gm.w * (v[i].w * rstd);
        } else {
            const float4 bt = ((const float4*)beta)[c];
            o.x = (v[i].x - mean) * rstd * gm.x + bt.x; o.y = (v[i].y - mean) * rstd * gm.y + bt.y;
            o.z = (v[i].z - mean) * rstd * gm.z + bt.z; o.w = (v[i].w - mean) * rstd * gm.w + bt.w;
        }
        if (y16) {
            u32x2 pk;
            pk[0] = pack_h2(o.x, o.y);
            pk[1] = pack_h2(o.z, o.w);
            *(u32x2*)(y16 + (size_t)row * D + 4 * c) = pk;
        }
        if (y32) ((float4*)(y32 + (size_t)row * D))[c] = o;
    }
}

template <bool RMS>
int launch_norm(const float* x, const void* delta, const float* gamma, const float* beta, void* y16, float* y32, int M, int D,
                float eps, hipStream_t st, const char* name, const void* delta2 = nullptr) {
    if (M <= 0 || D <= 0 || D % 4 || D > 8192 || ((uintptr_t)x & 15) || ((uintptr_t)gamma & 15) || ((uintptr_t)delta & 7) || ((uintptr_t)delta2 & 7) ||
        (beta && ((uintptr_t)beta & 15)) || ((uintptr_t)y16 & 7) || ((uintptr_t)y32 & 15)) {
        vly_set_error("%s: unsupported shape/alignment M=%d D=%d", name, M, D);
        return -22;
    }
    // one workgroup per row when one wave per row would leave the chip short of waves (decode, and the
    // Llama residual stream at M = B*S ~ 1-3 k rows of 16-20 KB: 5 waves per CU could not cover HBM latency)
    if (M <= 64 || (D >= 2048 && M <= 4096)) {
        if (delta) hipLaunchKernelGGL((norm_row_kernel<RMS, true>), dim3(M), dim3(256), 0, st, (float*)x, (const uint16_t*)delta,
                                      (const uint16_t*)delta2, gamma, beta, (uint16_t*)y16, y32, D, eps);
        else hipLaunchKernelGGL((norm_row_kernel<RMS, false>), dim3(M), dim3(256), 0, st, (float*)x, (const uint16_t*)nullptr,
                                (const uint16_t*)nullptr, gamma, beta, (uint16_t*)y16, y32, D, eps);
        return vly_check_launch(name);
    }
    dim3 grid((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(64 * ROWS_PER_BLOCK);
    const int nv = (D / 4 + 63) / 64;
#define VLY_NORM(NV)                                                                                              \
    do {                                                                                                          \
        if (delta) hipLaunchKernelGGL((norm_kernel<NV, RMS, true>), grid, block, 0, st, (float*)x, (const uint16_t*)delta, \
                                      (const uint16_t*)delta2, gamma, beta, (uint16_t*)y16, y32, M, D, eps);      \
        else hipLaunchKernelGGL((norm_kernel<NV, RMS, false>), grid, block, 0, st, (float*)x, (const uint16_t*)nullptr,   \
                                (const uint16_t*)nullptr, gamma, beta, (uint16_t*)y16, y32, M, D, eps);           \
    } while (0)
    if (nv <= 4) VLY_NORM(4);
    else if (nv <= 8) VLY_NORM(8);
    else if (nv <= 16) VLY_NORM(16);
    else if (nv <= 20) VLY_NORM(20);
    else VLY_NORM(32);
#undef VLY_NORM
    return vly_check_launch(name);
}

// ---------------------------------------------------------------------------------------------
// im2col for Conv2d(3->1024, k=14, s=14): images bf16 [F,3,224,224] -> [F*256, 640].
// One wave per output row (patch): 640 columns = 10 per lane.  Column k = c*196 + ky*14 + kx.
// bytes/frame = 301,056 read + 327,680 written.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) patchify_kernel(const uint16_t* __restrict__ img, uint16_t* __restrict__ out, int rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int f = row >> 8, p = row & 255, py = p >> 4, px = p & 15;
    const uint16_t* base = img + (size_t)f * 3 * 224 * 224 + (size_t)(py * 14) * 224 + px * 14;
    uint16_t* o = out + (size_t)row * 640;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int k = lane + 64 * i;
        uint16_t v = 0;
        if (k < 588) {
            const int c = k / 196, r = k - c * 196, ky = r / 14, kx = r - ky * 14;
            v = base[(size_t)c * 224 * 224 + ky * 224 + kx];
        }
        o[k] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// ViT embedding assembly + pre_layrnorm: row t of frame f = (t == 0 ? cls : patch_out[f*256+t-1], fp32)
// + pos[t], then LayerNorm -> fp32 residual stream.  One wave per row, D = 1024 (4 float4/lane).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) vit_embed_ln_kernel(const float* __restrict__ patch_out, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ h, int rows, float eps) {
    constexpr int D = 1024;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int f = row / 257, t = row - f * 257;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        float4 e;
        if (t == 0) {
            e = ((const float4*)cls)[c];
        } else {
            e = ((const float4*)(patch_out + ((size_t)(f * 256 + t - 1)) * D))[c];
        }
        const float4 pp = ((const float4*)(pos + (size_t)t * D))[c];
        e.x += pp.x; e.y += pp.y; e.z += pp.z; e.w += pp.w;
        v[i] = e;
        s += e.x + e.y + e.z + e.w;
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += a * a + b * b + c * c + d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        const float4 gm = ((const float4*)gamma)[c], bt = ((const float4*)beta)[c];
        float4 o;
        o.x = (v[i].x - mean) * rstd * gm.x + bt.x; o.y = (v[i].y - mean) * rstd * gm.y + bt.y;
        o.z = (v[i].z - mean) * rstd * gm.z + bt.z; o.w = (v[i].w - mean) * rstd * gm.w + bt.w;
        ((float4*)(h + (size_t)row * D))[c] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// Temporal pooling: feats fp32 [B,T,257,W] -> out bf16 [B,256+T,W].
// Thread per 4 columns of one output row; rows < 256 reduce over T, rows >= 256 copy a CLS row.
// bytes/clip = 4*T*257*W read + 2*(256+T)*W written.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pool_kernel(const float* __restrict__ feats, uint16_t* __restrict__ out, int B, int T, int W, int mode,
                                                   const float* __restrict__ scores) {
    const int wv = W >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * (256 + T) * wv;
    if (idx >= total) return;
    const int c = (int)(idx % wv);
    const long rr = idx / wv;
    const int r = (int)(rr % (256 + T)), b = (int)(rr / (256 + T));
    const float* fb = feats + (size_t)b * T * 257 * W;
    float4 o;
    if (r < 256 && mode == VLY_POOL_IMPORTANCE) {
        // valley_model.py:113-121: softmax over the T frame scores, weighted sum of the frames
        const float* sc = scores + (size_t)b * T;
        float mx = sc[0];
        for (int t = 1; t < T; ++t) mx = fmaxf(mx, sc[t]);
        float den = 0.f;
        for (int t = 0; t < T; ++t) den += __expf(sc[t] - mx);
        o = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < T; ++t) {
            const float wt = __expf(sc[t] - mx) / den;
            const float4 a = ((const float4*)(fb + ((size_t)t * 257 + 1 + r) * W))[c];
            o.x += wt * a.x; o.y += wt * a.y; o.z += wt * a.z; o.w += wt * a.w;
        }
    } else if (r < 256) {
        o = ((const float4*)(fb + (size_t)(1 + r) * W))[c];
        for (int t = 1; t < T; ++t) {
            const float4 a = ((const float4*)(fb + ((size_t)t * 257 + 1 + r) * W))[c];
            if (mode == VLY_POOL_MAX) { o.x = fmaxf(o.x, a.x); o.y = fmaxf(o.y, a.y); o.z = fmaxf(o.z, a.z); o.w = fmaxf(o.w, a.w); }
            else { o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }
        }
        if (mode == VLY_POOL_MEAN) { const float inv = 1.f / (float)T; o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv; }
    } else {
        o = ((const float4*)(fb + (size_t)(r - 256) * 257 * W))[c];
    }
    u32x2 pk;
    pk[0] = pack_h2(o.x, o.y);
    pk[1] = pack_h2(o.z, o.w);
    *(u32x2*)(out + ((size_t)b * (256 + T) + r) * W + 4 * c) = pk;
}

// ---------------------------------------------------------------------------------------------
// Embedding gather + visual splice: out[r,:] = (map[r] >= 0 ? embed[map[r]] : visual[-map[r]-1]).
// One wave per row, 16-byte bf16x8 reads, fp32 writes.  bytes/row = 2H read + 4H written.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) embed_splice_kernel(const int32_t* __restrict__ map, const uint16_t* __restrict__ embed,
                                                           const uint16_t* __restrict__ visual, float* __restrict__ out, int R, int H) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int v = map[row];
    const uint16_t* src = v >= 0 ? embed + (size_t)v * H : visual + (size_t)(-v - 1) * H;
    float* o = out + (size_t)row * H;
    for (int c = lane; c < (H >> 3); c += 64) {
        const u32x4 pk = *(const u32x4*)(src + 8 * c);
        float4 a, b;
        a.x = h_lo(pk[0]); a.y = h_hi(pk[0]);
        a.z = h_lo(pk[1]); a.w = h_hi(pk[1]);
        b.x = h_lo(pk[2]); b.y = h_hi(pk[2]);
        b.z = h_lo(pk[3]); b.w = h_hi(pk[3]);
        ((float4*)o)[2 * c] = a;
        ((float4*)o)[2 * c + 1] = b;
    }
}

// ---------------------------------------------------------------------------------------------
// RoPE (rotate-half) on q,k + KV-cache append.  One wave per (row, head): lane i handles the
// pair (d = i, d = i + 64) of head_dim 128.  cos/sin come from fp32 tables [ctx_max, 64] the host
// builds exactly as hf does (inv_freq = theta^(-2i/128), angle = pos * inv_freq, fp32).
// q' = q*cos + rot(q)*sin with rot(q)[d<64] = -q[d+64], rot(q)[d>=64] = q[d-64].
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rope_kv_kernel(uint16_t* __restrict__ qkv, uint16_t* __restrict__ kc, uint16_t* __restrict__ vc,
                                                      const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                      int B, int S, int heads, int past, const int32_t* __restrict__ past_dev, int ctx_max) {
    // 8 lanes per (row, head): lane j rotates dims [8j, 8j+8) against [64+8j, 64+8j+8) with 16-byte accesses.
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long unit = t >> 3;                                      // (b*S + s)*heads + h
    if (unit >= (long)B * S * heads) return;
    const int j = (int)(t & 7);
    const int h = (int)(unit % heads);
    const long rs = unit / heads;
    const int s = (int)(rs % S), b = (int)(rs / S);
    const int Hq = heads * 128;
    if (past_dev) past = min(*past_dev, ctx_max - S);       // device-side position (hipGraph replay); clamp = no OOB ever
    const int pos = past + s;
    const float4 c0 = *(const float4*)(cos_t + (size_t)pos * 64 + 8 * j), c1 = *(const float4*)(cos_t + (size_t)pos * 64 + 8 * j + 4);
    const float4 s0 = *(const float4*)(sin_t + (size_t)pos * 64 + 8 * j), s1 = *(const float4*)(sin_t + (size_t)pos * 64 + 8 * j + 4);
    const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    uint16_t* q = qkv + (size_t)rs * 3 * Hq + h * 128 + 8 * j;
    uint16_t* k = q + Hq;
    const uint16_t* v = q + 2 * Hq;
    const size_t co = (((size_t)b * heads + h) * ctx_max + pos) * 128 + 8 * j;
    auto rot = [&](const u32x4 lo, const u32x4 hi, u32x4& olo, u32x4& ohi) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a0 = h_lo(lo[i]), a1 = h_hi(lo[i]);
            const float b0 = h_lo(hi[i]), b1 = h_hi(hi[i]);
            olo[i] = pack_h2(rope_rot(a0, b0, cs[2 * i], sn[2 * i], -1.f), rope_rot(a1, b1, cs[2 * i + 1], sn[2 * i + 1], -1.f));
            ohi[i] = pack_h2(rope_rot(b0, a0, cs[2 * i], sn[2 * i], 1.f), rope_rot(b1, a1, cs[2 * i + 1], sn[2 * i + 1], 1.f));
        }
    };
    u32x4 olo, ohi;
    rot(*(const u32x4*)q, *(const u32x4*)(q + 64), olo, ohi);
    *(u32x4*)q = olo;
    *(u32x4*)(q + 64) = ohi;
    rot(*(const u32x4*)k, *(const u32x4*)(k + 64), olo, ohi);
    *(u32x4*)(kc + co) = olo;
    *(u32x4*)(kc + co + 64) = ohi;
    *(u32x4*)(vc + co) = *(const u32x4*)v;
    *(u32x4*)(vc + co + 64) = *(const u32x4*)(v + 64);
}

// fp32 -> bf16 cast (round to nearest even), 8 elements per thread.
__global__ void __launch_bounds__(256) cast_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, long n8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const float4 a = ((const float4*)x)[2 * i], b = ((const float4*)x)[2 * i + 1];
    u32x4 pk;
    pk[0] = pack_h2(a.x, a.y); pk[1] = pack_h2(a.z, a.w);
    pk[2] = pack_h2(b.x, b.y); pk[3] = pack_h2(b.z, b.w);
    ((u32x4*)y)[i] = pk;
}

// argmax over rows of fp32 [M,N]; first maximal index (torch.argmax tie rule on CPU).  One 1024-thread
// workgroup per row, 16-byte loads issued four at a time (a 256-thread scalar loop spent 39 us per
// 32 k-wide row on dependent load latency: 0.7 % of a 13B decode step).
__global__ void __launch_bounds__(1024) argmax_kernel(const float* __restrict__ x, int32_t* __restrict__ idx, int N, int ld) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const float* r = x + (size_t)blockIdx.x * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    auto take = [&](float v, int i) {
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    };
    const int tid = threadIdx.x;
    if ((((uintptr_t)r) & 15) == 0) {
        const int nv = N >> 2;
        for (int c0 = 0; c0 < nv; c0 += 4096) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * 1024 + tid;
                v[u] = c < nv ? ((const float4*)r)[c] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = (c0 + u * 1024 + tid) * 4;
                take(v[u].x, i); take(v[u].y, i + 1); take(v[u].z, i + 2); take(v[u].w, i + 3);
            }
        }
        for (int i = (nv << 2) + tid; i < N; i += 1024) take(r[i], i);
    } else {
        for (int i = tid; i < N; i += 1024) take(r[i], i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        idx[blockIdx.x] = bi;
    }
}

}  // namespace

extern "C" int vly_layernorm(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                             int M, int D, float eps, void* stream) {
    if (!beta) { vly_set_error("vly_layernorm: beta is required"); return -22; }
    return launch_norm<false>(x, nullptr, gamma, beta, y_bf16, y_f32, M, D, eps, (hipStream_t)stream, "vly_layernorm");
}

extern "C" int vly_rmsnorm(const float* x, const float* gamma, void* y_bf16, int M, int D, float eps, void* stream) {
    return launch_norm<true>(x, nullptr, gamma, nullptr, y_bf16, nullptr, M, D, eps, (hipStream_t)stream, "vly_rmsnorm");
}

extern "C" int vly_add_layernorm(float* h, const void* delta_bf16, const float* gamma, const float* beta, void* y_bf16,
                                 int M, int D, float eps, void* stream) {
    if (!delta_bf16 || (gamma && !beta)) { vly_set_error("vly_add_layernorm: delta (and beta with gamma) required"); return -22; }
    return launch_norm<false>(h, delta_bf16, gamma, beta, y_bf16, nullptr, M, D, eps, (hipStream_t)stream, "vly_add_layernorm");
}

extern "C" int vly_add_rmsnorm(float* h, const void* delta_bf16, const float* gamma, void* y_bf16, int M, int D, float eps,
                               void* stream) {
    if (!delta_bf16) { vly_set_error("vly_add_rmsnorm: delta required"); return -22; }
    return launch_norm<true>(h, delta_bf16, gamma, nullptr, y_bf16, nullptr, M, D, eps, (hipStream_t)stream, "vly_add_rmsnorm");
}

extern "C" int vly_patchify(const void* images, void* patches, int F, void* stream) {
    if (F <= 0) { vly_set_error("vly_patchify: F=%d", F); return -22; }
    const int rows = F * 256;
    hipLaunchKernelGGL(patchify_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)images,
                       (uint16_t*)patches, rows);
    return vly_check_launch("vly_patchify");
}

extern "C" int vly_vit_embed_ln(const float* patch_out, const float* cls, const float* pos, const float* gamma,
                                const float* beta, float* h, int F, float eps, void* stream) {
    if (F <= 0) { vly_set_error("vly_vit_embed_ln: F=%d", F); return -22; }
    const int rows = F * 257;
    hipLaunchKernelGGL(vit_embed_ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       patch_out, cls, pos, gamma, beta, h, rows, eps);
    return vly_check_launch("vly_vit_embed_ln");
}

// Frame scores of the v2 "temporal importance" pooling: score[f] = w . flatten(feats[f, 1:257, :]) + b
// (valley_model.py:42,115-116: Linear(256*H -> 1) on the flattened patch tokens).  One workgroup per
// frame streams 256*W contiguous floats against the weight vector; HBM-bound, 8*256*W bytes per frame.
// 1024 threads x 4 independent 16-byte load pairs per iteration keep ~128 KB in flight per frame (the first version, 256
// threads x one pair, was latency-bound: 274 us for 16 frames at W = 4096, 4 % of the HBM rate).
__global__ void __launch_bounds__(1024) temporal_score_kernel(const float* __restrict__ feats, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ scores, int W) {
    __shared__ float red[16];
    const float4* x = (const float4*)(feats + ((size_t)blockIdx.x * 257 + 1) * W);
    const float4* wv = (const float4*)w;
    const int n4 = 64 * W;                                     // 256 * W / 4
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = threadIdx.x;
    for (; i + 3 * 1024 < n4; i += 4 * 1024) {
        const float4 a0 = x[i], a1 = x[i + 1024], a2 = x[i + 2048], a3 = x[i + 3072];
        const float4 b0 = wv[i], b1 = wv[i + 1024], b2 = wv[i + 2048], b3 = wv[i + 3072];
        s0 += a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w;
        s1 += a1.x * b1.x + a1.y * b1.y + a1.z * b1.z + a1.w * b1.w;
        s2 += a2.x * b2.x + a2.y * b2.y + a2.z * b2.z + a2.w * b2.w;
        s3 += a3.x * b3.x + a3.y * b3.y + a3.z * b3.z + a3.w * b3.w;
    }
    for (; i < n4; i += 1024) {
        const float4 a = x[i], b = wv[i];
        s0 += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    float s = wave_sum((s0 + s1) + (s2 + s3));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = bias ? bias[0] : 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k];
        scores[blockIdx.x] = t;
    }
}

extern "C" int vly_temporal_scores(const float* feats, const float* w, const float* bias, float* scores, int F, int W,
                                   void* stream) {
    if (F <= 0 || W <= 0 || W % 4 || ((uintptr_t)feats & 15) || ((uintptr_t)w & 15)) {
        vly_set_error("vly_temporal_scores: bad args F=%d W=%d", F, W);
        return -22;
    }
    hipLaunchKernelGGL(temporal_score_kernel, dim3(F), dim3(1024), 0, (hipStream_t)stream, feats, w, bias, scores, W);
    return vly_check_launch("vly_temporal_scores");
}

extern "C" int vly_pool_tokens(const float* feats, void* out, int B, int T, int W, int mode, const float* scores,
                               void* stream) {
    if (B <= 0 || T <= 0 || W <= 0 || W % 4 || mode < VLY_POOL_MEAN || mode > VLY_POOL_IMPORTANCE ||
        (mode == VLY_POOL_IMPORTANCE && !scores)) {
        vly_set_error("vly_pool_tokens: bad args B=%d T=%d W=%d mode=%d", B, T, W, mode);
        return -22;
    }
    const long total = (long)B * (256 + T) * (W / 4);
    hipLaunchKernelGGL(pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, feats,
                       (uint16_t*)out, B, T, W, mode, scores);
    return vly_check_launch("vly_pool_tokens");
}

extern "C" int vly_embed_splice(const int32_t* row_map, const void* embed, const void* visual, float* out, int R, int H,
                                void* stream) {
    if (R <= 0 || H <= 0 || H % 8) { vly_set_error("vly_embed_splice: bad args R=%d H=%d", R, H); return -22; }
    hipLaunchKernelGGL(embed_splice_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, row_map,
                       (const uint16_t*)embed, (const uint16_t*)visual, out, R, H);
    return vly_check_launch("vly_embed_splice");
}

extern "C" int vly_rope_kv(void* qkv, void* kcache, void* vcache, const float* cos_table, const float* sin_table,
                           int B, int S, int heads, int past_len, const int32_t* past_len_dev, int ctx_max, void* stream) {
    if (B <= 0 || S <= 0 || heads <= 0 || past_len < 0 || past_len + S > ctx_max) {
        vly_set_error("vly_rope_kv: bad args B=%d S=%d heads=%d past=%d ctx_max=%d", B, S, heads, past_len, ctx_max);
        return -22;
    }
    const long units = (long)B * S * heads;
    hipLaunchKernelGGL(rope_kv_kernel, dim3((unsigned)((units * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint16_t*)qkv,
                       (uint16_t*)kcache, (uint16_t*)vcache, cos_table, sin_table, B, S, heads, past_len, past_len_dev, ctx_max);
    return vly_check_launch("vly_rope_kv");
}

extern "C" int vly_add2_rmsnorm(float* h, const void* delta0_bf16, const void* delta1_bf16, const float* gamma, void* y_bf16,
                                int M, int D, float eps, void* stream) {
    if (!delta0_bf16 || !delta1_bf16) { vly_set_error("vly_add2_rmsnorm: two deltas required"); return -22; }
    return launch_norm<true>(h, delta0_bf16, gamma, nullptr, y_bf16, nullptr, M, D, eps, (hipStream_t)stream, "vly_add2_rmsnorm",
                             delta1_bf16);
}

extern "C" int vly_add2_layernorm(float* h, const void* delta0_bf16, const void* delta1_bf16, const float* gamma,
                                  const float* beta, void* y_bf16, int M, int D, float eps, void* stream) {
    if (!delta0_bf16 || !delta1_bf16 || (gamma && !beta)) {
        vly_set_error("vly_add2_layernorm: two deltas (and beta with gamma) required"); return -22;
    }
    return launch_norm<false>(h, delta0_bf16, gamma, beta, y_bf16, nullptr, M, D, eps, (hipStream_t)stream, "vly_add2_layernorm",
                              delta1_bf16);
}

// W [N,K] row-major (row stride ldw) -> [K/64][ceil(N/64)][64][64] blocks, rows >= N zero: one 16-byte chunk per thread
__global__ void __launch_bounds__(256) pack_weight_kernel(const uint16_t* __restrict__ W, uint16_t* __restrict__ P, int N, int K,
                                                         int ldw, int nblk, size_t chunks) {
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= chunks) return;
    const int c8 = (int)(c & 7), r = (int)((c >> 3) & 63);
    const size_t blk = c >> 9;
    const int nb = (int)(blk % nblk), kt = (int)(blk / nblk);
    const int n = nb * 64 + r;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (n < N) v = *(const u32x4*)(W + (size_t)n * ldw + kt * 64 + c8 * 8);
    *(u32x4*)(P + c * 8) = v;
}

extern "C" int vly_pack_weight_bf16(const void* W, void* packed, int N, int K, int ldw, void* stream) {
    if (N <= 0 || K <= 0 || K % 64 || ldw < K || ldw % 8 || ((uintptr_t)W & 15) || ((uintptr_t)packed & 15) || W == packed) {
        vly_set_error("vly_pack_weight_bf16: bad args N=%d K=%d ldw=%d", N, K, ldw);
        return -22;
    }
    const int nblk = (N + 63) / 64;
    const size_t chunks = (size_t)nblk * (K / 64) * 512;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)W, (uint16_t*)packed, N, K, ldw, nblk, chunks);
    return vly_check_launch("vly_pack_weight_bf16");
}

extern "C" int vly_argmax(const float* x, int32_t* idx, int M, int N, int ld, void* stream) {
    if (M <= 0 || N <= 0 || ld < N) { vly_set_error("vly_argmax: bad args"); return -22; }
    hipLaunchKernelGGL(argmax_kernel, dim3(M), dim3(1024), 0, (hipStream_t)stream, x, idx, N, ld);
    return vly_check_launch("vly_argmax");
}

extern "C" int vly_cast_f32_bf16(const float* x, void* y, long n, void* stream) {
    if (n <= 0 || n % 8 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) { vly_set_error("vly_cast_f32_bf16: bad args n=%ld", n); return -22; }
    const long n8 = n / 8;
    hipLaunchKernelGGL(cast_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)y, n8);
    return vly_check_launch("vly_cast_f32_bf16");
}

__global__ void incr_kernel(int32_t* p, int n, int delta) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) p[i] += delta;
}

extern "C" int vly_incr_i32(int32_t* p, int n, int delta, void* stream) {
    if (!p || n <= 0) { vly_set_error("vly_incr_i32: bad args"); return -22; }
    hipLaunchKernelGGL(incr_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, p, n, delta);
    return vly_check_launch("vly_incr_i32");
}
