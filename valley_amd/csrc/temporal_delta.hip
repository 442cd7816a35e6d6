// Kernels of Valley's v3 "temporal transformer delta" pooling (valley/model/valley_model.py:123-133):
//     x[p, t, :]  = patch_feature[t, p, :] + position_matrix[t, :]          (256 sequences of length T)
//     delta[p, :] = TransformerEncoderLayer(d_model=H, nhead=8, post-LN, ReLU FFN 2048)(x)[p, T-1, :]
//     pooled[p,:] = delta[p, :] + mean_t patch_feature[t, p, :]
// Only the LAST time step of the encoder output is used, so queries, the FFN and both LayerNorms run
// on 256 rows per clip, and only K/V need all 256*T rows.  The GEMMs go through vly_gemm_bf16; this
// file holds the glue that is not a GEMM: the gather/transpose/position-add that builds the GEMM
// inputs, the (tiny: T keys, head_dim H/8) attention, and the final add + CLS pick.
// All HBM-bound; feats are the PROJECTED features fp32 [B, T, 257, H].
#include "common.hpp"
#include "../../include/valley_hip.h"

namespace {

// storage type of the GEMM operands these kernels write / read: the library's 16-bit type, or fp32 (the *_f32 entry points
// of the fp32 "precise" mode, round 4: the same kernels with nothing rounded)
VLY_DEVICE void st4(uint16_t* p, float4 x) {
    u32x2 pk;
    pk[0] = pack_h2(x.x, x.y);
    pk[1] = pack_h2(x.z, x.w);
    *(u32x2*)p = pk;
}
VLY_DEVICE void st4(float* p, float4 x) { *(float4*)p = x; }
VLY_DEVICE float ld1(const uint16_t* p) { return h2f(*p); }
VLY_DEVICE float ld1(const float* p) { return *p; }
VLY_DEVICE void st1(uint16_t* p, float x) { *p = f2h(x); }
VLY_DEVICE void st1(float* p, float x) { *p = x; }

// thread per (clip b, patch p, 4 columns): walks the T frames once.
template <typename ST>
__global__ void __launch_bounds__(256) delta_prep_kernel(const float* __restrict__ feats, const float* __restrict__ pos,
                                                         ST* __restrict__ x_all, ST* __restrict__ x_last16,
                                                         float* __restrict__ x_last32, float* __restrict__ mean, int B, int T, int H) {
    const int hv = H >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * 256 * hv) return;
    const int c = (int)(idx % hv);
    const long bp = idx / hv;                                   // b*256 + p
    const int p = (int)(bp & 255), b = (int)(bp >> 8);
    const float* fb = feats + (size_t)b * T * 257 * H;
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < T; ++t) {
        const float4 a = ((const float4*)(fb + ((size_t)t * 257 + 1 + p) * H))[c];
        const float4 q = ((const float4*)(pos + (size_t)t * H))[c];
        m.x += a.x; m.y += a.y; m.z += a.z; m.w += a.w;
        const float4 x = make_float4(a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w);
        st4(x_all + ((size_t)bp * T + t) * H + 4 * c, x);
        if (t == T - 1) {
            if (x_last16) st4(x_last16 + (size_t)bp * H + 4 * c, x);
            ((float4*)(x_last32 + (size_t)bp * H))[c] = x;
        }
    }
    const float inv = 1.f / (float)T;
    ((float4*)(mean + (size_t)bp * H))[c] = make_float4(m.x * inv, m.y * inv, m.z * inv, m.w * inv);
}

// one wave per (sequence, head): q [hd] against T keys, softmax over T, weighted sum of T values.
// kv bf16 [nseq*T, 2H] (k | v), q bf16 [nseq, H], out bf16 [nseq, H].  hd = H/8 <= 1024, T <= 32.
template <int EPL, typename ST>                                  // elements per lane (hd <= 64*EPL)
__global__ void __launch_bounds__(256) delta_attn_kernel(const ST* __restrict__ q, const ST* __restrict__ kv,
                                                         ST* __restrict__ out, int nseq, int T, int H, int nhead) {
    const int lane = threadIdx.x & 63;
    const long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= (long)nseq * nhead) return;
    const int h = (int)(unit % nhead);
    const long s = unit / nhead;
    const int hd = H / nhead;
    const float scale = rsqrtf((float)hd);
    float qv[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int d = lane + 64 * e;
        qv[e] = d < hd ? ld1(q + (size_t)s * H + h * hd + d) * scale : 0.f;
    }
    float sc[32];
    float mx = -1e30f;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        sc[t] = -1e30f;
        if (t < T) {
            const ST* kr = kv + ((size_t)s * T + t) * 2 * H + h * hd;
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int d = lane + 64 * e;
                if (d < hd) a = fmaf(qv[e], ld1(kr + d), a);
            }
            a = wave_sum(a);
            sc[t] = a;
            mx = fmaxf(mx, a);
        }
    }
    float den = 0.f;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const float p = t < T ? __expf(sc[t] - mx) : 0.f;
        sc[t] = p;
        den += p;
    }
    float o[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] = 0.f;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        if (t < T) {
            const ST* vr = kv + ((size_t)s * T + t) * 2 * H + H + h * hd;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int d = lane + 64 * e;
                if (d < hd) o[e] = fmaf(sc[t], ld1(vr + d), o[e]);
            }
        }
    }
    const float inv = 1.f / den;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int d = lane + 64 * e;
        if (d < hd) st1(out + (size_t)s * H + h * hd + d, o[e] * inv);
    }
}

// out bf16 [B, 256+T, H]: rows < 256 = delta + mean, rows >= 256 = CLS token of frame r-256.
template <typename ST>
__global__ void __launch_bounds__(256) delta_finish_kernel(const float* __restrict__ delta, const float* __restrict__ mean,
                                                           const float* __restrict__ feats, ST* __restrict__ out,
                                                           int B, int T, int H) {
    const int hv = H >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * (256 + T) * hv;
    if (idx >= total) return;
    const int c = (int)(idx % hv);
    const long rr = idx / hv;
    const int r = (int)(rr % (256 + T)), b = (int)(rr / (256 + T));
    float4 o;
    if (r < 256) {
        const float4 d = ((const float4*)(delta + ((size_t)b * 256 + r) * H))[c];
        const float4 m = ((const float4*)(mean + ((size_t)b * 256 + r) * H))[c];
        o = make_float4(d.x + m.x, d.y + m.y, d.z + m.z, d.w + m.w);
    } else {
        o = ((const float4*)(feats + ((size_t)b * T + (r - 256)) * 257 * H))[c];
    }
    st4(out + ((size_t)b * (256 + T) + r) * H + 4 * c, o);
}

}  // namespace

template <typename ST>
static int delta_prep_launch(const char* name, const float* feats, const float* pos, ST* x_all, ST* x_last, float* x_last_f32,
                             float* mean, int B, int T, int H, void* stream) {
    if (B <= 0 || T <= 0 || H <= 0 || H % 4) { vly_set_error("%s: bad args B=%d T=%d H=%d", name, B, T, H); return -22; }
    const long n = (long)B * 256 * (H / 4);
    hipLaunchKernelGGL(delta_prep_kernel<ST>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, feats, pos,
                       x_all, x_last, x_last_f32, mean, B, T, H);
    return vly_check_launch(name);
}

extern "C" int vly_delta_prep(const float* feats, const float* pos, void* x_all, void* x_last_bf16, float* x_last_f32,
                              float* mean, int B, int T, int H, void* stream) {
    return delta_prep_launch("vly_delta_prep", feats, pos, (uint16_t*)x_all, (uint16_t*)x_last_bf16, x_last_f32, mean, B, T, H, stream);
}

extern "C" int vly_delta_prep_f32(const float* feats, const float* pos, float* x_all, float* x_last, float* mean, int B, int T,
                                  int H, void* stream) {
    return delta_prep_launch<float>("vly_delta_prep_f32", feats, pos, x_all, nullptr, x_last, mean, B, T, H, stream);
}

template <typename ST>
static int delta_attention_launch(const char* name, const ST* q, const ST* kv, ST* out, int nseq, int T, int H, int nhead,
                                  void* stream) {
    if (nseq <= 0 || T <= 0 || T > 32 || H <= 0 || nhead <= 0 || H % nhead || H / nhead > 1024) {
        vly_set_error("%s: bad args nseq=%d T=%d H=%d nhead=%d (T <= 32, head_dim <= 1024)", name, nseq, T, H, nhead);
        return -22;
    }
    const long units = (long)nseq * nhead;
    dim3 grid((unsigned)((units + 3) / 4)), block(256);
    const int epl = (H / nhead + 63) / 64;
#define VLY_DA(E) hipLaunchKernelGGL((delta_attn_kernel<E, ST>), grid, block, 0, (hipStream_t)stream, q, kv, out, nseq, T, H, nhead)
    if (epl <= 1) VLY_DA(1);
    else if (epl <= 4) VLY_DA(4);
    else if (epl <= 8) VLY_DA(8);
    else if (epl <= 10) VLY_DA(10);
    else VLY_DA(16);
#undef VLY_DA
    return vly_check_launch(name);
}

extern "C" int vly_delta_attention(const void* q, const void* kv, void* out, int nseq, int T, int H, int nhead, void* stream) {
    return delta_attention_launch("vly_delta_attention", (const uint16_t*)q, (const uint16_t*)kv, (uint16_t*)out, nseq, T, H, nhead,
                                  stream);
}

extern "C" int vly_delta_attention_f32(const float* q, const float* kv, float* out, int nseq, int T, int H, int nhead, void* stream) {
    return delta_attention_launch("vly_delta_attention_f32", q, kv, out, nseq, T, H, nhead, stream);
}

template <typename ST>
static int delta_finish_launch(const char* name, const float* delta, const float* mean, const float* feats, ST* out, int B, int T,
                               int H, void* stream) {
    if (B <= 0 || T <= 0 || H <= 0 || H % 4) { vly_set_error("%s: bad args", name); return -22; }
    const long n = (long)B * (256 + T) * (H / 4);
    hipLaunchKernelGGL(delta_finish_kernel<ST>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, delta, mean,
                       feats, out, B, T, H);
    return vly_check_launch(name);
}

extern "C" int vly_delta_finish(const float* delta, const float* mean, const float* feats, void* out, int B, int T, int H,
                                void* stream) {
    return delta_finish_launch("vly_delta_finish", delta, mean, feats, (uint16_t*)out, B, T, H, stream);
}

extern "C" int vly_delta_finish_f32(const float* delta, const float* mean, const float* feats, float* out, int B, int T, int H,
                                    void* stream) {
    return delta_finish_launch("vly_delta_finish_f32", delta, mean, feats, out, B, T, H, stream);
}
