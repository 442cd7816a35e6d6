// fp32 "precise" path (VALLEY_PRECISION=fp32): every tensor on the path is fp32 and every contraction runs on the exact
// f32-input matrix instruction v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md §Matrix cores: bitwise an fmaf chain, 157 TF
// peak = 1/16 of the bf16 rate).  Purpose: demonstrate BASELINE.json's "logits within 1e-3 of reference" against the
// fp32 reference, which bf16 operands cannot reach (one bf16 rounding is 4e-3 relative) — SURVEY.md §7 "hard parts",
// §8c tolerances.  These kernels are written for clarity and exactness, not for the roofline; the production path is
// the bf16 pipeline (gemm_bf16.hip, attention.hip, ...).
//
// Replaces, at fp32: K1-K18 of SURVEY.md §2.2 —
//   vly_gemm_f32          nn.Linear / conv-as-GEMM (+bias, quick_gelu | SwiGLU | ReLU, +residual)   hf:clip 148-154,293-350; hf:llama 160-173,230-241
//   vly_attention_f32     softmax(QK^T * hd^-0.5 [+ causal/padding mask]) V, fp32 softmax            hf:clip 259-277; hf:llama 191-213
//   vly_norm_f32          LayerNorm / RMSNorm                                                         hf:clip 605,642; hf:llama 51-67
//   vly_rope_kv_f32       rotate-half RoPE of q,k + KV-cache append                                  hf:llama 127-157
//   vly_patchify_f32      im2col of the 14x14 patch conv                                              hf:clip 148-154
//   vly_pool_tokens_f32   temporal mean / max / importance pooling + CLS pick                         valley_model.py:206-215,113-121
//   vly_embed_splice_f32  embedding gather + visual-token splice by row map                           valley_model.py:160,195-247
#include <type_traits>
#include <cstdlib>
#include <cstring>
#include "common.hpp"
#include "../../include/valley_hip.h"

namespace {

typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

// ------------------------------------------------------------------------------------------------------------------
// GEMM: C[M,N] = epi(A[M,K] . W[N,K]^T + bias) + residual, all fp32.  64x64 tile per 256-thread workgroup, each of the
// four waves owns 32x32 of it as ONE 32x32x2 accumulator block; K advances 16 per LDS tile (8 MFMAs per wave and tile).
// The W fragment is the MFMA's first operand (as in the bf16 kernels), so a lane ends up with 4 consecutive n for one m:
// register r of the 16 holds C[m = lane & 31][n = 8*(r/4) + 4*(lane >> 5) + r % 4].
// ------------------------------------------------------------------------------------------------------------------
constexpr int GB = 64, GK = 16, GLD = GK + 1;        // +1 float: ds_read_b32 of a column is conflict-free

template <int EPI>
__global__ void __launch_bounds__(256) gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                       const float* __restrict__ bias, const float* __restrict__ R,
                                                       float* __restrict__ C, int M, int N, int K, int lda, int ldw, int ldc,
                                                       int ldr) {
    __shared__ float sA[GB * GLD], sW[GB * GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * GB, n0 = blockIdx.x * GB;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int l31 = lane & 31, hk = lane >> 5;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // staging: thread -> (row = tid / 4, 4 consecutive k); rows past the edge re-read the last valid row (masked at the store)
    const int srow = tid >> 2, sk = (tid & 3) * 4;
    const float* ga = A + (size_t)min(m0 + srow, M - 1) * lda + sk;
    const float* gw = W + (size_t)min(n0 + srow, N - 1) * ldw + sk;
    for (int k0 = 0; k0 < K; k0 += GK) {
        const f32x4 va = *(const f32x4*)(ga + k0), vw = *(const f32x4*)(gw + k0);
        __syncthreads();                                  // everyone finished reading the previous tile
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sA[srow * GLD + sk + i] = va[i];
            sW[srow * GLD + sk + i] = vw[i];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2) {
            const float w = sW[(wn + l31) * GLD + kk + hk];
            const float a = sA[(wm + l31) * GLD + kk + hk];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w, a, acc, 0, 0, 0);
        }
    }
    const int m = m0 + wm + l31;
    if (m >= M) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn + 8 * q + 4 * hk;
        if (n >= N) continue;                              // N % 4 == 0: a group of 4 is inside or outside as a whole
        f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        if (bias) v += *(const f32x4*)(bias + n);
        if constexpr (EPI == VLY_EPI_QUICK_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.f + expf(-1.702f * v[r]));
        }
        if constexpr (EPI == VLY_EPI_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if constexpr (EPI == VLY_EPI_SWIGLU) {             // rows of W interleaved (gate, up): out column = n / 2
            const float o0 = v[0] / (1.f + expf(-v[0])) * v[1];
            const float o1 = v[2] / (1.f + expf(-v[2])) * v[3];
            *(float2*)(C + (size_t)m * ldc + (n >> 1)) = make_float2(o0, o1);
        } else {
            if (R) v += *(const f32x4*)(R + (size_t)m * ldr + n);
            *(f32x4*)(C + (size_t)m * ldc + n) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Attention: one wave per (query row, head, batch).  Keys are visited in chunks of 64 (lane j scores key j of the
// chunk against the query held in LDS), online softmax in fp32, then every lane owns output dims {lane, lane + 64}.
// ------------------------------------------------------------------------------------------------------------------
template <int HD>
__global__ void __launch_bounds__(64) attention_f32_kernel(const float* __restrict__ Q, long q_bs, int q_rs,
                                                           const float* __restrict__ Kp, const float* __restrict__ Vp,
                                                           long kv_bs, long kv_hs, int kv_rs, const uint8_t* __restrict__ key_valid,
                                                           int kv_valid_stride, float* __restrict__ O, long o_bs, int o_rs,
                                                           int n_q, int n_kv, int causal, int past, float scale) {
    __shared__ float sq[HD];
    const int lane = threadIdx.x, qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const float* q = Q + b * q_bs + (size_t)qi * q_rs + h * HD;
    for (int d = lane; d < HD; d += 64) sq[d] = q[d];
    __syncthreads();
    const float* kb = Kp + b * kv_bs + h * kv_hs;
    const float* vb = Vp + b * kv_bs + h * kv_hs;
    const uint8_t* valid = key_valid ? key_valid + (size_t)b * kv_valid_stride : nullptr;
    const int last = causal ? min(n_kv - 1, qi + past) : n_kv - 1;        // keys 0..last are visible
    float m_run = -INFINITY, l_run = 0.f, o0 = 0.f, o1 = 0.f;
    for (int j0 = 0; j0 <= last; j0 += 64) {
        const int j = j0 + lane;
        float s = -INFINITY;
        if (j <= last && (!valid || valid[j])) {
            const float* kr = kb + (size_t)j * kv_rs;
            float acc = 0.f;
#pragma unroll 8
            for (int d = 0; d < HD; d += 4) {
                const f32x4 kv = *(const f32x4*)(kr + d);
                acc = fmaf(sq[d], kv[0], acc);
                acc = fmaf(sq[d + 1], kv[1], acc);
                acc = fmaf(sq[d + 2], kv[2], acc);
                acc = fmaf(sq[d + 3], kv[3], acc);
            }
            s = acc * scale;
        }
        const float m_new = fmaxf(m_run, wave_max(s));
        if (m_new == -INFINITY) continue;                  // nothing visible yet (left padding)
        const float p = s == -INFINITY ? 0.f : expf(s - m_new);
        const float corr = m_run == -INFINITY ? 0.f : expf(m_run - m_new);
        l_run = l_run * corr + wave_sum(p);
        o0 *= corr;
        o1 *= corr;
        const int nj = min(64, last + 1 - j0);
        for (int t = 0; t < nj; ++t) {
            const float pt = __shfl(p, t, 64);
            const float* vr = vb + (size_t)(j0 + t) * kv_rs;
            o0 = fmaf(pt, vr[lane], o0);
            if (HD > 64) o1 = fmaf(pt, vr[lane + 64], o1);
        }
        m_run = m_new;
    }
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;     // a fully masked (padded) query row yields zeros
    float* o = O + b * o_bs + (size_t)qi * o_rs + h * HD;
    o[lane] = o0 * inv;
    if (HD > 64) o[lane + 64] = o1 * inv;
}

// ------------------------------------------------------------------------------------------------------------------
// Attention on the exact f32-input MFMA (round 6: the one-wave-per-query kernel above was 48 % of a step of the split-operand
// engine).  A workgroup = 64 queries of one (batch, head), 16 per wave; keys in tiles of 64 staged in LDS as fp32.
//   S^T = K Q^T  (v_mfma_f32_16x16x4_f32: A = K[key r][k], B = Q^T[k][query r]) so that lane (g, r) ends up with the scores of ONE
//   query r for keys 16 blk + 4 g + e: the softmax statistics of a query live in 4 lanes (xor 16, xor 32);
//   O^T = V^T P^T (A = V^T[d r][key], B = P^T[key][query r]): the lane's accumulators again belong to query r — rescaling is per lane.
// The contraction index is permuted so that a lane's operands are contiguous: in K step ks lane group g takes d = g HD/4 + ks (QK^T)
// and key = 16 g + ks (PV); P goes through a per-wave LDS strip [query][key] to change owner.  Every product and sum is an fp32
// fma (the MFMA is bit-for-bit an fmaf chain); the order of summation differs from the scalar kernel, nothing else.
// KT keys per LDS tile: 32 (round 6).  With 64, head_dim 128 needed 85 KB — ONE workgroup per CU, one wave per SIMD, every tile's memory
// round trip and both barriers exposed (0.35 ms per 13B layer for 68 us of MFMA; 43 KB and 164 registers now: three per CU, prefill on the
// split-operand engine 173 -> 166.6 ms) — and head_dim 64 52 KB / 152 registers (three per CU; 27 KB / 112 now: four, and 257 keys pad to 288
// instead of 320: ViT pass 74.2 -> 71.7 ms).  profiles/r06/r06_x3_attn_kt32_ab.txt, r06_x3_attn_kt32_hd64_ab.txt.
template <int HD, int KT>
__global__ void __launch_bounds__(256) attention_f32_mfma_kernel(const float* __restrict__ Q, long q_bs, int q_rs, const float* __restrict__ Kp,
                                                                 const float* __restrict__ Vp, long kv_bs, long kv_hs, int kv_rs,
                                                                 const uint8_t* __restrict__ key_valid, int kv_valid_stride,
                                                                 float* __restrict__ O, long o_bs, int o_rs, int n_q, int n_kv, int causal,
                                                                 int past, float scale) {
    constexpr int KS = HD / 4, LDK = HD + 4, LDP = KT + 4;               // K steps of QK^T; padded LDS rows (floats)
    constexpr int NB = KT / 16, GS = KT / 4;                             // 16-key blocks per tile; keys per lane group in the second product
    __shared__ __attribute__((aligned(16))) float sK[KT * LDK], sV[KT * LDK], sP[4][16 * LDP];
    __shared__ __attribute__((aligned(16))) int sOk[KT];                 // key k0 + row exists and is valid (branch-free masking below)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z, qb0 = blockIdx.x * 64, qi = qb0 + wave * 16 + r;
    const bool q_ok = qi < n_q;
    float q[KS];
    {
        const float* qp = Q + b * q_bs + (size_t)min(qi, n_q - 1) * q_rs + h * HD + g * KS;
#pragma unroll
        for (int k = 0; k < KS; k += 4) {
            const f32x4 t = *(const f32x4*)(qp + k);
            q[k] = t[0] * scale; q[k + 1] = t[1] * scale; q[k + 2] = t[2] * scale; q[k + 3] = t[3] * scale;
        }
    }
    const float* kb = Kp + b * kv_bs + h * kv_hs;
    const float* vb = Vp + b * kv_bs + h * kv_hs;
    const uint8_t* valid = key_valid ? key_valid + (size_t)b * kv_valid_stride : nullptr;
    const int my_last = causal ? min(n_kv - 1, qi + past) : n_kv - 1;                       // keys 0 .. my_last are visible to this query
    const int blk_last = causal ? min(n_kv - 1, min(qb0 + 63, n_q - 1) + past) : n_kv - 1;  // ... to any query of the workgroup
    // ... to any query of this WAVE, and whether it has a query at all: 16-key blocks (first product) and key quadruples (second product)
    // past it carry probabilities that are exactly zero — their MFMAs are skipped, wave-uniformly (ViT: 257 = 4 x 64 + 1 keys and queries, so
    // the fifth tile is one key and the fifth query block one row: 17 % + 15 % of the kernel's MFMAs; causal: the diagonal tile's upper half)
    const bool wave_active = qb0 + wave * 16 < n_q;
    const int wave_last = causal ? min(n_kv - 1, min(qb0 + wave * 16 + 15, n_q - 1) + past) : n_kv - 1;
    f32x4 o[HD / 16];
#pragma unroll
    for (int d = 0; d < HD / 16; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    for (int k0 = 0; k0 <= blk_last; k0 += KT) {
        __syncthreads();                                                  // the previous tile's readers are done
        for (int s = tid; s < KT * (HD / 4); s += 256) {                  // K and V rows k0 .. k0 + KT - 1 (zeros past the end)
            const int row = s / (HD / 4), c4 = (s % (HD / 4)) * 4;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
            if (k0 + row < n_kv) {
                kv = *(const f32x4*)(kb + (size_t)(k0 + row) * kv_rs + c4);
                vv = *(const f32x4*)(vb + (size_t)(k0 + row) * kv_rs + c4);
            }
            *(f32x4*)(sK + row * LDK + c4) = kv;
            *(f32x4*)(sV + row * LDK + c4) = vv;
        }
        if (tid < KT) sOk[tid] = (k0 + tid < n_kv && (!valid || valid[min(k0 + tid, n_kv - 1)])) ? 1 : 0;
        __syncthreads();
        if (!wave_active || k0 > wave_last) continue;                     // (the loads and the barriers above are the whole workgroup's)
        // a tile this wave sees whole runs the branch-free body; only a tile cut by wave_last tests its blocks / key quadruples
        const auto tile_body = [&](auto part_c) {
            constexpr bool PART = decltype(part_c)::value;
            // ---- S^T: NB key blocks x KS K steps
            f32x4 sc[NB];
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                sc[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (PART && k0 + blk * 16 > wave_last) continue;              // every score of the block is masked below
                const float* kr = sK + (blk * 16 + r) * LDK + g * KS;
#pragma unroll
                for (int k = 0; k < KS; k += 4) {
                    const f32x4 a = *(const f32x4*)(kr + k);
                    sc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], q[k], sc[blk], 0, 0, 0);
                    sc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], q[k + 1], sc[blk], 0, 0, 0);
                    sc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], q[k + 2], sc[blk], 0, 0, 0);
                    sc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], q[k + 3], sc[blk], 0, 0, 0);
                }
            }
            // ---- mask, online softmax of query r (its KT scores sit in 4 lanes x KT / 4 registers)
            float mx = -INFINITY;
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const int ok4[4] = {sOk[blk * 16 + 4 * g], sOk[blk * 16 + 4 * g + 1], sOk[blk * 16 + 4 * g + 2], sOk[blk * 16 + 4 * g + 3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = k0 + blk * 16 + 4 * g + e;
                    const int vis = (q_ok ? 1 : 0) & (key <= my_last ? 1 : 0) & ok4[e];
                    sc[blk][e] = vis ? sc[blk][e] : -INFINITY;
                    mx = fmaxf(mx, sc[blk][e]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float corr = (m_run == -INFINITY || m_new == -INFINITY) ? 0.f : expf(m_run - m_new);
            float ps = 0.f;
            float* pw = &sP[wave][r * LDP + 4 * g];
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                f32x4 p;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p[e] = sc[blk][e] == -INFINITY ? 0.f : expf(sc[blk][e] - m_new);
                    ps += p[e];
                }
                *(f32x4*)(pw + blk * 16) = p;                                 // P[query r][key 16 blk + 4 g + e]
            }
            ps += __shfl_xor(ps, 16, 64);
            ps += __shfl_xor(ps, 32, 64);
            l_run = l_run * corr + ps;
            if (m_new != -INFINITY) m_run = m_new;
#pragma unroll
            for (int d = 0; d < HD / 16; ++d) o[d] *= corr;
            // ---- O^T += V^T P^T: K step ks of lane group g is key GS g + ks
            float pb[GS];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");           // P: written and read by the same wave, other lanes
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {
                const float* pr = &sP[wave][r * LDP + GS * g];
#pragma unroll
                for (int k = 0; k < GS; k += 4) {
                    const f32x4 t = *(const f32x4*)(pr + k);
                    pb[k] = t[0]; pb[k + 1] = t[1]; pb[k + 2] = t[2]; pb[k + 3] = t[3];
                }
            }
#pragma unroll
            for (int d = 0; d < HD / 16; ++d) {
                const float* vr = sV + (GS * g) * LDK + d * 16 + r;
#pragma unroll
                for (int k = 0; k < GS; ++k)
                    if (!PART || k0 + k <= wave_last) o[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[k * LDK], pb[k], o[d], 0, 0, 0);   // keys k, GS + k, 2 GS + k, 3 GS + k
            }
    
        };
        if (k0 + KT - 1 <= wave_last) tile_body(std::false_type{});
        else tile_body(std::true_type{});
    }
    if (q_ok) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;               // a fully masked (padded) query row yields zeros
        float* op = O + b * o_bs + (size_t)qi * o_rs + h * HD + 4 * g;
#pragma unroll
        for (int d = 0; d < HD / 16; ++d) *(f32x4*)(op + d * 16) = o[d] * inv;
    }
}

// LayerNorm (beta != null) / RMSNorm (rms != 0): wave per row, fp32 statistics over the whole row.
__global__ void __launch_bounds__(256) norm_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ y, int M, int D,
                                                       float eps, int rms) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += xr[d];
    const float mean = rms ? 0.f : wave_sum(s) / D;
    float v = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float c = xr[d] - mean;
        v = fmaf(c, c, v);
    }
    const float rstd = rsqrtf(wave_sum(v) / D + eps);
    float* yr = y + (size_t)row * D;
    for (int d = lane; d < D; d += 64) {
        const float n = (xr[d] - mean) * rstd;
        yr[d] = rms ? gamma[d] * n : fmaf(n, gamma[d], beta[d]);
    }
}

// ---- the two norms, four elements per lane (round 6; D % 4 == 0 and 16-byte aligned rows — the scalar kernels around them serve the rest).
// norm_split3_kernel moved 337 MB per ViT call in 103 us (3.3 TB/s): 4-byte loads, three 2-byte stores per element.  Here a lane reads 16
// bytes and writes three 8-byte pieces: the statistics are ONE routine for both kernels (lane l sums the float4 l, l + 64, ... in order,
// (a0 + a1) + (a2 + a3) each; the butterfly of wave_sum after), so norm_split3 stays value for value norm + split3
// (tests/test_r6_gpu.py::test_norm_split3_equals_norm_then_split).
VLY_DEVICE void norm_row_stats4(const float* __restrict__ xr, int D, int lane, int rms, float eps, float& mean, float& rstd) {
    const int nv = D >> 2;
    float s = 0.f;
    if (!rms)
        for (int v = lane; v < nv; v += 64) {
            const f32x4 a = *(const f32x4*)(xr + 4 * v);
            s += (a[0] + a[1]) + (a[2] + a[3]);
        }
    mean = rms ? 0.f : wave_sum(s) / D;
    float q = 0.f;
    for (int v = lane; v < nv; v += 64) {
        const f32x4 a = *(const f32x4*)(xr + 4 * v);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float c = a[r] - mean;
            q = fmaf(c, c, q);
        }
    }
    rstd = rsqrtf(wave_sum(q) / D + eps);
}
VLY_DEVICE f32x4 norm_apply4(const f32x4& a, const f32x4& gm, const f32x4& bt, float mean, float rstd, int rms) {
    f32x4 y;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float n = (a[r] - mean) * rstd;
        y[r] = rms ? gm[r] * n : fmaf(n, gm[r], bt[r]);
    }
    return y;
}
__global__ void __launch_bounds__(256) norm_f32_vec_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ y, int M, int D, float eps,
                                                           int rms) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * D;
    float mean, rstd;
    norm_row_stats4(xr, D, lane, rms, eps, mean, rstd);
    float* yr = y + (size_t)row * D;
    for (int v = lane; v < (D >> 2); v += 64) {
        const f32x4 bt = rms ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(beta + 4 * v);
        *(f32x4*)(yr + 4 * v) = norm_apply4(*(const f32x4*)(xr + 4 * v), *(const f32x4*)(gamma + 4 * v), bt, mean, rstd, rms);
    }
}
__global__ void __launch_bounds__(256) norm_split3_vec_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, uint16_t* __restrict__ out, int M, int D, int Kp,
                                                              float eps, int rms) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * D;
    float mean, rstd;
    norm_row_stats4(xr, D, lane, rms, eps, mean, rstd);
    uint16_t* orow = out + (size_t)row * (3 * Kp);
    for (int v = lane; v < (Kp >> 2); v += 64) {
        f32x4 yv = {0.f, 0.f, 0.f, 0.f};
        if (4 * v < D) {
            const f32x4 bt = rms ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(beta + 4 * v);
            yv = norm_apply4(*(const f32x4*)(xr + 4 * v), *(const f32x4*)(gamma + 4 * v), bt, mean, rstd, rms);
        }
        uint16_t hi[4], lo[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hi[r] = f2h(yv[r]);
            lo[r] = f2h(yv[r] - h2f(hi[r]));
        }
        const u32x2 H = {(uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16)};
        const u32x2 L = {(uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16)};
        *(u32x2*)(orow + 4 * v) = H;
        *(u32x2*)(orow + Kp + 4 * v) = H;
        *(u32x2*)(orow + 2 * Kp + 4 * v) = L;
    }
}

// The same norm writing the split-operand image [hi | hi | lo] of its result straight away (x3 GEMMs, round 6): the fp32 result never
// goes to memory — 4 B read + 6 B written per element instead of 4 + 4 (norm) and 4 + 6 (vly_split3_f32).  Same arithmetic as
// norm_f32_kernel + split3_kernel<NONE>, value for value.
__global__ void __launch_bounds__(256) norm_split3_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, uint16_t* __restrict__ out, int M, int D, int Kp,
                                                          float eps, int rms) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += xr[d];
    const float mean = rms ? 0.f : wave_sum(s) / D;
    float v = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float c = xr[d] - mean;
        v = fmaf(c, c, v);
    }
    const float rstd = rsqrtf(wave_sum(v) / D + eps);
    uint16_t* orow = out + (size_t)row * (3 * Kp);
    for (int d = lane; d < Kp; d += 64) {
        float y = 0.f;
        if (d < D) {
            const float n = (xr[d] - mean) * rstd;
            y = rms ? gamma[d] * n : fmaf(n, gamma[d], beta[d]);
        }
        const uint16_t hi = f2h(y), lo = f2h(y - h2f(hi));
        orow[d] = hi;
        orow[Kp + d] = hi;
        orow[2 * Kp + d] = lo;
    }
}

// rotate-half RoPE of q (in place) and k (into the cache) + v append; one thread per (token, head, d < 64)
__global__ void __launch_bounds__(256) rope_kv_f32_kernel(float* __restrict__ qkv, float* __restrict__ kc, float* __restrict__ vc,
                                                          const float* __restrict__ cs, const float* __restrict__ sn, int B, int S,
                                                          int heads, int past, int ctx_max) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int d = (int)(i & 63);
    const long u = i >> 6;
    if (u >= (long)B * S * heads) return;
    const int h = (int)(u % heads);
    const long tok = u / heads;
    const int s = (int)(tok % S), b = (int)(tok / S);
    const int pos = past + s;
    const float c = cs[pos * 64 + d], sv = sn[pos * 64 + d];
    const int H = heads * 128;
    float* row = qkv + (size_t)tok * 3 * H;
    float* q = row + h * 128;
    const float q0 = q[d], q1 = q[d + 64];
    q[d] = rope_rot(q0, q1, c, sv, -1.f);
    q[d + 64] = rope_rot(q1, q0, c, sv, 1.f);
    const float* k = row + H + h * 128;
    const float* v = row + 2 * H + h * 128;
    const size_t o = (((size_t)b * heads + h) * ctx_max + pos) * 128;
    kc[o + d] = rope_rot(k[d], k[d + 64], c, sv, -1.f);
    kc[o + d + 64] = rope_rot(k[d + 64], k[d], c, sv, 1.f);
    vc[o + d] = v[d];
    vc[o + d + 64] = v[d + 64];
}

// im2col of Conv2d(3 -> 1024, k = 14, s = 14): [F,3,224,224] -> [F*256, KP], column = c*196 + ky*14 + kx, zero padded to KP
__global__ void __launch_bounds__(256) patchify_f32_kernel(const float* __restrict__ img, float* __restrict__ out, long total, int KP) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int col = (int)(i % KP);
    const long row = i / KP;
    float v = 0.f;
    if (col < 588) {
        const int c = col / 196, r = col % 196, ky = r / 14, kx = r % 14;
        const int f = (int)(row >> 8), p = (int)(row & 255), py = p >> 4, px = p & 15;
        v = img[(((size_t)f * 3 + c) * 224 + py * 14 + ky) * 224 + px * 14 + kx];
    }
    out[i] = v;
}

// feats [B*T*257, W] -> out [B, 256 + T, W]: rows 0..255 pooled patch tokens, rows 256.. the T CLS tokens
__global__ void __launch_bounds__(256) pool_f32_kernel(const float* __restrict__ feats, float* __restrict__ out, int B, int T, int W,
                                                       int mode, const float* __restrict__ scores) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * (256 + T) * W) return;
    const int w = (int)(i % W);
    const long r = i / W;
    const int row = (int)(r % (256 + T)), b = (int)(r / (256 + T));
    const float* f = feats + (size_t)b * T * 257 * W;
    float v;
    if (row >= 256) {
        v = f[(size_t)(row - 256) * 257 * W + w];
    } else if (mode == VLY_POOL_MAX) {
        v = -INFINITY;
        for (int t = 0; t < T; ++t) v = fmaxf(v, f[((size_t)t * 257 + 1 + row) * W + w]);
    } else if (mode == VLY_POOL_IMPORTANCE) {
        float mx = -INFINITY, den = 0.f;
        for (int t = 0; t < T; ++t) mx = fmaxf(mx, scores[b * T + t]);
        v = 0.f;
        for (int t = 0; t < T; ++t) {
            const float e = expf(scores[b * T + t] - mx);
            den += e;
            v = fmaf(e, f[((size_t)t * 257 + 1 + row) * W + w], v);
        }
        v /= den;
    } else {
        v = 0.f;
        for (int t = 0; t < T; ++t) v += f[((size_t)t * 257 + 1 + row) * W + w];
        v /= T;
    }
    out[i] = v;
}

__global__ void __launch_bounds__(256) embed_splice_f32_kernel(const int32_t* __restrict__ row_map, const float* __restrict__ embed,
                                                               const float* __restrict__ visual, float* __restrict__ out, long total,
                                                               int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int r = row_map[i / H], c = (int)(i % H);
    out[i] = r >= 0 ? embed[(size_t)r * H + c] : visual[(size_t)(-r - 1) * H + c];
}

}  // namespace

// ---- split-operand form of the fp32 GEMMs (round 6, VALLEY_F32_GEMM=x3; valley_amd/ops_f32.py) ---------------------------------------
// An fp32 value x = hi + lo + O(2^-16 x) with hi = rn16(x), lo = rn16(x - hi) (two 16-bit storage values).  A product a w then is
// a_hi w_hi + a_hi w_lo + a_lo w_hi up to 2^-16 relative (the dropped a_lo w_lo and the tails), every partial product exact in fp32.
// Written as ONE contraction over 3 K: A3 = [a_hi | a_hi | a_lo], W3 = [w_hi | w_lo | w_hi], so that the existing 16-bit MFMA kernels
// (vly_gemm_bf16, fp32 accumulation, fp32 output + bias + residual) compute it unchanged at a third of their rate — ~5x the exact
// f32-input MFMA.  This kernel builds either image from fp32 rows: order 0 = [hi | hi | lo] (activations), 1 = [hi | lo | hi]
// (weights); every segment is Kp >= K wide (pad columns zero: the GEMM wants 3 Kp % 64 == 0).  epi applies the producing GEMM's
// activation first, with vly_gemm_f32's own expressions: QUICK_GELU, RELU, or SWIGLU (x holds 2 K interleaved (gate, up) columns).
template <int EPI>
__global__ void __launch_bounds__(256) split3_kernel(const float* __restrict__ x, int ldx, uint16_t* __restrict__ out, int M, int K, int Kp,
                                                      int order) {
    const long total = (long)M * (Kp >> 2);
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int m = (int)(t / (Kp >> 2)), k = (int)(t % (Kp >> 2)) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (k < K) {                                                          // (K % 4 == 0)
        if constexpr (EPI == VLY_EPI_SWIGLU) {
            const f32x4 a = *(const f32x4*)(x + (size_t)m * ldx + 2 * k), b = *(const f32x4*)(x + (size_t)m * ldx + 2 * k + 4);
            v[0] = a[0] / (1.f + expf(-a[0])) * a[1];
            v[1] = a[2] / (1.f + expf(-a[2])) * a[3];
            v[2] = b[0] / (1.f + expf(-b[0])) * b[1];
            v[3] = b[2] / (1.f + expf(-b[2])) * b[3];
        } else {
            const f32x4 a = *(const f32x4*)(x + (size_t)m * ldx + k);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = a[r];
                if constexpr (EPI == VLY_EPI_QUICK_GELU) v[r] = v[r] / (1.f + expf(-1.702f * v[r]));
                if constexpr (EPI == VLY_EPI_RELU) v[r] = fmaxf(v[r], 0.f);
            }
        }
    }
    uint16_t hi[4], lo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        hi[r] = f2h(v[r]);
        lo[r] = f2h(v[r] - h2f(hi[r]));
    }
    const u32x2 H = {(uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16)};
    const u32x2 L = {(uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16)};
    uint16_t* row = out + (size_t)m * (3 * Kp) + k;
    *(u32x2*)row = H;
    *(u32x2*)(row + Kp) = order ? L : H;
    *(u32x2*)(row + 2 * Kp) = order ? H : L;
}

extern "C" int vly_split3_f32(const float* x, int ldx, void* out3, int M, int K, int Kp, int epilogue, int order, void* stream) {
    const int kin = epilogue == VLY_EPI_SWIGLU ? 2 * K : K;
    if (M <= 0 || K <= 0 || K % 4 || Kp < K || Kp % 4 || ldx < kin || ldx % 4 || ((uintptr_t)x & 15) || ((uintptr_t)out3 & 7) || (order & ~1)) {
        vly_set_error("vly_split3_f32: bad args M=%d K=%d Kp=%d ldx=%d order=%d", M, K, Kp, ldx, order);
        return -22;
    }
    const long total = (long)M * (Kp >> 2);
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case VLY_EPI_NONE: hipLaunchKernelGGL(split3_kernel<VLY_EPI_NONE>, grid, block, 0, st, x, ldx, (uint16_t*)out3, M, K, Kp, order); break;
        case VLY_EPI_QUICK_GELU: hipLaunchKernelGGL(split3_kernel<VLY_EPI_QUICK_GELU>, grid, block, 0, st, x, ldx, (uint16_t*)out3, M, K, Kp, order); break;
        case VLY_EPI_RELU: hipLaunchKernelGGL(split3_kernel<VLY_EPI_RELU>, grid, block, 0, st, x, ldx, (uint16_t*)out3, M, K, Kp, order); break;
        case VLY_EPI_SWIGLU: hipLaunchKernelGGL(split3_kernel<VLY_EPI_SWIGLU>, grid, block, 0, st, x, ldx, (uint16_t*)out3, M, K, Kp, order); break;
        default: vly_set_error("vly_split3_f32: bad epilogue %d", epilogue); return -22;
    }
    return vly_check_launch("vly_split3_f32");
}

extern "C" int vly_gemm_f32(const float* A, const float* W, const float* bias, const float* residual, float* C, int M, int N,
                            int K, int lda, int ldw, int ldc, int ldr, int epilogue, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || K % GK || N % 4 || lda % 4 || ldw % 4 || ldc % 2 || ((uintptr_t)A & 15) || ((uintptr_t)W & 15) ||
        ((uintptr_t)C & 7) || (epilogue != VLY_EPI_SWIGLU && (ldc % 4 || ((uintptr_t)C & 15))) ||
        (bias && ((uintptr_t)bias & 15)) || (residual && (ldr % 4 || ((uintptr_t)residual & 15))) ||
        (epilogue == VLY_EPI_SWIGLU && residual)) {
        vly_set_error("vly_gemm_f32: unsupported shape/alignment M=%d N=%d K=%d lda=%d ldw=%d ldc=%d ldr=%d epi=%d", M, N, K, lda, ldw,
                      ldc, ldr, epilogue);
        return -22;
    }
    dim3 grid((N + GB - 1) / GB, (M + GB - 1) / GB), block(256);
    hipStream_t st = (hipStream_t)stream;
#define VLY_F32_LAUNCH(E) hipLaunchKernelGGL((gemm_f32_kernel<E>), grid, block, 0, st, A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr)
    switch (epilogue) {
        case VLY_EPI_NONE: VLY_F32_LAUNCH(VLY_EPI_NONE); break;
        case VLY_EPI_QUICK_GELU: VLY_F32_LAUNCH(VLY_EPI_QUICK_GELU); break;
        case VLY_EPI_SWIGLU: VLY_F32_LAUNCH(VLY_EPI_SWIGLU); break;
        case VLY_EPI_RELU: VLY_F32_LAUNCH(VLY_EPI_RELU); break;
        default: vly_set_error("vly_gemm_f32: bad epilogue %d", epilogue); return -22;
    }
#undef VLY_F32_LAUNCH
    return vly_check_launch("vly_gemm_f32");
}

extern "C" int vly_attention_f32(const float* q, long q_batch_stride, int q_row_stride, const float* k, const float* v,
                                 long kv_batch_stride, long kv_head_stride, int kv_row_stride, const uint8_t* key_valid,
                                 int key_valid_stride, float* out, long out_batch_stride, int out_row_stride, int B, int heads,
                                 int n_q, int n_kv, int head_dim, int causal, int past_len, void* stream) {
    if (B <= 0 || heads <= 0 || n_q <= 0 || n_kv <= 0 || (head_dim != 64 && head_dim != 128) || kv_row_stride % 4 ||
        ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || kv_batch_stride % 4 || kv_head_stride % 4) {
        vly_set_error("vly_attention_f32: bad args B=%d heads=%d n_q=%d n_kv=%d hd=%d", B, heads, n_q, n_kv, head_dim);
        return -22;
    }
    const float scale = 1.f / sqrtf((float)head_dim);
    hipStream_t st = (hipStream_t)stream;
    // 16 queries or more per (batch, head): the tiled kernel on the f32-input MFMA (VLY_ATTN_F32=scalar keeps the one-wave-per-query form)
    static const bool scalar_only = getenv("VLY_ATTN_F32") && !strcmp(getenv("VLY_ATTN_F32"), "scalar");
    if (n_q >= 16 && !scalar_only && q_row_stride % 4 == 0 && q_batch_stride % 4 == 0 && out_row_stride % 4 == 0 && out_batch_stride % 4 == 0 &&
        !((uintptr_t)q & 15) && !((uintptr_t)out & 15)) {
        dim3 grid((n_q + 63) / 64, heads, B), block(256);
        if (head_dim == 64)
            hipLaunchKernelGGL((attention_f32_mfma_kernel<64, 32>), grid, block, 0, st, q, q_batch_stride, q_row_stride, k, v, kv_batch_stride,
                               kv_head_stride, kv_row_stride, key_valid, key_valid_stride, out, out_batch_stride, out_row_stride, n_q, n_kv,
                               causal, past_len, scale);
        else
            hipLaunchKernelGGL((attention_f32_mfma_kernel<128, 32>), grid, block, 0, st, q, q_batch_stride, q_row_stride, k, v, kv_batch_stride,
                               kv_head_stride, kv_row_stride, key_valid, key_valid_stride, out, out_batch_stride, out_row_stride, n_q, n_kv,
                               causal, past_len, scale);
        return vly_check_launch("vly_attention_f32");
    }
    dim3 grid(n_q, heads, B), block(64);
    if (head_dim == 64)
        hipLaunchKernelGGL((attention_f32_kernel<64>), grid, block, 0, st, q, q_batch_stride, q_row_stride, k, v, kv_batch_stride,
                           kv_head_stride, kv_row_stride, key_valid, key_valid_stride, out, out_batch_stride, out_row_stride, n_q, n_kv,
                           causal, past_len, scale);
    else
        hipLaunchKernelGGL((attention_f32_kernel<128>), grid, block, 0, st, q, q_batch_stride, q_row_stride, k, v, kv_batch_stride,
                           kv_head_stride, kv_row_stride, key_valid, key_valid_stride, out, out_batch_stride, out_row_stride, n_q, n_kv,
                           causal, past_len, scale);
    return vly_check_launch("vly_attention_f32");
}

extern "C" int vly_norm_f32(const float* x, const float* gamma, const float* beta, float* y, int M, int D, float eps, int rms,
                            void* stream) {
    if (M <= 0 || D <= 0 || !gamma || (!rms && !beta)) { vly_set_error("vly_norm_f32: bad args M=%d D=%d", M, D); return -22; }
    const bool vec = D % 4 == 0 && !(((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15);
    if (vec) hipLaunchKernelGGL(norm_f32_vec_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, M, D, eps, rms);
    else hipLaunchKernelGGL(norm_f32_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, M, D, eps, rms);
    return vly_check_launch("vly_norm_f32");
}

extern "C" int vly_norm_split3_f32(const float* x, const float* gamma, const float* beta, void* out3, int M, int D, int Kp, float eps, int rms,
                                   void* stream) {
    if (M <= 0 || D <= 0 || Kp < D || !gamma || (!rms && !beta) || !out3) { vly_set_error("vly_norm_split3_f32: bad args M=%d D=%d Kp=%d", M, D, Kp); return -22; }
    // (the same predicate as vly_norm_f32's: the pair norm + split3 and this kernel then take the same statistics routine)
    const bool vec = D % 4 == 0 && Kp % 4 == 0 && !(((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta) & 15) && !((uintptr_t)out3 & 7);
    if (vec) hipLaunchKernelGGL(norm_split3_vec_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (uint16_t*)out3, M, D, Kp, eps, rms);
    else hipLaunchKernelGGL(norm_split3_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (uint16_t*)out3, M, D, Kp, eps, rms);
    return vly_check_launch("vly_norm_split3_f32");
}

extern "C" int vly_rope_kv_f32(float* qkv, float* kcache, float* vcache, const float* cos_table, const float* sin_table, int B,
                               int S, int heads, int past_len, int ctx_max, void* stream) {
    if (B <= 0 || S <= 0 || heads <= 0 || past_len < 0 || past_len + S > ctx_max) {
        vly_set_error("vly_rope_kv_f32: bad args B=%d S=%d heads=%d past=%d ctx_max=%d", B, S, heads, past_len, ctx_max);
        return -22;
    }
    const long n = (long)B * S * heads * 64;
    hipLaunchKernelGGL(rope_kv_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, qkv, kcache, vcache,
                       cos_table, sin_table, B, S, heads, past_len, ctx_max);
    return vly_check_launch("vly_rope_kv_f32");
}

extern "C" int vly_patchify_f32(const float* images, float* patches, int F, int k_padded, void* stream) {
    if (F <= 0 || k_padded < 588) { vly_set_error("vly_patchify_f32: bad args F=%d KP=%d", F, k_padded); return -22; }
    const long total = (long)F * 256 * k_padded;
    hipLaunchKernelGGL(patchify_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, images, patches,
                       total, k_padded);
    return vly_check_launch("vly_patchify_f32");
}

extern "C" int vly_pool_tokens_f32(const float* feats, float* out, int B, int T, int W, int mode, const float* scores, void* stream) {
    if (B <= 0 || T <= 0 || W <= 0 || mode < VLY_POOL_MEAN || mode > VLY_POOL_IMPORTANCE || (mode == VLY_POOL_IMPORTANCE && !scores)) {
        vly_set_error("vly_pool_tokens_f32: bad args B=%d T=%d W=%d mode=%d", B, T, W, mode);
        return -22;
    }
    const long total = (long)B * (256 + T) * W;
    hipLaunchKernelGGL(pool_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, feats, out, B, T, W,
                       mode, scores);
    return vly_check_launch("vly_pool_tokens_f32");
}

extern "C" int vly_embed_splice_f32(const int32_t* row_map, const float* embed, const float* visual, float* out, int R, int H,
                                    void* stream) {
    if (R <= 0 || H <= 0) { vly_set_error("vly_embed_splice_f32: bad args R=%d H=%d", R, H); return -22; }
    const long total = (long)R * H;
    hipLaunchKernelGGL(embed_splice_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, row_map, embed,
                       visual, out, total, H);
    return vly_check_launch("vly_embed_splice_f32");
}
