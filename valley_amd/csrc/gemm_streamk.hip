// Persistent stream-K bf16 GEMM for gfx950: same math and epilogues as gemm_bf16.hip, different
// work decomposition.
//
// The hot-path GEMMs are small enough that whole-tile scheduling leaves CUs idle (M = 1312 rows at
// configs[1]: 288 tiles of 256x256 on 256 CUs = 2 rounds for 1.125 rounds of work).  Here the
// iteration space (tile, k-tile) is linearised tile-major and cut into G equal contiguous ranges,
// one per persistent workgroup (G = CUs x workgroups/CU):
//   * the K loop runs straight across tile boundaries: the global->LDS prefetch of iteration i+1
//     (which may belong to the NEXT tile) is issued before iteration i's MFMAs and therefore also
//     overlaps tile i's epilogue — no per-tile prologue/epilogue bubble;
//   * a range that starts mid-tile ("contributor" segment, always the FIRST thing a workgroup does)
//     stores its fp32 accumulators to its private slab and publishes a flag; the workgroup that owns
//     k = 0 of that tile ("owner", always its LAST segment) adds the slabs of the following
//     workgroups in fixed order and runs the epilogue.  Owners therefore wait only for work that was
//     started at kernel entry: no deadlock as long as all G workgroups are resident (G <= CUs x
//     occupancy, enforced by the host), every spin is bounded anyway.
//   * hand-off = cdna_hip_programming.md Guideline 16: plain slab stores, every wave drains vmcnt,
//     barrier, one lane agent-scope release + asm vmcnt(0) + relaxed agent flag store; the owner polls
//     relaxed, one agent-scope acquire, barrier, plain loads.  Flags carry a per-launch epoch so they
//     never need re-zeroing.
// Summation order inside a tile depends on where the range cuts fall, i.e. on M: results are
// deterministic for a given shape but NOT bit-identical across batch sizes (the tile kernel is).
#include "common.hpp"
#include "../../include/valley_hip.h"

namespace {

constexpr int BK = 64;
constexpr unsigned SPIN_LIMIT = 1u << 24;
constexpr size_t FLAG_BYTES = 16384;                      // room for 4088 workgroup flags + error word

template <int BM, int BN, int WM, int WN, int EPI, int OUT>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64)
gemm_sk_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, const float* __restrict__ bias,
               const float* __restrict__ R, void* __restrict__ Cv, int M, int N, int K, int lda, int ldw, int ldc,
               int ldr, int tiles_m, int tiles_n, int m_fast, int total_iters, int unit, float* __restrict__ slabs,
               unsigned* __restrict__ flags, unsigned epoch) {
    constexpr int NW = (BM / WM) * (BN / WN);
    constexpr int NT = NW * 64;
    constexpr int MI = WM / 16, NI = WN / 16;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
    constexpr int PA = BM * 8 / NT, PW = BN * 8 / NT;
    constexpr int SLAB = BM * BN;                          // floats per workgroup slab

    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;

    // XCD-contiguous workgroup order: hardware puts block b on XCD b % 8 (speed only)
    const int G = gridDim.x;
    const int w = (G & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3));
    const int nk = K / BK;
    // unit == 1: ranges may cut tiles anywhere (stream-K); unit == nk: ranges are whole tiles (persistent
    // data-parallel: no fix-up, still no per-tile prologue bubble)
    const int units = total_iters / unit;
    const int q = units / G, rem = units % G;
    auto range_begin = [&](int x) { return (x * q + min(x, rem)) * unit; };
    const int it0 = range_begin(w), it1 = range_begin(w + 1);
    if (it0 >= it1) return;

    uint32_t offA[PA], offW[PW];
    auto set_tile = [&](int tile) {
        const int m0 = (m_fast ? tile % tiles_m : tile / tiles_n) * BM;
        const int n0 = (m_fast ? tile / tiles_m : tile % tiles_n) * BN;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int s = p * NT + tid, row = s >> 3, cp = s & 7;
            offA[p] = (uint32_t)min(m0 + row, M - 1) * (uint32_t)lda + (uint32_t)((cp ^ (row & 7)) << 3);
        }
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            const int s = p * NT + tid, row = s >> 3, cp = s & 7;
            offW[p] = (uint32_t)min(n0 + row, N - 1) * (uint32_t)ldw + (uint32_t)((cp ^ (row & 7)) << 3);
        }
    };
    auto stage = [&](int k, int buf) {
        char* sA = smem + buf * STAGE;
        char* sW = sA + A_BYTES;
        const int k0 = k * BK;
#pragma unroll
        for (int p = 0; p < PA; ++p) glds16(A + offA[p] + k0, sA + (p * NT + wave * 64) * 16);
#pragma unroll
        for (int p = 0; p < PW; ++p) glds16(W + offW[p] + k0, sW + (p * NT + wave * 64) * 16);
    };

    const int wm0 = (wave / (BN / WN)) * WM, wn0 = (wave % (BN / WN)) * WN;
    const int rdA = (wm0 + l15) * 128, rdW = (wn0 + l15) * 128;
    const int sw0 = ((0 + g) ^ (l15 & 7)) << 4, sw1 = ((4 + g) ^ (l15 & 7)) << 4;

    int tile = it0 / nk;
    int seg_k0 = it0 - tile * nk;                          // first k of the running segment
    int st_tile = tile, st_k = seg_k0;                     // coordinates of the iteration staged next
    set_tile(st_tile);
    stage(st_k, 0);
    float* my_slab = slabs + (size_t)w * SLAB;

    int it = it0;
    while (it < it1) {
        // ---------------- one segment: iterations [it, seg_end) all belong to `tile` -----------------
        const int tile_last = (tile + 1) * nk;             // first iteration of the next tile
        const int seg_end = min(it1, tile_last);
        f32x4 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (; it < seg_end; ++it) {
            __syncthreads();                               // iteration `it` landed; other buffer is free
            if (it + 1 < it1) {                            // prefetch runs across tile boundaries
                if (++st_k == nk) { st_k = 0; ++st_tile; set_tile(st_tile); }
                stage(st_k, (it + 1 - it0) & 1);
            }
            const char* sA = smem + ((it - it0) & 1) * STAGE;
            const char* sW = sA + A_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int sw = kk ? sw1 : sw0;
                bf16x8 af[MI], wf[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(sA + rdA + i * 2048 + sw);
#pragma unroll
                for (int j = 0; j < NI; ++j) wf[j] = *(const bf16x8*)(sW + rdW + j * 2048 + sw);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
            }
        }

        if (seg_k0 > 0) {
            // contributor: publish partial sums (lane-linear float4 image, fully coalesced)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) *(f32x4*)(my_slab + ((i * NI + j) * NT + tid) * 4) = acc[i][j];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(flags + w, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (seg_end < tile_last) {
                // owner of a tile finished by the following workgroups: add their slabs in order
                for (int w2 = w + 1; w2 < G && range_begin(w2) < tile_last; ++w2) {
                    if (tid == 0) {
                        unsigned spins = 0;
                        while (__hip_atomic_load(flags + w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                            __builtin_amdgcn_s_sleep(8);
                            if (++spins > SPIN_LIMIT) {            // never hang the GPU: flag the failure
                                __hip_atomic_store(flags + 4000, 0xDEADu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    __syncthreads();
                    const float* s2 = slabs + (size_t)w2 * SLAB;
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
#pragma unroll
                        for (int j = 0; j < NI; ++j) acc[i][j] += *(const f32x4*)(s2 + ((i * NI + j) * NT + tid) * 4);
                        if (i & 1) asm volatile("" ::: "memory");   // at most 2*NI slab loads in flight (VGPR budget)
                    }
                }
            }
            // ------------------------------ epilogue ------------------------------------------------
            const int m0 = (m_fast ? tile % tiles_m : tile / tiles_n) * BM;
            const int n0 = (m_fast ? tile / tiles_m : tile % tiles_n) * BN;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                asm volatile("" ::: "memory");                  // one fragment row of bias/residual loads at a time
                const int m = m0 + wm0 + i * 16 + l15;
                if (m >= M) continue;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int n = n0 + wn0 + j * 16 + g * 4;
                    if (n >= N) continue;
                    f32x4 v = acc[i][j];
                    if (bias) v += *(const f32x4*)(bias + n);
                    if constexpr (EPI == VLY_EPI_QUICK_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.f + __expf(-1.702f * v[r]));
                    }
                    if constexpr (EPI == VLY_EPI_RELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    if constexpr (EPI == VLY_EPI_SWIGLU) {
                        const float o0 = v[0] / (1.f + __expf(-v[0])) * v[1];
                        const float o1 = v[2] / (1.f + __expf(-v[2])) * v[3];
                        const size_t o = (size_t)m * ldc + (n >> 1);
                        if constexpr (OUT == VLY_OUT_BF16) *(uint32_t*)((uint16_t*)Cv + o) = pack_bf16x2(o0, o1);
                        else *(float2*)((float*)Cv + o) = make_float2(o0, o1);
                    } else {
                        if (R) v += *(const f32x4*)(R + (size_t)m * ldr + n);
                        const size_t o = (size_t)m * ldc + n;
                        if constexpr (OUT == VLY_OUT_BF16) {
                            u32x2 pk;
                            pk[0] = pack_bf16x2(v[0], v[1]);
                            pk[1] = pack_bf16x2(v[2], v[3]);
                            *(u32x2*)((uint16_t*)Cv + o) = pk;
                        } else {
                            *(f32x4*)((float*)Cv + o) = v;
                        }
                    }
                }
            }
        }
        seg_k0 = 0;
        ++tile;
    }
}

int g_num_cus = 0;
int num_cus() {
    if (!g_num_cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        g_num_cus = n;
    }
    return g_num_cus;
}

template <int BM, int BN, int WM, int WN, int PER_CU>
int launch_sk(const void* A, const void* W, const float* bias, const float* R, void* C, int M, int N, int K, int lda,
              int ldw, int ldc, int ldr, int epi, int out, void* ws, size_t ws_bytes, unsigned epoch, hipStream_t st,
              bool aligned = false) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
    const long total = (long)tm * tn * (K / BK);
    const int m_fast = vly_tile_order_m_fast(M, N, K, tm, tn);
    int G = num_cus() * PER_CU;
    G -= G & 7;
    const int unit = aligned ? K / BK : 1;
    if (total / unit < G) G = (int)(total / unit);
    // workspace = [FLAG_BYTES of flags (fixed place: stale contents are always old epochs)] [G slabs]
    const size_t need = FLAG_BYTES + (size_t)G * BM * BN * 4;
    if (!ws || ws_bytes < need || (size_t)(G + 8) * 4 > FLAG_BYTES) {
        vly_set_error("vly_gemm_bf16_streamk: workspace too small (%zu < %zu bytes)", ws_bytes, need);
        return -22;
    }
    unsigned* flags = (unsigned*)ws;
    float* slabs = (float*)((char*)ws + FLAG_BYTES);
    dim3 grid(G), block(NT);
#define VLY_SK_LAUNCH(E, O)                                                                                 \
    hipLaunchKernelGGL((gemm_sk_kernel<BM, BN, WM, WN, E, O>), grid, block, 0, st, (const uint16_t*)A,      \
                       (const uint16_t*)W, bias, R, C, M, N, K, lda, ldw, ldc, ldr, tm, tn, m_fast, (int)total, unit, slabs, flags, epoch)
    if (epi == VLY_EPI_NONE && out == VLY_OUT_BF16) VLY_SK_LAUNCH(VLY_EPI_NONE, VLY_OUT_BF16);
    else if (epi == VLY_EPI_NONE && out == VLY_OUT_F32) VLY_SK_LAUNCH(VLY_EPI_NONE, VLY_OUT_F32);
    else if (epi == VLY_EPI_QUICK_GELU && out == VLY_OUT_BF16) VLY_SK_LAUNCH(VLY_EPI_QUICK_GELU, VLY_OUT_BF16);
    else if (epi == VLY_EPI_SWIGLU && out == VLY_OUT_BF16) VLY_SK_LAUNCH(VLY_EPI_SWIGLU, VLY_OUT_BF16);
    else if (epi == VLY_EPI_RELU && out == VLY_OUT_BF16) VLY_SK_LAUNCH(VLY_EPI_RELU, VLY_OUT_BF16);
    else {
        vly_set_error("vly_gemm_bf16_streamk: unsupported epilogue/out_dtype combination (%d,%d)", epi, out);
        return -22;
    }
#undef VLY_SK_LAUNCH
    return vly_check_launch("vly_gemm_bf16_streamk");
}

// Modelled time (arbitrary units) of each configuration: balanced MFMA work over the persistent
// grid at the configuration's relative efficiency + the fix-up traffic when ranges cut tiles.
int pick_sk_tile(int M, int N, int K) {
    const int cus = num_cus();
    double best = 1e300;
    int arg = 1;
    const struct { int id, bm, bn, per_cu; double eff; } cfgs[] = {
        {1, 256, 256, 1, 1.00}, {3, 256, 128, 1, 0.86}, {2, 128, 128, 2, 0.78}};
    for (const auto& c : cfgs) {
        const long tiles = (long)((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
        const long iters = tiles * (K / BK);
        const int G = cus * c.per_cu;
        const double its_per_wg = (double)((iters + G - 1) / G);
        // one iteration of a bm x bn x 64 tile on one CU at ~4.5 TFLOP/s/CU (1.15 PF chip) ~ us
        const double t_iter = 2.0 * c.bm * c.bn * BK / (4.5e6 * c.eff) * c.per_cu;
        double t = its_per_wg * t_iter;
        if (tiles % G != 0)                                             // ranges cut tiles: slab write + read
            t += 2.0 * c.bm * c.bn * 4 / 60e3 * c.per_cu;               // at ~60 GB/s per workgroup, in us
        if (t < best) { best = t; arg = c.id; }
    }
    return arg;
}

}  // namespace

extern "C" size_t vly_gemm_streamk_workspace_bytes(void) {
    // worst case over the configurations: G x BM x BN fp32 slabs + flags
    const size_t g1 = (size_t)num_cus();
    const size_t a = g1 * 256 * 256 * 4, b = 2 * g1 * 128 * 128 * 4;
    return FLAG_BYTES + (a > b ? a : b);
}

extern "C" int vly_gemm_streamk_tile_for(int M, int N, int K) { return pick_sk_tile(M, N, K); }

extern "C" int vly_gemm_bf16_streamk(const void* A, const void* W, const float* bias, const float* residual, void* C,
                                     int M, int N, int K, int lda, int ldw, int ldc, int ldr, int epilogue,
                                     int out_dtype, int tile_hint, void* workspace, size_t workspace_bytes,
                                     unsigned epoch, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) { vly_set_error("vly_gemm_bf16_streamk: empty problem"); return -22; }
    if (K % BK || lda % 8 || ldw % 8 || N % 4 || ldc % 2 || (epilogue == VLY_EPI_SWIGLU && N % 8) ||
        ((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C & 7) || ((uintptr_t)workspace & 15) ||
        (residual && (ldr % 4 || ((uintptr_t)residual & 15))) || (bias && ((uintptr_t)bias & 15)) || epoch == 0) {
        vly_set_error("vly_gemm_bf16_streamk: unsupported shape/alignment M=%d N=%d K=%d lda=%d ldw=%d ldc=%d ldr=%d",
                      M, N, K, lda, ldw, ldc, ldr);
        return -22;
    }
    if ((size_t)M * lda >= (1ull << 32) || (size_t)N * ldw >= (1ull << 32)) {
        vly_set_error("vly_gemm_bf16_streamk: operand exceeds 2^32 elements");
        return -22;
    }
    if (epilogue == VLY_EPI_SWIGLU && residual) { vly_set_error("vly_gemm_bf16_streamk: SWIGLU takes no residual"); return -22; }
    hipStream_t st = (hipStream_t)stream;
    const int t = tile_hint ? tile_hint : pick_sk_tile(M, N, K);
    switch (t) {
        case 1: return launch_sk<256, 256, 128, 64, 1>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, workspace, workspace_bytes, epoch, st);
        case 2: return launch_sk<128, 128, 64, 64, 2>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, workspace, workspace_bytes, epoch, st);
        case 3: return launch_sk<256, 128, 64, 64, 1>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, workspace, workspace_bytes, epoch, st);
        case 4: return launch_sk<128, 256, 64, 64, 1>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, workspace, workspace_bytes, epoch, st);
        case 5: return launch_sk<192, 256, 96, 64, 1>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, workspace, workspace_bytes, epoch, st);
        // 41..45: persistent whole-tile ranges (no fix-up)
        case 41: return launch_sk<256, 256, 128, 64, 1>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, workspace, workspace_bytes, epoch, st, true);
        case 42: return launch_sk<128, 128, 64, 64, 2>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, workspace, workspace_bytes, epoch, st, true);
        case 43: return launch_sk<256, 128, 64, 64, 1>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, workspace, workspace_bytes, epoch, st, true);
        case 44: return launch_sk<128, 256, 64, 64, 1>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, workspace, workspace_bytes, epoch, st, true);
        case 45: return launch_sk<192, 256, 96, 64, 1>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, workspace, workspace_bytes, epoch, st, true);
        default: vly_set_error("vly_gemm_bf16_streamk: bad tile_hint %d", tile_hint); return -22;
    }
}
