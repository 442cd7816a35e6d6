// Persistent hybrid stream-K bf16 GEMM for gfx950: same math and epilogues as gemm_bf16.hip, different
// work decomposition.
//
// The hot-path GEMMs are small enough that whole-tile scheduling leaves CUs idle (M = 1312 rows at
// configs[1]; M = 8224 = 32 x 257 ViT rows is just past a multiple of every tile height), and their K
// loops are short enough (16 K tiles in the ViT) that the per-tile prologue (first loads: LDS-DMA issue ->
// landed is ~1.1 us) and epilogue are a large part of a tile's time.  One workgroup per CU stays resident
// and walks a private list of SEGMENTS (tile, k-range); the global->LDS pipeline runs straight across
// segment boundaries, so the loads of the next tile are in flight under the MFMAs and the epilogue of
// the current one.
//
// Schedule ("two-tile stream-K + data-parallel", per XCD so that the workgroups that run concurrently on
// one XCD / one L2 always work on ADJACENT tiles of the XCD's contiguous chunk of the tile order):
//   * XCD x owns tiles [T*x/8, T*(x+1)/8); its P workgroups (block b -> XCD b % 8, local index b / 8) take
//     whole tiles round-robin for all but the last full round ("DP" tiles);
//   * the last full round plus the remainder (P .. 2P-1 tiles, or everything when there are fewer than P;
//     split mode 2: the remainder only - a shorter pool phase, more workgroups per pool tile) is the
//     stream-K POOL: its (tile, k) iterations are linearised and cut into equal contiguous ranges, so
//     every workgroup gets the same number of MFMA iterations.  Pool ranges start at different k, so
//     unlike the DP rounds its workgroups do not read the same K slices of a shared panel at the same
//     time: the pool phase is the L2-unfriendly part and is kept short;
//   * a range that starts mid-tile is a CONTRIBUTOR segment (always the first thing a workgroup does): it
//     stores its fp32 accumulators to its slab and publishes a flag; the workgroup that owns k = 0 of the
//     tile adds the slabs of the following workgroups in fixed order and runs the epilogue.  Owners wait
//     only for work that was started at kernel entry: no deadlock as long as all workgroups are resident
//     (grid <= CUs x occupancy, enforced by the host); every spin is bounded anyway.
//   * hand-off = cdna_hip_programming.md Guideline 16: plain slab stores, every wave drains vmcnt,
//     barrier, one lane agent-scope release + vmcnt(0) + relaxed agent flag store; the owner polls
//     relaxed, one agent-scope acquire, barrier, plain loads.  Flags carry a per-launch epoch so they
//     never need re-zeroing.
// K loop variants (template NS, RS), as in gemm_bf16.hip: NS = 2 or 3 whole-K-tile LDS stages (3 = two K
// tiles of loads in flight, for the tiles whose stage is <= 53 KB), RS = role split (waves 4-7 run one
// phase behind waves 0-3, their SIMD partners).  Waits are counted (`vmcnt(L)`): gfx9 returns loads and
// stores in issue order, so "at most the newest L operations outstanding" always covers the loads that
// are about to be read, also when epilogue stores sit between them in the queue.
// Summation order inside a pool tile depends on where the range cuts fall, i.e. on M: results are
// deterministic for a given shape but NOT bit-identical across batch sizes (the tile kernel is).
#include "common.hpp"
#include "../../include/valley_hip.h"

int valley_p4_streamk(int tile, const void* A, const void* W, const float* bias, const float* R, void* C, int M, int N, int K, int lda,
                      int ldw, int ldc, int ldr, int epi, int out, void* ws, size_t ws_bytes, unsigned epoch, hipStream_t st);   // gemm_bf16.hip

namespace {

constexpr int BK = 64;
constexpr unsigned SPIN_LIMIT = 1u << 24;
constexpr size_t FLAG_BYTES = 16384;                      // room for 4000 workgroup flags + error word
constexpr int MIN_IT = 4;                                 // shortest stream-K range worth a slab round trip
constexpr int MAX_WAYS = 8;                               // most workgroups sharing one pool tile (serial slab adds)

struct Cursor {                                           // position in a workgroup's iteration stream
    int tile, k, kend, sk_left, r;
};

template <int BM, int BN, int WM, int WN, int EPI, int OUT, int NS, int RS>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64)
gemm_sk_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, const float* __restrict__ bias,
               const float* __restrict__ R, void* __restrict__ Cv, int M, int N, int K, int lda, int ldw, int ldc,
               int ldr, int tiles_m, int tiles_n, int m_fast, int split, float* __restrict__ slabs,
               unsigned* __restrict__ flags, unsigned epoch) {
    constexpr int NW = (BM / WM) * (BN / WN);
    constexpr int NT = NW * 64;
    constexpr int MI = WM / 16, NI = WN / 16;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
    constexpr int PA = BM * 8 / NT, PW = BN * 8 / NT;
    constexpr int LPI = PA + PW;                           // glds instructions per wave per iteration
    constexpr int SLAB = BM * BN;                          // floats per workgroup slab
    static_assert(NS == 2 || NS == 3, "2 or 3 LDS stages");
    static_assert(NS * STAGE <= 160 * 1024, "LDS budget");
    static_assert(!RS || NW == 8, "role split needs two waves per SIMD");

    __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int grp = wave >> 2;                             // role-split group (RS only)

    // ---------------- schedule --------------------------------------------------------------------------
    const int P = gridDim.x >> 3;                          // workgroups per XCD (host: gridDim.x % 8 == 0)
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int wid = x * P + j;                             // slab / flag index
    const int nk = K / BK;
    const int T = tiles_m * tiles_n;
    const int c0 = (int)((long)T * x / 8), c1 = (int)((long)T * (x + 1) / 8), Tx = c1 - c0;
    int pool, dp_rounds;
    {
        const int full = Tx / P, rem = Tx - full * P;
        if (!split) { pool = 0; dp_rounds = full + (rem ? 1 : 0); }           // whole tiles only, last round partial
        else if (rem == 0) { pool = 0; dp_rounds = full; }
        else if (full == 0) { pool = Tx; dp_rounds = 0; }
        else if (split == 2) { pool = rem; dp_rounds = full; }                // short tail, up to MAX_WAYS per tile
        else { pool = P + rem; dp_rounds = full - 1; }
    }
    const int pool_it = pool * nk;
    const int Pe = max(1, min(P, min(pool_it / MIN_IT, pool * MAX_WAYS)));   // workgroups that share the pool
    auto range_begin = [&](int jj) { return (int)((long)pool_it * min(jj, Pe) / Pe); };
    const int s0 = range_begin(j), s1 = range_begin(j + 1);
    int n_dp = dp_rounds;
    if (dp_rounds > 0 && c0 + pool + (dp_rounds - 1) * P + j >= c1) --n_dp;
    const int n_it = (s1 - s0) + n_dp * nk;
    if (n_it == 0) return;

    auto cur_init = [&](Cursor& c) {
        c.r = 0;
        if (s1 > s0) {
            c.tile = c0 + s0 / nk;
            c.k = s0 % nk;
            c.kend = min(nk, c.k + (s1 - s0));
            c.sk_left = (s1 - s0) - (c.kend - c.k);
        } else {
            c.sk_left = 0; c.tile = c0 + pool + j; c.k = 0; c.kend = nk; c.r = 1;
        }
    };
    auto cur_next_seg = [&](Cursor& c) {                   // the segment is used up and more iterations remain
        if (c.sk_left > 0) {
            ++c.tile; c.k = 0; c.kend = min(nk, c.sk_left); c.sk_left -= c.kend;
        } else {
            c.tile = c0 + pool + c.r * P + j; ++c.r; c.k = 0; c.kend = nk;
        }
    };

    // ---------------- staging ---------------------------------------------------------------------------
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
    Cursor sc;                                             // next iteration to stage
    int staged = 0, sbuf = 0;                              // iterations staged so far, LDS stage of the next one
    cur_init(sc);
    set_tile(sc.tile);
    auto stage_next = [&]() {                              // issue the loads of iteration `staged` (if any)
        if (staged >= n_it) return;
        char* sA = smem + sbuf * STAGE;
        char* sW = sA + A_BYTES;
        const int k0 = sc.k * BK;
#pragma unroll
        for (int p = 0; p < PA; ++p) glds16(A + offA[p] + k0, sA + (p * NT + wave * 64) * 16);
#pragma unroll
        for (int p = 0; p < PW; ++p) glds16(W + offW[p] + k0, sW + (p * NT + wave * 64) * 16);
        ++staged;
        sbuf = sbuf == NS - 1 ? 0 : sbuf + 1;
        if (++sc.k == sc.kend && staged < n_it) { cur_next_seg(sc); set_tile(sc.tile); }
    };
    // my loads of iteration i+1 have landed (at most the loads of i+2 stay in flight)
    auto wait_next = [&](int i) {
        if (NS == 3 && i + 2 < n_it) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    const int wm0 = (wave / (BN / WN)) * WM, wn0 = (wave % (BN / WN)) * WN;
    const int rdA = (wm0 + l15) * 128, rdW = A_BYTES + (wn0 + l15) * 128;
    const int sw0 = ((0 + g) ^ (l15 & 7)) << 4, sw1 = ((4 + g) ^ (l15 & 7)) << 4;
    float* my_slab = slabs + (size_t)wid * SLAB;

    stage_next();
    if (NS == 3) stage_next();
    wait_next(-1);                                         // iteration 0 landed (mine)
    __builtin_amdgcn_s_barrier();                          // ... and everyone's
    if (RS && grp == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one phase behind

    Cursor cc;
    cur_init(cc);
    int cbuf = 0;                                          // LDS stage of the iteration being computed
    int i = 0;                                             // iterations computed so far
    bool synced = false;                                   // the last segment brought the wave groups in step
    while (true) {
        // ---------------- one segment: iterations k in [cc.k, cc.kend) of cc.tile ------------------------
        const int tile = cc.tile, seg_k0 = cc.k, seg_k1 = cc.kend;
        f32x4 acc[MI][NI];
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int k = seg_k0; k < seg_k1; ++k, ++i) {
            const char* cur = smem + cbuf * STAGE;
            if constexpr (!RS) {
                stage_next();                              // iteration i + NS - 1 (its buffer was last read in i - 1)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int sw = kk ? sw1 : sw0;
                    bf16x8 af[MI], wf[NI];
#pragma unroll
                    for (int a = 0; a < MI; ++a) af[a] = *(const bf16x8*)(cur + rdA + a * 2048 + sw);
#pragma unroll
                    for (int b = 0; b < NI; ++b) wf[b] = *(const bf16x8*)(cur + rdW + b * 2048 + sw);
#pragma unroll
                    for (int a = 0; a < MI; ++a)
#pragma unroll
                        for (int b = 0; b < NI; ++b) acc[a][b] = mfma16(wf[b], af[a], acc[a][b]);
                }
                if (i + 1 < n_it) wait_next(i);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    // ---------------- R phase
                    if (kk == 0) stage_next();
                    const int sw = kk ? sw1 : sw0;
                    bf16x8 af[MI], wf[NI];
#pragma unroll
                    for (int b = 0; b < NI; ++b) wf[b] = *(const bf16x8*)(cur + rdW + b * 2048 + sw);
#pragma unroll
                    for (int a = 0; a < MI; ++a) af[a] = *(const bf16x8*)(cur + rdA + a * 2048 + sw);
                    if (kk == 1 && grp == 1) wait_next(i);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    // ---------------- M phase
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int a = 0; a < MI; ++a)
#pragma unroll
                        for (int b = 0; b < NI; ++b) acc[a][b] = mfma16(wf[b], af[a], acc[a][b]);
                    __builtin_amdgcn_s_setprio(0);
                    if (kk == 1 && grp == 0) wait_next(i);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                }
            }
            cbuf = cbuf == NS - 1 ? 0 : cbuf + 1;
        }

        const bool contributor = seg_k0 > 0, has_followers = !contributor && seg_k1 < nk;
        // segments that synchronise the whole workgroup: bring the two wave groups back in step first
        synced = contributor || has_followers;
        if (RS && synced && grp == 0) __builtin_amdgcn_s_barrier();

        if (contributor) {
            // publish partial sums (lane-linear float4 image, fully coalesced)
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b) *(f32x4*)(my_slab + ((a * NI + b) * NT + tid) * 4) = acc[a][b];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(flags + wid, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (has_followers) {
                // owner of a pool tile finished by the following workgroups: add their slabs in order
                const int tile_last = (tile - c0 + 1) * nk;
                for (int j2 = j + 1; j2 < Pe && range_begin(j2) < tile_last; ++j2) {
                    const int w2 = x * P + j2;
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
                    for (int a = 0; a < MI; ++a) {
#pragma unroll
                        for (int b = 0; b < NI; ++b) acc[a][b] += *(const f32x4*)(s2 + ((a * NI + b) * NT + tid) * 4);
                        if (a & 1) asm volatile("" ::: "memory");   // at most 2*NI slab loads in flight (VGPR budget)
                    }
                }
            }
            // ------------------------------ epilogue ------------------------------------------------
            const int m0 = (m_fast ? tile % tiles_m : tile / tiles_n) * BM;
            const int n0 = (m_fast ? tile / tiles_m : tile % tiles_n) * BN;
#pragma unroll
            for (int a = 0; a < MI; ++a) {
                asm volatile("" ::: "memory");                  // one fragment row of bias/residual loads at a time
                const int m = m0 + wm0 + a * 16 + l15;
                if (m >= M) continue;
#pragma unroll
                for (int b = 0; b < NI; ++b) {
                    const int n = n0 + wn0 + b * 16 + g * 4;
                    if (n >= N) continue;
                    f32x4 v = acc[a][b];
                    if (bias) v += *(const f32x4*)(bias + n);
                    if constexpr (EPI == VLY_EPI_QUICK_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = x_sigmoid(v[r], 1.702f);
                    }
                    if constexpr (EPI == VLY_EPI_RELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    if constexpr (EPI == VLY_EPI_SWIGLU) {
                        const float o0 = x_sigmoid(v[0], 1.f) * v[1];
                        const float o1 = x_sigmoid(v[2], 1.f) * v[3];
                        const size_t o = (size_t)m * ldc + (n >> 1);
                        if constexpr (OUT == VLY_OUT_BF16) *(uint32_t*)((uint16_t*)Cv + o) = pack_h2(o0, o1);
                        else *(float2*)((float*)Cv + o) = make_float2(o0, o1);
                    } else {
                        if (R) v += *(const f32x4*)(R + (size_t)m * ldr + n);
                        const size_t o = (size_t)m * ldc + n;
                        if constexpr (OUT == VLY_OUT_BF16) {
                            u32x2 pk;
                            pk[0] = pack_h2(v[0], v[1]);
                            pk[1] = pack_h2(v[2], v[3]);
                            *(u32x2*)((uint16_t*)Cv + o) = pk;
                        } else {
                            *(f32x4*)((float*)Cv + o) = v;
                        }
                    }
                }
            }
        }
        if (i >= n_it) break;
        if (RS && synced && grp == 1) __builtin_amdgcn_s_barrier();   // one phase behind again
        cur_next_seg(cc);
    }
    if (RS && !synced && grp == 0) __builtin_amdgcn_s_barrier();      // group 1's last phase barrier
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

struct SkArgs {
    const void *A, *W;
    const float *bias, *R;
    void* C;
    int M, N, K, lda, ldw, ldc, ldr, epi, out;
    void* ws;
    size_t ws_bytes;
    unsigned epoch;
    hipStream_t st;
};

template <int BM, int BN, int WM, int WN, int PER_CU, int NS, int RS>
int launch_sk(const SkArgs& a, int split) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    const int tm = (a.M + BM - 1) / BM, tn = (a.N + BN - 1) / BN;
    const int m_fast = vly_tile_order_m_fast(a.M, a.N, a.K, tm, tn);
    int G = num_cus() * PER_CU;
    G -= G & 7;
    if (G < 8) G = 8;
    // workspace = [FLAG_BYTES of flags (fixed place: stale contents are always old epochs)] [G slabs]
    const size_t need = FLAG_BYTES + (size_t)G * BM * BN * 4;
    if (!a.ws || a.ws_bytes < need || (size_t)(G + 8) * 4 > 16000) {
        vly_set_error("vly_gemm_bf16_streamk: workspace too small (%zu < %zu bytes)", a.ws_bytes, need);
        return -22;
    }
    unsigned* flags = (unsigned*)a.ws;
    float* slabs = (float*)((char*)a.ws + FLAG_BYTES);
    dim3 grid(G), block(NT);
#define VLY_SK_LAUNCH(E, O)                                                                                        \
    hipLaunchKernelGGL((gemm_sk_kernel<BM, BN, WM, WN, E, O, NS, RS>), grid, block, 0, a.st, (const uint16_t*)a.A, \
                       (const uint16_t*)a.W, a.bias, a.R, a.C, a.M, a.N, a.K, a.lda, a.ldw, a.ldc, a.ldr, tm, tn,  \
                       m_fast, split, slabs, flags, a.epoch)
    if (a.epi == VLY_EPI_NONE && a.out == VLY_OUT_BF16) VLY_SK_LAUNCH(VLY_EPI_NONE, VLY_OUT_BF16);
    else if (a.epi == VLY_EPI_NONE && a.out == VLY_OUT_F32) VLY_SK_LAUNCH(VLY_EPI_NONE, VLY_OUT_F32);
    else if (a.epi == VLY_EPI_QUICK_GELU && a.out == VLY_OUT_BF16) VLY_SK_LAUNCH(VLY_EPI_QUICK_GELU, VLY_OUT_BF16);
    else if (a.epi == VLY_EPI_SWIGLU && a.out == VLY_OUT_BF16) VLY_SK_LAUNCH(VLY_EPI_SWIGLU, VLY_OUT_BF16);
    else if (a.epi == VLY_EPI_RELU && a.out == VLY_OUT_BF16) VLY_SK_LAUNCH(VLY_EPI_RELU, VLY_OUT_BF16);
    else {
        vly_set_error("vly_gemm_bf16_streamk: unsupported epilogue/out_dtype combination (%d,%d)", a.epi, a.out);
        return -22;
    }
#undef VLY_SK_LAUNCH
    return vly_check_launch("vly_gemm_bf16_streamk");
}

// Default configuration when the caller gives no hint: the role-split 3-stage 192x192 kernel is the best
// all-rounder on the hot path's shapes; very small problems use the 128x128 tile (two workgroups per CU).
int pick_sk_tile(int M, int N, int K) {
    (void)K;
    const long tiles192 = (long)((M + 191) / 192) * ((N + 191) / 192);
    return tiles192 < num_cus() / 2 ? 2 : 86;
}

}  // namespace

extern "C" size_t vly_gemm_streamk_workspace_bytes(void) {
    // worst case over the configurations: G x BM x BN fp32 slabs + flags
    // (2 x: the persistent kernel's split-K remainder, hints 297 / 298 / 299, keeps one slab per contributor UNIT — up to S - 1 per
    // remainder tile; with less it settles for fewer slices)
    const size_t g1 = (size_t)num_cus();
    const size_t a = 2 * g1 * 256 * 256 * 4, b = 2 * g1 * 128 * 128 * 4;
    return FLAG_BYTES + (a > b ? a : b);
}

extern "C" int vly_gemm_streamk_tile_for(int M, int N, int K) { return pick_sk_tile(M, N, K); }

extern "C" int vly_gemm_bf16_streamk(const void* A, const void* W, const float* bias, const float* residual, void* C,
                                     int M, int N, int K, int lda, int ldw, int ldc, int ldr, int epilogue,
                                     int out_dtype, int tile_hint, void* workspace, size_t workspace_bytes,
                                     unsigned epoch, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) { vly_set_error("vly_gemm_bf16_streamk: empty problem"); return -22; }
    const bool p4 = tile_hint >= 297 && tile_hint <= 299;   // the persistent 4-wave kernel with a stream-K pool (gemm_bf16.hip)
    if (K % BK || lda % 8 || ldw % 8 || (ldw <= 0 && !(p4 && ldw == VLY_LDW_PACKED64)) /* row-major weights, or the block layout for 297 - 299 */ || N % 4 || ldc % 2 || (epilogue == VLY_EPI_SWIGLU && N % 8) ||
        ((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C & 7) || ((uintptr_t)workspace & 15) ||
        (residual && (ldr % 4 || ((uintptr_t)residual & 15))) || (bias && ((uintptr_t)bias & 15)) || epoch == 0) {
        vly_set_error("vly_gemm_bf16_streamk: unsupported shape/alignment M=%d N=%d K=%d lda=%d ldw=%d ldc=%d ldr=%d",
                      M, N, K, lda, ldw, ldc, ldr);
        return -22;
    }
    if ((size_t)M * lda >= (1ull << 32) || (size_t)N * (ldw > 0 ? ldw : K) >= (1ull << 32)) {
        vly_set_error("vly_gemm_bf16_streamk: operand exceeds 2^32 elements");
        return -22;
    }
    if (epilogue == VLY_EPI_SWIGLU && residual) { vly_set_error("vly_gemm_bf16_streamk: SWIGLU takes no residual"); return -22; }
    if (p4)
        return valley_p4_streamk(tile_hint, A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, workspace,
                                 workspace_bytes, epoch, (hipStream_t)stream);
    const SkArgs a{A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype,
                   workspace, workspace_bytes, epoch, (hipStream_t)stream};
    const int t = tile_hint ? tile_hint : pick_sk_tile(M, N, K);
    // t % 10 = tile (1 = 256x256, 2 = 128x128, 3 = 256x128, 4 = 128x256, 5 = 192x256, 6 = 192x192);
    // t / 10 = loop: 0 = 2 stages, 4 = 2 stages without K splitting (whole tiles round-robin), 5 = 2 stages
    // role-split, 7 = 3 stages, 8 = 3 stages role-split;  +100 = the pool is the remainder round only.
    const int split = t >= 100 ? 2 : 1;                    // +100: remainder-only pool
    switch (t % 100) {
        case 1: return launch_sk<256, 256, 128, 64, 1, 2, 0>(a, split);
        case 2: return launch_sk<128, 128, 64, 64, 2, 2, 0>(a, split);
        case 3: return launch_sk<256, 128, 64, 64, 1, 2, 0>(a, split);
        case 4: return launch_sk<128, 256, 64, 64, 1, 2, 0>(a, split);
        case 5: return launch_sk<192, 256, 96, 64, 1, 2, 0>(a, split);
        case 6: return launch_sk<192, 192, 96, 48, 1, 2, 0>(a, split);
        case 41: return launch_sk<256, 256, 128, 64, 1, 2, 0>(a, 0);
        case 42: return launch_sk<128, 128, 64, 64, 2, 2, 0>(a, 0);
        case 45: return launch_sk<192, 256, 96, 64, 1, 2, 0>(a, 0);
        case 51: return launch_sk<256, 256, 128, 64, 1, 2, 1>(a, split);
        case 55: return launch_sk<192, 256, 96, 64, 1, 2, 1>(a, split);
        case 73: return launch_sk<256, 128, 64, 64, 1, 3, 0>(a, split);
        case 74: return launch_sk<128, 256, 64, 64, 1, 3, 0>(a, split);
        case 76: return launch_sk<192, 192, 96, 48, 1, 3, 0>(a, split);
        case 83: return launch_sk<256, 128, 64, 64, 1, 3, 1>(a, split);
        case 84: return launch_sk<128, 256, 64, 64, 1, 3, 1>(a, split);
        case 86: return launch_sk<192, 192, 96, 48, 1, 3, 1>(a, split);
        default: vly_set_error("vly_gemm_bf16_streamk: bad tile_hint %d", tile_hint); return -22;
    }
}
