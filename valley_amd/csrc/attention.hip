// Attention kernels for gfx950 (MFMA 16x16x32 bf16, fp32 softmax).
//
// Both kernels compute S^T = K Q^T with the K fragment as the MFMA "A" operand, so a lane holds,
// for ONE query row q = lane&15, the scores of keys kv = 16t + 4*(lane>>4) + r: the softmax row
// reduction is in-register plus two xor-shuffles (16, 32), and the probabilities are already in the
// B-operand position of the second product O^T = V^T P^T (the k-slot order of a lane's 8 values is
// chosen to match, so P never crosses lanes).  V is transposed while it is staged into LDS
// (d-major rows, stride = 16*odd bytes mod 256 so the 8-byte fragment reads are conflict-free);
// K keeps its row-major image with the 16-byte-chunk XOR swizzle.
//
//   vit_attention   : one workgroup per (frame, head); all 257 keys/values of the head live in
//                     LDS (70.7 KB -> 2 workgroups per CU), no online softmax needed.
//                     flop/frame/layer = 4*257*257*64*16 = 0.271 GFLOP.
//   llama_attention : flash-style over KV-cache tiles of 64 keys with online softmax, causal +
//                     key-validity mask, head_dim 128; also serves decode (S = 1).
#include <cstdlib>
#include "common.hpp"
#include "../../include/valley_hip.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;

// Softmax exponentials: exp2 of (score - running max) <= 0.  The library exp2f wraps v_exp_f32 in a denormal-safe
// rescale (v_cmp + 2 v_cndmask + v_add + v_exp + v_ldexp: six instructions, 396 of the ViT kernel's 978 VALU
// instructions per query tile); the bare instruction flushes results below 2^-126 to zero, which is exactly what a
// softmax weight that small is worth (VLY_FAST_EXP2=0 restores the library call for A/B builds).
#ifndef VLY_DECODE_MERGE_FENCE
#define VLY_DECODE_MERGE_FENCE 0    // 1: release / acquire fences around the merged decode attention's ticket (see split_merge_if_last)
#endif
#ifndef VLY_VIT_PRIO
#define VLY_VIT_PRIO 0              // 1: s_setprio 1 around the MFMA runs of the ViT kernel (QK^T, PV), 0 in the softmax
#endif
#ifndef VLY_VIT_STAGGER
#define VLY_VIT_STAGGER 0           // N > 0: waves 4-7 of a workgroup start their query tiles N x 64 clocks late (the two waves of a SIMD out of phase)
#endif
#ifndef VLY_VIT_TIMING
#define VLY_VIT_TIMING 0            // anatomy builds (tools/vit_attn_times.py): every wave of every workgroup stamps s_memtime at the seams of its query tiles
#endif
#ifndef VLY_VIT_STAGE2
#define VLY_VIT_STAGE2 1            // 1: all of a workgroup's start-up loads in flight at once (0: round 2's load / wait / store loops)
#endif
#ifndef VLY_VIT_ROT
#define VLY_VIT_ROT 0               // 1: the 17th query tile of a head goes to a wave picked from the workgroup index (not always wave 0 = SIMD 0)
#endif
#ifndef VLY_VIT_FOLD
#define VLY_VIT_FOLD 0              // 1: softmax on the raw scores — max by v_max3, the 64^-0.5 scale inside the exponent's FMA (one VALU op per score fewer; not bit-identical)
#endif
#ifndef VLY_VIT_STORE_LINES
#define VLY_VIT_STORE_LINES 0       // 1: the ViT kernel's outputs leave as whole 128-byte lines (measured equal, bit-identical: profiles/r06/r06_vit_attn_store_lines.txt)
#endif
#ifndef VLY_FAST_EXP2
#define VLY_FAST_EXP2 1
#endif
VLY_DEVICE float sm_exp2(float x) {
#if VLY_FAST_EXP2
    return __builtin_amdgcn_exp2f(x);
#else
    return exp2f(x);
#endif
}

VLY_DEVICE uint32_t sel16(const u32x4& a, const u32x4& b, int dd) {   // element dd (0..15) of 16 bf16 in (a,b)
    const uint32_t w = dd < 8 ? a[dd >> 1] : b[(dd - 8) >> 1];
    return (dd & 1) ? (w >> 16) : (w & 0xffffu);
}

// The two 8-byte halves of a V^T fragment (keys 4g..4g+3 and 16+4g..) must stay TWO ds_read_b64: left alone, hipcc's
// load/store optimizer fuses them into one ds_read2_b64 — half the LDS rate (8 instead of 2 x 2 cycles per wave
// instruction, MI355X_MICROARCH §LDS) and banked mod 32 instead of mod 64, where the V^T row strides below (592 / 144 bytes,
// chosen conflict-free for ds_read_b64) collide 2-way.  PMC, round 3 (profiles/history/r03/r03_pmc_attention.txt): 37 % of the ViT
// kernel's LDS cycles were bank conflicts and its waves sat in s_waitcnt 56 % of the time.  An offset the compiler cannot
// see through keeps the second read on its own base register.
VLY_DEVICE int opaque_i32(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// Output of one query tile: lane (q = l15, g) holds O[q][d = dt*16 + 4g + r] — four 8-byte pieces, one per dt, each a quarter
// of a 32-byte run of its row.  Stored as they are, a wave instruction writes 16 rows x 32 bytes and every 128-byte line is
// touched by four instructions: the timing variant without the stores runs 75 instead of 90 us per layer at 128 frames
// (profiles/history/r03/r03_vit_attn_timing_variants.jsonl).  A 4 x 4 transpose across the four lanes of a row (the SwiGLU epilogue's
// v_permlane16/32_swap pattern, once per 32-bit word) leaves lane g with the whole 32-byte run d = 16g .. 16g + 15.
VLY_DEVICE void store_tile_rows(uint16_t* row_ptr, const f32x4 (&o)[4], float inv, int g) {
    uint32_t y[2][4];
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        uint32_t d[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) d[dt] = pack_h2(o[dt][2 * w] * inv, o[dt][2 * w + 1] * inv);
        const auto p01 = __builtin_amdgcn_permlane16_swap(d[0], d[1], false, false);
        const auto p23 = __builtin_amdgcn_permlane16_swap(d[2], d[3], false, false);
        const auto q0 = __builtin_amdgcn_permlane32_swap(p01[0], p23[0], false, false);
        const auto q1 = __builtin_amdgcn_permlane32_swap(p01[1], p23[1], false, false);
        y[w][0] = q0[0]; y[w][1] = q1[0]; y[w][2] = q0[1]; y[w][3] = q1[1];       // y[w][j] = word w of lane j's piece dt = g
    }
    u32x4* dst = (u32x4*)(row_ptr + 16 * g);
    dst[0] = u32x4{y[0][0], y[1][0], y[0][1], y[1][1]};
    dst[1] = u32x4{y[0][2], y[1][2], y[0][3], y[1][3]};
}

// v (op)= v[lane ^ 16], then v[lane ^ 32]: the four lane groups of a query column, by the two gfx950 lane swaps instead of __shfl_xor's
// ds_bpermute round trips (same partners, same order: bit-identical; see wave_reduce in common.hpp)
template <typename OP>
VLY_DEVICE float rows4_reduce(float v, OP op) {
    {
        const uint32_t x = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);     // rows {0, 0, 2, 2} and {1, 1, 3, 3}
        v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    {
        const uint32_t x = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);     // {lo, lo} and {hi, hi}
        v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    return v;
}
VLY_DEVICE float rows4_max(float v) { return rows4_reduce(v, [](float a, float b) { return fmaxf(a, b); }); }
VLY_DEVICE float rows4_sum(float v) { return rows4_reduce(v, [](float a, float b) { return a + b; }); }

// ---------------------------------------------------------------------------------------------
// ViT attention
// ---------------------------------------------------------------------------------------------
constexpr int VN = 257;            // tokens
constexpr int VNT = 17;            // 16-key tiles (272 padded keys)
constexpr int VNC = 9;             // 32-key chunks for P·V (288 padded keys)
constexpr int VLD = 3072;          // qkv row stride (elements)
constexpr int VT_STRIDE = 296;     // V^T row stride in elements (592 B = 2*256 + 16*5)
constexpr int VK_BYTES = VNT * 16 * 128;

constexpr int VNW = 8;             // waves per workgroup: 2 workgroups x 8 waves = 4 waves per SIMD hide the LDS / MFMA
                                   // latency chains of a 16-query block (4 waves: 31.5 us per layer at 32 frames)

#if VLY_VIT_TIMING
constexpr int VTS_WG = 4096, VTS_N = 24;                    // stamped workgroups (the first 4096), stamps per wave
__device__ unsigned long long vit_ts[VTS_WG * VNW * VTS_N];
#define VLY_VSTAMP()                                                                                     \
    do {                                                                                                  \
        if (ts_row && lane == 0 && tsn < VTS_N) ts_row[tsn] = __builtin_readcyclecounter();                \
        ++tsn;                                                                                            \
    } while (0)
#else
#define VLY_VSTAMP() do {} while (0)
#endif

__global__ void __launch_bounds__(VNW * 64, 4) vit_attn_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) char smem[VK_BYTES + 64 * VT_STRIDE * 2];
    char* sK = smem;
    uint16_t* sVt = (uint16_t*)(smem + VK_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    [[maybe_unused]] const int hi16 = opaque_i32(16);          // see opaque_i32
    const int h = blockIdx.x & 15;
#if VLY_VIT_TIMING
    int tsn = 0;
    unsigned long long* ts_row = blockIdx.x < VTS_WG ? vit_ts + ((size_t)blockIdx.x * VNW + wave) * VTS_N : nullptr;
    VLY_VSTAMP();                                                    // 0: start
    if (ts_row && lane == 0) {                                       // 1: HW_ID (which XCD / CU / SIMD this wave sits on)
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        ts_row[tsn] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    }
    ++tsn;
#endif
    // (Timing variants of this kernel — staging only, no LDS reads, no stores, L2-resident inputs, no HBM traffic, no exp2, no
    // softmax arithmetic — were compile-time switches here while profiles/history/r03/r03_vit_attn_timing_variants{,2}.jsonl were
    // measured; they and the old 8-byte store pattern left with commit 49dfb3f's successor.)
    const int f = blockIdx.x >> 4;
    const uint16_t* base = qkv + (size_t)f * VN * VLD + h * 64;

#if VLY_VIT_STAGE2
    // ---- staging, round 6: EVERY global load of the workgroup's start is issued before the first LDS store — the K slots (five per
    //      thread), the V rows (two key pairs per thread) and the first query tile's fragments — so the start costs one memory round trip
    //      plus the transfer.  (The loops below this block — one load, s_waitcnt vmcnt(0), one store per trip — cost five round trips
    //      in series: 12.7 k of a workgroup's 35 k cycles, profiles/r06/r06_vit_attn_anatomy.txt.)
    //      Loads are unconditional on clamped addresses (a load under a lane mask has to be merged with the masked lanes' value, which
    //      costs a vmcnt(0) on the spot); the padding is zeroed where the LDS words are formed.
    const int dq = wave & 3;
    u32x4 kr[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int sl = tid + i * VNW * 64, row = min(sl >> 3, VN - 1), c = sl & 7;
        kr[i] = *(const u32x4*)(base + (size_t)row * VLD + 1024 + c * 8);
    }
    u32x4 vr[2][4];                                        // [pass][a0, a1, b0, b1]: pass 0 = key-pair group wave >> 2, pass 1 = group 2 (waves 0-3)
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int pg = ps == 0 ? (wave >> 2) : 2, kv0 = 2 * (pg * 64 + lane), kv1 = kv0 + 1;
        if (ps == 0 || wave < 4) {                         // (wave-uniform)
            const uint16_t* r0 = base + (size_t)min(kv0, VN - 1) * VLD + 2048 + dq * 16;
            const uint16_t* r1 = base + (size_t)min(kv1, VN - 1) * VLD + 2048 + dq * 16;
            vr[ps][0] = *(const u32x4*)r0;
            vr[ps][1] = *(const u32x4*)(r0 + 8);
            vr[ps][2] = *(const u32x4*)r1;
            vr[ps][3] = *(const u32x4*)(r1 + 8);
        }
    }
    bf16x8 qn[2];
    {
        const int qc0 = min(wave * 16 + l15, VN - 1);
        qn[0] = *(const bf16x8*)(base + (size_t)qc0 * VLD + g * 8);
        qn[1] = *(const bf16x8*)(base + (size_t)qc0 * VLD + 32 + g * 8);
    }
    // K: [272][64] bf16, 128-byte rows, chunk ^= row & 7
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int sl = tid + i * VNW * 64, row = sl >> 3, c = sl & 7;
        if (sl < VNT * 16 * 8) *(u32x4*)(sK + row * 128 + ((c ^ (row & 7)) << 4)) = row < VN ? kr[i] : u32x4{0u, 0u, 0u, 0u};
    }
    // V^T: [64 d][288 kv]; wave w transposes d = 16 (w & 3) .. + 15, lane <-> key pair
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int pg = ps == 0 ? (wave >> 2) : 2, kv0 = 2 * (pg * 64 + lane);
        if (ps == 0 || (wave < 4 && lane < VNC * 16 - 128)) {
            const uint32_t m0 = kv0 < VN ? 0xffffu : 0u, m1 = kv0 + 1 < VN ? 0xffff0000u : 0u;
#pragma unroll
            for (int dd = 0; dd < 16; ++dd) {
                const uint32_t w = (sel16(vr[ps][0], vr[ps][1], dd) & m0) | ((sel16(vr[ps][2], vr[ps][3], dd) << 16) & m1);
                *(uint32_t*)(sVt + (dq * 16 + dd) * VT_STRIDE + kv0) = w;
            }
        }
    }
#else
    // ---- K: [272][64] bf16, 128-byte rows, chunk ^= row & 7 ------------------------------------
    for (int s = tid; s < VNT * 16 * 8; s += VNW * 64) {
        const int row = s >> 3, c = s & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < VN) v = *(const u32x4*)(base + (size_t)row * VLD + 1024 + c * 8);
        *(u32x4*)(sK + row * 128 + ((c ^ (row & 7)) << 4)) = v;
    }
    // ---- V^T: [64 d][288 kv]; wave w transposes d = 16(w&3)..+15, lane <-> key pair; the key-pair groups
    //      pg = 0,1,2 are dealt to the wave quads (w>>2) round-robin
    const int dq = wave & 3;
    for (int pg = wave >> 2; pg < 3; pg += VNW / 4) {
        const int p = pg * 64 + lane;
        if (p < VNC * 16) {
            const int kv0 = 2 * p, kv1 = kv0 + 1;
            u32x4 a0 = {0u, 0u, 0u, 0u}, a1 = a0, b0 = a0, b1 = a0;
            if (kv0 < VN) {
                const uint16_t* r0 = base + (size_t)kv0 * VLD + 2048 + dq * 16;
                a0 = *(const u32x4*)r0;
                a1 = *(const u32x4*)(r0 + 8);
            }
            if (kv1 < VN) {
                const uint16_t* r1 = base + (size_t)kv1 * VLD + 2048 + dq * 16;
                b0 = *(const u32x4*)r1;
                b1 = *(const u32x4*)(r1 + 8);
            }
#pragma unroll
            for (int dd = 0; dd < 16; ++dd) {
                const uint32_t w = sel16(a0, a1, dd) | (sel16(b0, b1, dd) << 16);
                *(uint32_t*)(sVt + (dq * 16 + dd) * VT_STRIDE + kv0) = w;
            }
        }
    }
#endif
    VLY_VSTAMP();                                          // 2: staging stores issued
    __syncthreads();
    VLY_VSTAMP();                                          // 3: barrier passed

    const float sc = 0.125f * LOG2E;                       // 64^-0.5, folded with log2(e) for exp2
    // the Q fragments come straight from global memory: the next query tile's are requested before this tile's math
    // (a wave handles 2-3 query tiles; an exposed global round trip per tile was ~1/4 of the kernel)
#if !VLY_VIT_STAGE2
    bf16x8 qn[2];
    {
        const int qc0 = min(wave * 16 + l15, VN - 1);
        qn[0] = *(const bf16x8*)(base + (size_t)qc0 * VLD + g * 8);
        qn[1] = *(const bf16x8*)(base + (size_t)qc0 * VLD + 32 + g * 8);
    }
#endif
#if VLY_VIT_STAGGER
    if (wave >= 4) __builtin_amdgcn_s_sleep(VLY_VIT_STAGGER);
#endif
#if VLY_VIT_ROT
    const int rot = (blockIdx.x ^ (blockIdx.x >> 3) ^ (blockIdx.x >> 8)) & (VNW - 1);
#else
    const int rot = 0;
#endif
    // wave w takes query tiles w, w + 8 and — one wave of the workgroup — tile 16 (the 257th query)
    const auto next_tile = [&](int qt) { return qt + VNW < 16 ? qt + VNW : (qt < 16 && wave == rot ? 16 : VNT); };
    for (int qt = wave; qt < VNT; qt = next_tile(qt)) {
        const int q = qt * 16 + l15;
        bf16x8 qf[2] = {qn[0], qn[1]};
        if (next_tile(qt) < VNT) {
            const int qc1 = min(next_tile(qt) * 16 + l15, VN - 1);
            qn[0] = *(const bf16x8*)(base + (size_t)qc1 * VLD + g * 8);
            qn[1] = *(const bf16x8*)(base + (size_t)qc1 * VLD + 32 + g * 8);
        }

        // (scale, then max / exp2(s - m).  Folding the scale into the exponent's FMA with a max3 chain — fewer VALU instructions —
        // measured 4 % SLOWER: 113.4 vs 109.0 us at 128 frames, round 2.)
        f32x4 s[VNT];
#if VLY_VIT_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int t = 0; t < VNT; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 kf = *(const bf16x8*)(sK + (t * 16 + l15) * 128 + (((kk * 4 + g) ^ (l15 & 7)) << 4));
                acc = mfma16(kf, qf[kk], acc);
            }
#if VLY_VIT_FOLD
            s[t] = acc;
#else
            s[t] = acc * sc;
#endif
            if ((t & 3) == 3) asm volatile("" ::: "memory");     // cap the K-fragment reads in flight (VGPR budget: 2 blocks/CU)
        }
#if VLY_VIT_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        // keys 257..271 are padding: only (g == 0, r == 0) of the last tile is real
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (g != 0 || r != 0) s[VNT - 1][r] = NEG_BIG;
#if VLY_VIT_TIMING
        asm volatile("" : "+v"(s[VNT - 1][0]));                // (the stamp after the first product: all of it issued; s_memtime does not wait for the MFMAs)
        VLY_VSTAMP();                                        // 4 + 4 i: QK^T issued
#endif

        float m = NEG_BIG;
        float l = 0.f;
#if VLY_VIT_FOLD
#pragma unroll
        for (int t = 0; t < VNT; ++t) {
            m = __builtin_fmaxf(__builtin_fmaxf(m, s[t][0]), s[t][1]);        // v_max3_f32
            m = __builtin_fmaxf(__builtin_fmaxf(m, s[t][2]), s[t][3]);
        }
        m = rows4_max(m);
        {
            const float nm = -m * sc;
            float l2 = 0.f;
#pragma unroll
            for (int t = 0; t < VNT; ++t) {
                const float p0 = sm_exp2(__builtin_fmaf(s[t][0], sc, nm)), p1 = sm_exp2(__builtin_fmaf(s[t][1], sc, nm));
                const float p2 = sm_exp2(__builtin_fmaf(s[t][2], sc, nm)), p3 = sm_exp2(__builtin_fmaf(s[t][3], sc, nm));
                s[t] = f32x4{p0, p1, p2, p3};
                l += p0 + p1;
                l2 += p2 + p3;
            }
            l += l2;
        }
#else
#pragma unroll
        for (int t = 0; t < VNT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, s[t][r]);
        m = rows4_max(m);
#pragma unroll
        for (int t = 0; t < VNT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = sm_exp2(s[t][r] - m);
                s[t][r] = p;
                l += p;
            }
#endif
        l = rows4_sum(l);

#if VLY_VIT_TIMING
        asm volatile("" : "+v"(l));
        VLY_VSTAMP();                                        // 5 + 4 i: softmax done
#endif
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#if VLY_VIT_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int c = 0; c < VNC; ++c) {
            u32x4 pk;
            pk[0] = pack_h2(s[2 * c][0], s[2 * c][1]);
            pk[1] = pack_h2(s[2 * c][2], s[2 * c][3]);
            if (2 * c + 1 < VNT) {
                pk[2] = pack_h2(s[2 * c + 1][0], s[2 * c + 1][1]);
                pk[3] = pack_h2(s[2 * c + 1][2], s[2 * c + 1][3]);
            } else {
                pk[2] = 0u;
                pk[3] = 0u;
            }
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const uint16_t* vp = sVt + (dt * 16 + l15) * VT_STRIDE + 32 * c + 4 * g;
                const u32x2 lo = *(const u32x2*)vp;
                const u32x2 hi = *(const u32x2*)(vp + hi16);
                u32x4 vv;
                vv[0] = lo[0]; vv[1] = lo[1]; vv[2] = hi[0]; vv[3] = hi[1];
                o[dt] = mfma16(__builtin_bit_cast(bf16x8, vv), pf, o[dt]);
            }
            asm volatile("" ::: "memory");
        }
#if VLY_VIT_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#if VLY_VIT_TIMING
        asm volatile("" : "+v"(o[3][3]));
        VLY_VSTAMP();                                        // 6 + 4 i: PV issued
#endif
        {   // (every lane takes part in the lane swaps; rows past token 256 write to the clamped row's twin and are masked)
            const float inv = 1.f / l;
            uint16_t* op = out + ((size_t)f * VN + min(q, VN - 1)) * 1024 + h * 64;
            {   // 32-byte runs per lane after a 4 x 4 lane transpose (8-byte pieces per lane measured 3 % slower: r03_ab_vit_store.jsonl)
                uint32_t y[2][4];
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    uint32_t d[4];
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) d[dt] = pack_h2(o[dt][2 * w] * inv, o[dt][2 * w + 1] * inv);
                    const auto p01 = __builtin_amdgcn_permlane16_swap(d[0], d[1], false, false);
                    const auto p23 = __builtin_amdgcn_permlane16_swap(d[2], d[3], false, false);
                    const auto q0 = __builtin_amdgcn_permlane32_swap(p01[0], p23[0], false, false);
                    const auto q1 = __builtin_amdgcn_permlane32_swap(p01[1], p23[1], false, false);
                    y[w][0] = q0[0]; y[w][1] = q1[0]; y[w][2] = q0[1]; y[w][3] = q1[1];
                }
#if VLY_VIT_STORE_LINES
                // Whole 128-byte lines per store instruction (round 6): a lane's 32-byte run is two 16-byte pieces A (d = 16 g ..) and
                // B (d = 16 g + 8 ..); B goes to the lane of row l15 ^ 8 (row_ror:8), and then instruction 1 writes rows 0-7 of the
                // tile (lanes l15 < 8: their A; lanes l15 >= 8: the B of row l15 - 8) and instruction 2 rows 8-15 — 8 rows x 128 B
                // each instead of 16 rows x 64 B twice.  Same bytes, same values.
                const u32x4 pa = {y[0][0], y[1][0], y[0][1], y[1][1]}, pb = {y[0][2], y[1][2], y[0][3], y[1][3]};
                u32x4 px;
#pragma unroll
                for (int i = 0; i < 4; ++i) px[i] = __builtin_amdgcn_update_dpp(0u, pb[i], 0x128, 0xf, 0xf, false);     // row_ror:8
                const bool lo8 = l15 < 8;
                const int q1 = lo8 ? q : q - 8, q2 = lo8 ? q + 8 : q;
                uint16_t* ob = out + (size_t)f * VN * 1024 + h * 64 + 16 * g;
                if (q1 < VN) *(u32x4*)(ob + (size_t)q1 * 1024 + (lo8 ? 0 : 8)) = lo8 ? pa : px;
                if (q2 < VN) *(u32x4*)(ob + (size_t)q2 * 1024 + (lo8 ? 8 : 0)) = lo8 ? px : pa;
                (void)op;
#else
                if (q < VN) {
                    u32x4* dst = (u32x4*)(op + 16 * g);
                    dst[0] = u32x4{y[0][0], y[1][0], y[0][1], y[1][1]};
                    dst[1] = u32x4{y[0][2], y[1][2], y[0][3], y[1][3]};
                }
#endif
            }
        }
        VLY_VSTAMP();                                        // 7 + 4 i: stores issued
    }
#if VLY_VIT_TIMING
    if (ts_row && lane == 0 && tsn < VTS_N) ts_row[VTS_N - 1] = (unsigned long long)tsn;
#endif
}
#undef VLY_VSTAMP

typedef __attribute__((ext_vector_type(4))) short s16x4;
#include "attention_vit_persist.inc"
#include "attention_vit_pp.inc"
// ---------------------------------------------------------------------------------------------
// Llama attention over the KV cache (head_dim 128, causal + key-validity mask, online softmax)
// grid = (heads, B, ceil(S/(16*LNW))), last query block first; LNW waves x 16 query rows.
// ---------------------------------------------------------------------------------------------

// LNW waves x 16 query rows per workgroup: with 8 waves a K/V tile staged into LDS serves 128 queries (half the
// redundant tile loads and transposes of the 64-query version) and a CU holds 16 waves.
#ifndef VLY_LNW
#define VLY_LNW 8               // 4 or 8 (A/B: tools/ab_lib.py build x --src attention.hip -DVLY_LNW=4; run-attn)
#endif
constexpr int LNW = VLY_LNW;

// ---- round 3: the same attention with K / V tiles by LDS-DMA, V through ds_read_b64_tr_b16, two workgroups per CU ---------
// llama_attn_kernel holds 160-222 registers (one 8-wave workgroup per CU; PMC: half of all wave cycles parked) and pays two
// barriers per 64-key tile because its tiles pass through registers (K) and a register transpose (V).  Here a tile is 16 + 16
// one-KB LDS-DMA pieces (asm buffer_load ... lds, M0 = destination: no registers, no VALU), K swizzled on the SOURCE address
// (chunk ^= row & 15, the layout the S^T fragment reads expect), V ROW-major with its 32-byte chunk pairs XORed by row & 7 and
// read as the second product's A operand by the hardware transpose read (vit_attn_persist_kernel's scheme at 256-byte rows: the 8
// rows of a half wave land in 8 distinct 32-byte slots of the 256-byte bank window).  Two buffers: tile t + 1 flies under
// tile t, ONE barrier per tile.  No staging registers -> <= 128 per lane -> two workgroups per CU: one workgroup's start-up
// (Q fragments, first tile) hides under the other's tiles.  Same arithmetic, same order as llama_attn_kernel: bit-identical.
VLY_DEVICE bf16x8 vt_frag256(const char* lo_addr) {          // two transpose reads: keys +0..3 and +16..19 of this lane group's rows
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo_addr));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo_addr + 16 * 256));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

constexpr int L2_TILE = 64 * 256;           // one K or V tile: 64 keys x 128 d x 2 B
constexpr int L2_BUF = 2 * L2_TILE;

__global__ void __launch_bounds__(LNW * 64, 4) llama_attn2_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ kc,
                                                                 const uint16_t* __restrict__ vc, const uint8_t* __restrict__ key_valid,
                                                                 uint16_t* __restrict__ out, int S, int heads, int past,
                                                                 const int32_t* __restrict__ past_dev, int kv_stride, int ctx_max) {
    static_assert(LNW == 8, "16 + 16 pieces on 8 waves");
    __shared__ __attribute__((aligned(16))) char smem[2 * L2_BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, b = blockIdx.y, qb = (int)gridDim.z - 1 - (int)blockIdx.z;      // longest block first
    const int Hq = heads * 128;
    if (past_dev) past = min(*past_dev, ctx_max - S);
    const int kv_len = past + S;

    const int q = qb * (LNW * 16) + wave * 16 + l15;      // query row inside this call
    const int qc = min(q, S - 1);
    const int qpos = past + q;                            // absolute position: keys <= qpos are visible
    const int q_last = min(qb * (LNW * 16) + LNW * 16 - 1, S - 1);
    const int wave_qpos_max = past + min(qb * (LNW * 16) + wave * 16 + 15, S - 1);   // tiles beyond it are fully masked for this wave
    const int kv_end = min(past + q_last + 1, kv_len);
    const int ntiles = (kv_end + 63) >> 6;

    // ---- staging: wave w issues K pieces 2w, 2w + 1 and V pieces 2w, 2w + 1 of a tile (piece p = rows 4p .. 4p + 3)
    // the descriptors start at THIS (batch, head)'s rows (a 64-bit, workgroup-uniform base): the 32-bit lane offsets below only
    // span one head's ctx_max x 256 bytes, so the cache may be of any size (configs[3]'s replicated prefill holds 18 GB;
    // round 3 offset from the cache base and fell back to llama_attn_kernel at >= 4 GB)
    const size_t head_off = ((size_t)b * heads + h) * (size_t)ctx_max * 128;
    const __amdgpu_buffer_rsrc_t rsK = vly_rsrc(kc + head_off), rsV = vly_rsrc(vc + head_off);
    const __amdgpu_buffer_rsrc_t rsM = vly_rsrc(key_valid ? key_valid + (size_t)b * kv_stride : key_valid);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t hb = 0u;
    uint32_t okraw = 1u;
    bool okin = false;
    auto stage = [&](int kt, int buf) {
        const int kv0 = kt * 64;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int pc = 2 * wave + pp, row = 4 * pc + (lane >> 4), cp = lane & 15;
            const uint32_t rb = hb + (uint32_t)min(kv0 + row, kv_len - 1) * 256u;          // (rows past kv_len re-read the last key: masked)
            const uint32_t voK = rb + (uint32_t)((cp ^ (row & 15)) << 4);
            const uint32_t voV = rb + (uint32_t)((cp ^ ((row & 7) << 1)) << 4);
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)buf * L2_BUF + (uint32_t)pc * 1024u);
            asm volatile("s_mov_b32 m0, %0" ::"s"(dst) : "memory");
            asm volatile("s_nop 0" ::: "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voK), "s"(rsK) : "memory");
            asm volatile("s_mov_b32 m0, %0" ::"s"(dst + (uint32_t)L2_TILE) : "memory");
            asm volatile("s_nop 0" ::: "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voV), "s"(rsV) : "memory");
        }
        const int kvl = kv0 + lane;                                    // (a buffer load: no 64-bit per-lane pointer to keep alive)
        okin = kvl < kv_len;                                           // (the loaded byte is only looked at at the ballot: no wait for it here)
        if (key_valid) okraw = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rsM, (uint32_t)min(kvl, kv_len - 1), 0, 0);
    };
    // per-lane byte offset of the V fragment of d tile dt inside a V tile image: row 4g + t (t = l15 >> 2), 8-byte half l15 & 1,
    // chunk (2 dt + ((l15 & 3) >> 1)) ^ ((row & 7) << 1); + 8192 c per 32-key chunk (+ 4096 = 16 rows for the second read)
    // = ((dt ^ (row & 7)) << 5) | (((l15 & 3) >> 1) << 4): one base + one XOR term per lane; both go through opaque_i32 at
    // the top of every tile, or hipcc precomputes the eight per-dt offsets outside the loop and spills to hold them
    const int vr = 4 * g + (l15 >> 2);
    const int vbase0 = L2_TILE + vr * 256 + (l15 & 1) * 8 + (((l15 & 3) >> 1) << 4), vx0 = (vr & 7) << 5;

    if (ntiles > 0) stage(0, 0);
    const uint16_t* qp = qkv + ((size_t)b * S + qc) * 3 * Hq + h * 128;
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8*)(qp + kk * 32 + g * 8);

    const float sc = 0.08838834764831845f * LOG2E;        // 128^-0.5 * log2(e)
    float m = NEG_BIG, l = 0.f;
    f32x4 o[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < ntiles; ++kt) {
        const int kv0 = kt * 64, buf = kt & 1;
        const char* sK = smem + buf * L2_BUF;
        // this wave's pieces of tile kt (and its Q fragments, and the key-validity byte) have landed — the BUILTIN wait, so
        // that hipcc does not add its own vmcnt(0) behind the next tile's DMA issue —, then everyone's, and everyone is done
        // with the other buffer
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        const unsigned long long vmask = __ballot(okin && (okraw & 0xffu) != 0u);
        if (kt + 1 < ntiles) stage(kt + 1, buf ^ 1);      // in flight while this tile is multiplied
        if (kv0 > wave_qpos_max) continue;                // every key of this tile is in this wave's future (wave-uniform)
        const int vbase = opaque_i32(vbase0), vx = opaque_i32(vx0);

        // ---- S^T tile ---------------------------------------------------------------------------
        f32x4 s[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 kf = *(const bf16x8*)(sK + (t * 16 + l15) * 256 + (((kk * 4 + g) ^ l15) << 4));
                acc = mfma16(kf, qf[kk], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kl = t * 16 + 4 * g + r;
                const bool vis = ((vmask >> kl) & 1ull) && (kv0 + kl <= qpos);
                s[t][r] = vis ? acc[r] * sc : NEG_BIG;
            }
        }
        // ---- online softmax ------------------------------------------------------------------------
        float rm = NEG_BIG;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) rm = fmaxf(rm, s[t][r]);
        rm = rows4_max(rm);
        const float mn = fmaxf(m, rm);
        const float alpha = sm_exp2(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = sm_exp2(s[t][r] - mn);
                s[t][r] = p;
                ps += p;
            }
        l = l * alpha + ps;                                // per-lane partial; the 4 g-lanes share alpha
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) o[dt] *= alpha;
        // ---- O^T += V^T P^T ------------------------------------------------------------------------
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            u32x4 pk;
            pk[0] = pack_h2(s[2 * c][0], s[2 * c][1]);
            pk[1] = pack_h2(s[2 * c][2], s[2 * c][3]);
            pk[2] = pack_h2(s[2 * c + 1][0], s[2 * c + 1][1]);
            pk[3] = pack_h2(s[2 * c + 1][2], s[2 * c + 1][3]);
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
            for (int dt = 0; dt < 8; ++dt)
                o[dt] = mfma16(vt_frag256(sK + vbase + ((dt << 5) ^ vx) + c * 8192), pf, o[dt]);
        }
    }
    l = rows4_sum(l);
    if (q < S) {
        const float inv = 1.f / l;
        uint16_t* op = out + ((size_t)b * S + q) * Hq + h * 128 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            u32x2 pk;
            pk[0] = pack_h2(o[dt][0] * inv, o[dt][1] * inv);
            pk[1] = pack_h2(o[dt][2] * inv, o[dt][3] * inv);
            *(u32x2*)(op + dt * 16) = pk;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Decode attention (S = 1): one workgroup per (head, batch), HBM-bound on the KV cache
// (kv_len * 512 bytes per head).  Phase 1: thread <-> key, q.k over 128 d with 16-byte K loads, scores
// to LDS; phase 2: block max / sum; phase 3: 16 threads share a V row (8 d each), 16 keys per pass,
// fp32 partial sums reduced through LDS.
// ---------------------------------------------------------------------------------------------
constexpr int DEC_MAX_CTX = 8192;

__global__ void __launch_bounds__(256) decode_attn_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ kc,
                                                          const uint16_t* __restrict__ vc, const uint8_t* __restrict__ key_valid,
                                                          uint16_t* __restrict__ out, int heads, int past,
                                                          const int32_t* __restrict__ past_dev, int kv_stride, int ctx_max) {
    __shared__ float sc[DEC_MAX_CTX];
    __shared__ __attribute__((aligned(16))) float qs[128];
    __shared__ float red[8];
    __shared__ __attribute__((aligned(16))) float acc_s[16][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int Hq = heads * 128;
    if (past_dev) past = min(*past_dev, ctx_max - 1);
    const int kv_len = past + 1;
    const uint16_t* qp = qkv + (size_t)b * 3 * Hq + h * 128;
    if (tid < 128) qs[tid] = h2f(qp[tid]) * (0.08838834764831845f * LOG2E);
    __syncthreads();
    const uint16_t* kbase = kc + ((size_t)b * heads + h) * ctx_max * 128;
    const uint16_t* vbase = vc + ((size_t)b * heads + h) * ctx_max * 128;
    const uint8_t* kvld = key_valid ? key_valid + (size_t)b * kv_stride : nullptr;

    float m = NEG_BIG;
    for (int j = tid; j < kv_len; j += 256) {
        float s = NEG_BIG;
        if (!kvld || kvld[j]) {
            const u32x4* kr = (const u32x4*)(kbase + (size_t)j * 128);
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const u32x4 kk = kr[c];
                const f32x4 q0 = *(const f32x4*)(qs + 8 * c), q1 = *(const f32x4*)(qs + 8 * c + 4);
                a = fmaf(h_lo(kk[0]), q0[0], a); a = fmaf(h_hi(kk[0]), q0[1], a);
                a = fmaf(h_lo(kk[1]), q0[2], a); a = fmaf(h_hi(kk[1]), q0[3], a);
                a = fmaf(h_lo(kk[2]), q1[0], a); a = fmaf(h_hi(kk[2]), q1[1], a);
                a = fmaf(h_lo(kk[3]), q1[2], a); a = fmaf(h_hi(kk[3]), q1[3], a);
            }
            s = a;
        }
        sc[j] = s;
        m = fmaxf(m, s);
    }
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float l = 0.f;
    for (int j = tid; j < kv_len; j += 256) {
        const float p = sm_exp2(sc[j] - m);
        sc[j] = p;
        l += p;
    }
    l = wave_sum(l);
    if (lane == 0) red[4 + wave] = l;
    __syncthreads();
    l = red[4] + red[5] + red[6] + red[7];

    // phase 3: thread = (key group kg = tid >> 4, d chunk dc = tid & 15 -> d = 8*dc .. 8*dc+7)
    const int kg = tid >> 4, dc = tid & 15;
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
    for (int j0 = kg; j0 < kv_len; j0 += 64) {                  // 4 independent 16-byte V loads in flight per lane
        u32x4 vv[4];
        float pp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 16 * u;
            const bool in = j < kv_len;
            pp[u] = in ? sc[j] : 0.f;
            vv[u] = *(const u32x4*)(vbase + (size_t)(in ? j : kg) * 128 + 8 * dc);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float p = pp[u];
            o[0] = fmaf(p, h_lo(vv[u][0]), o[0]); o[1] = fmaf(p, h_hi(vv[u][0]), o[1]);
            o[2] = fmaf(p, h_lo(vv[u][1]), o[2]); o[3] = fmaf(p, h_hi(vv[u][1]), o[3]);
            o[4] = fmaf(p, h_lo(vv[u][2]), o[4]); o[5] = fmaf(p, h_hi(vv[u][2]), o[5]);
            o[6] = fmaf(p, h_lo(vv[u][3]), o[6]); o[7] = fmaf(p, h_hi(vv[u][3]), o[7]);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc_s[kg][8 * dc + i] = o[i];
    __syncthreads();
    if (tid < 128) {
        float t = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) t += acc_s[k2][tid];
        out[(size_t)b * Hq + h * 128 + tid] = f2h(t / l);
    }
}

// ---------------------------------------------------------------------------------------------
// One-token decode step of a layer's attention, fused: RoPE on the new q and k, KV-cache append of the
// rotated k and of v, and the attention of the new query over positions 0..pos.  One 512-thread workgroup
// per (head, batch).  Keys are processed in chunks of 512 with an online softmax; per chunk every thread
// issues ALL its loads up front (16 x 16 B of V for its (key group, d chunk), then the 256-byte K row of
// its key) so that a chunk costs one memory latency instead of one per 64 keys (the unfused kernel's V
// loop: 13 us at kv_len ~ 400, almost all of it exposed latency), and the separate rope/append kernel
// (4.7 us + a kernel boundary per layer) disappears.  Arithmetic identical to vly_rope_kv followed by
// vly_llama_attention (same bf16 rounding of the rotated q/k, same dot-product order).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) decode_fused_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ kc,
                                                           uint16_t* __restrict__ vc, const float* __restrict__ cos_t,
                                                           const float* __restrict__ sin_t, const uint8_t* __restrict__ key_valid,
                                                           uint16_t* __restrict__ out, int heads, int past,
                                                           const int32_t* __restrict__ past_dev, int kv_stride, int ctx_max,
                                                           int past_row_stride) {
    __shared__ float sc[512];
    __shared__ __attribute__((aligned(16))) float qs[128];
    __shared__ __attribute__((aligned(16))) float knew[128];
    __shared__ __attribute__((aligned(16))) float vnew[128];
    __shared__ float red[16];
    __shared__ __attribute__((aligned(16))) float acc_s[32][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int Hq = heads * 128;
    // device-side position: one value for the batch (captured decode step) or one per row (continuous batching:
    // every slot of the batch is its own sequence at its own position)
    if (past_dev) past = min(past_dev[(size_t)b * past_row_stride], ctx_max - 1);
    const int pos = past, kv_len = past + 1;
    const uint16_t* qp = qkv + (size_t)b * 3 * Hq + h * 128;
    uint16_t* kbase = kc + ((size_t)b * heads + h) * ctx_max * 128;
    uint16_t* vbase = vc + ((size_t)b * heads + h) * ctx_max * 128;
    const uint8_t* kvld = key_valid ? key_valid + (size_t)b * kv_stride : nullptr;

    // ---- RoPE on q, k (pair d, d+64) + append; v append -------------------------------------------------
    if (tid < 64) {
        const float cs = cos_t[(size_t)pos * 64 + tid], sn = sin_t[(size_t)pos * 64 + tid];
        const float q0 = h2f(qp[tid]), q1 = h2f(qp[tid + 64]);
        const float k0 = h2f(qp[Hq + tid]), k1 = h2f(qp[Hq + tid + 64]);
        const float scale = 0.08838834764831845f * LOG2E;
        qs[tid] = h2f(f2h(rope_rot(q0, q1, cs, sn, -1.f))) * scale;
        qs[tid + 64] = h2f(f2h(rope_rot(q1, q0, cs, sn, 1.f))) * scale;
        const uint16_t r0 = f2h(rope_rot(k0, k1, cs, sn, -1.f)), r1 = f2h(rope_rot(k1, k0, cs, sn, 1.f));
        knew[tid] = h2f(r0);
        knew[tid + 64] = h2f(r1);
        kbase[(size_t)pos * 128 + tid] = r0;
        kbase[(size_t)pos * 128 + tid + 64] = r1;
    } else if (tid < 192) {
        const int d = tid - 64;
        const uint16_t v = qp[2 * Hq + d];
        vnew[d] = h2f(v);
        vbase[(size_t)pos * 128 + d] = v;
    }
    __syncthreads();

    const int kg = tid >> 4, dc = tid & 15;                      // V role: key group (32), d chunk (8 dims)
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;

    for (int c0 = 0; c0 < kv_len; c0 += 512) {
        // ---- issue this chunk's V loads (keys c0 + kg + 32 u), then the K row of key c0 + tid -------------
        u32x4 vv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int j = c0 + kg + 32 * u;
            vv[u] = (j < pos) ? *(const u32x4*)(vbase + (size_t)j * 128 + 8 * dc) : u32x4{0u, 0u, 0u, 0u};
        }
        const int jk = c0 + tid;
        float s = NEG_BIG;
        if (jk < kv_len && (!kvld || kvld[jk])) {
            float a = 0.f;
            if (jk < pos) {
                const u32x4* kr = (const u32x4*)(kbase + (size_t)jk * 128);
                u32x4 kk[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) kk[c] = kr[c];
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const f32x4 q0 = *(const f32x4*)(qs + 8 * c), q1 = *(const f32x4*)(qs + 8 * c + 4);
                    a = fmaf(h_lo(kk[c][0]), q0[0], a); a = fmaf(h_hi(kk[c][0]), q0[1], a);
                    a = fmaf(h_lo(kk[c][1]), q0[2], a); a = fmaf(h_hi(kk[c][1]), q0[3], a);
                    a = fmaf(h_lo(kk[c][2]), q1[0], a); a = fmaf(h_hi(kk[c][2]), q1[1], a);
                    a = fmaf(h_lo(kk[c][3]), q1[2], a); a = fmaf(h_hi(kk[c][3]), q1[3], a);
                }
            } else {                                             // the new token's key: still in LDS
#pragma unroll 8
                for (int d = 0; d < 128; ++d) a = fmaf(knew[d], qs[d], a);
            }
            s = a;
        }
        // ---- online softmax over the chunk --------------------------------------------------------------
        float mc = wave_max(s);
        if (lane == 0) red[wave] = mc;
        __syncthreads();
        mc = red[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) mc = fmaxf(mc, red[w]);
        const float m_new = fmaxf(m_run, mc);
        const float alpha = sm_exp2(m_run - m_new);
        const float p = s > 0.5f * NEG_BIG ? sm_exp2(s - m_new) : 0.f;   // masked keys never count, even in an all-masked chunk
        sc[tid] = p;
        float lc = wave_sum(p);
        if (lane == 0) red[8 + wave] = lc;
        __syncthreads();                                         // sc[] and red[8..] visible
        lc = red[8];
#pragma unroll
        for (int w = 1; w < 8; ++w) lc += red[8 + w];
        l_run = vly_mul_add(l_run, alpha, lc);
        m_run = m_new;
        // ---- P.V for this chunk ---------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] *= alpha;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int jl = kg + 32 * u;                          // key index inside the chunk
            const float pj = sc[jl];
            if (c0 + jl == pos) {                                // the new token's value: from LDS
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = fmaf(pj, vnew[8 * dc + i], o[i]);
            } else {
                o[0] = fmaf(pj, h_lo(vv[u][0]), o[0]); o[1] = fmaf(pj, h_hi(vv[u][0]), o[1]);
                o[2] = fmaf(pj, h_lo(vv[u][1]), o[2]); o[3] = fmaf(pj, h_hi(vv[u][1]), o[3]);
                o[4] = fmaf(pj, h_lo(vv[u][2]), o[4]); o[5] = fmaf(pj, h_hi(vv[u][2]), o[5]);
                o[6] = fmaf(pj, h_lo(vv[u][3]), o[6]); o[7] = fmaf(pj, h_hi(vv[u][3]), o[7]);
            }
        }
        __syncthreads();                                         // sc[] / red[] are rewritten by the next chunk
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc_s[kg][8 * dc + i] = o[i];
    __syncthreads();
    if (tid < 128) {
        float t = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < 32; ++k2) t += acc_s[k2][tid];
        out[(size_t)b * Hq + h * 128 + tid] = f2h(t / l_run);
    }
}

// ---- decode attention split over the keys (flash-decoding) ----------------------------------------------------------------
// ---- vly_decode_attention_merged: the merge of a head's split partials by the LAST of its workgroups to finish -------------
// (round 4) vly_gemv_attnmerge_bf16 merges the partials in the o projection's prologue: every one of its 256 workgroups reads all
// heads x splits x 132 floats and spends ~4 us before its first dot product — on a launch whose weights are 7 us of HBM time.
// Here every split publishes its partial with write-through stores, drains them, and takes a ticket on the head's counter; the
// one that draws the last ticket merges (the SAME arithmetic, in split order: bit-identical) and writes the head's 128 outputs
// as 16-bit values, which the plain GEMV then reads like any activation.  No fence: write-through (sc1) stores + vmcnt(0) before
// the ticket, sc1 loads after it (the hand-off of decode_step.hip).  The counter is back at zero when the launch ends.
VLY_DEVICE void split_publish(float* p, float v, bool write_through) {
    if (write_through) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
VLY_DEVICE void split_merge_if_last(const float* __restrict__ partials, uint16_t* __restrict__ merged, unsigned* __restrict__ arrivals,
                                    int heads, int h, int b, float* flag_lds) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's partial has been written through
    __syncthreads();
    unsigned* ctr = arrivals + (size_t)b * heads + h;
    if (threadIdx.x == 0) {
#if VLY_DECODE_MERGE_FENCE
        // the memory-model form of the same hand-off (VERDICT r5 #8 / cdna_hip_programming Guideline 16): agent-scope release, the
        // wait restated where hipcc cannot drop it, THEN the ticket; the merging workgroup acquires below
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        const unsigned t = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag_lds[0] = t == VLY_DECODE_SPLITS - 1 ? 1.f : 0.f;
        if (t == VLY_DECODE_SPLITS - 1) {
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if VLY_DECODE_MERGE_FENCE
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        }
    }
    __syncthreads();
    if (flag_lds[0] == 0.f || threadIdx.x >= 32) return;
    // 32 lanes x 4 dims: the unit arithmetic of gemv_norm_kernel<.., PRO = 1>, operation for operation
    const float* hb = partials + ((size_t)b * heads + h) * (VLY_DECODE_SPLITS * 132);
    float ms[VLY_DECODE_SPLITS], ls[VLY_DECODE_SPLITS];
    float os[VLY_DECODE_SPLITS][4];
#pragma unroll
    for (int sp = 0; sp < VLY_DECODE_SPLITS; ++sp) {
        ms[sp] = __hip_atomic_load(hb + sp * 132, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ls[sp] = __hip_atomic_load(hb + sp * 132 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            os[sp][r] = __hip_atomic_load(hb + sp * 132 + 4 + 4 * (int)threadIdx.x + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float mx = ms[0];
#pragma unroll
    for (int sp = 1; sp < VLY_DECODE_SPLITS; ++sp) mx = fmaxf(mx, ms[sp]);
    float L = 0.f, O[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sp = 0; sp < VLY_DECODE_SPLITS; ++sp) {
        const float w = exp2f(ms[sp] - mx);
        L = fmaf(ls[sp], w, L);
#pragma unroll
        for (int r = 0; r < 4; ++r) O[r] = fmaf(os[sp][r], w, O[r]);
    }
    u32x2 pk;
    pk[0] = pack_h2(O[0] / L, O[1] / L);
    pk[1] = pack_h2(O[2] / L, O[3] / L);
    *(u32x2*)(merged + ((size_t)b * heads + h) * 128 + 4 * threadIdx.x) = pk;
}

// decode_fused_kernel gives a head to ONE workgroup: at batch 1 that is `heads` (40) of 256 CUs, each pulling its head's
// 2 x kv_len x 256 B of K and V through one CU's memory pipe (~10 us per layer at kv_len ~ 450 for 9 MB).  Here a head is
// VLY_DECODE_SPLITS workgroups of 256 threads, each over a 64-aligned quarter of the keys; every workgroup rotates q itself,
// the one whose range holds the new position also rotates / appends k and v.  It leaves (m, l, o[128]) — running max, sum of
// exponentials, unnormalised P·V, all fp32 — in `partials` [B][heads][SPLITS][132]; vly_gemv_attnmerge_bf16 (the o projection)
// merges them in its prologue, in split order.  Deterministic; differs from decode_fused_kernel in summation order only.
__global__ void __launch_bounds__(256) decode_split_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ kc,
                                                           uint16_t* __restrict__ vc, const float* __restrict__ cos_t,
                                                           const float* __restrict__ sin_t, const uint8_t* __restrict__ key_valid,
                                                           float* __restrict__ partials, int heads, int past,
                                                           const int32_t* __restrict__ past_dev, int kv_stride, int ctx_max,
                                                           int past_row_stride, uint16_t* __restrict__ merged,
                                                           unsigned* __restrict__ arrivals) {
    __shared__ float sc[256];
    __shared__ __attribute__((aligned(16))) float qs[128];
    __shared__ __attribute__((aligned(16))) float knew[128];
    __shared__ __attribute__((aligned(16))) float vnew[128];
    __shared__ float red[8];
    __shared__ __attribute__((aligned(16))) float acc_s[16][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // grid (heads, B, SPLITS): a head's splits are heads x B block ids apart — a multiple of 8 for every model of the path, i.e. on ONE
    // XCD (block id % 8).  gridDim.x == SPLITS marks the stress order (VLY_DECODE_SPLIT_SPREAD=1: grid (SPLITS, heads, B)): the four
    // splits of a head on four DIFFERENT XCDs, the placement the ticket hand-off must also survive (tools/decode_merge_stress.py)
    const bool spread = gridDim.z != VLY_DECODE_SPLITS;
    const int h = spread ? blockIdx.y : blockIdx.x, b = spread ? blockIdx.z : blockIdx.y, sp = spread ? blockIdx.x : blockIdx.z;
    const int Hq = heads * 128;
    if (past_dev) past = min(past_dev[(size_t)b * past_row_stride], ctx_max - 1);
    const int pos = past, kv_len = past + 1;
    const int chunk = (((kv_len + VLY_DECODE_SPLITS - 1) / VLY_DECODE_SPLITS) + 63) & ~63;
    const int lo = sp * chunk, hi = min(lo + chunk, kv_len);
    const bool owner = pos >= lo && pos < hi;                 // (exactly one split: pos = kv_len - 1)
    const uint16_t* qp = qkv + (size_t)b * 3 * Hq + h * 128;
    uint16_t* kbase = kc + ((size_t)b * heads + h) * ctx_max * 128;
    uint16_t* vbase = vc + ((size_t)b * heads + h) * ctx_max * 128;
    const uint8_t* kvld = key_valid ? key_valid + (size_t)b * kv_stride : nullptr;
    float* part = partials + (((size_t)b * heads + h) * VLY_DECODE_SPLITS + sp) * 132;
    if (lo >= hi) {                                           // an empty range (short contexts): the neutral element of the merge
        if (tid < 132) split_publish(part + tid, tid == 0 ? NEG_BIG : 0.f, merged != nullptr);
        if (merged) split_merge_if_last(partials, merged, arrivals, heads, h, b, sc);
        return;
    }
    // Round 4: every global load of the first pass is requested up front, BRANCH-FREE (clamped addresses, values masked where
    // they are used: behind exec-masked branches hipcc loses count of what is in flight and waits vmcnt(0)) and in the order of
    // use — the rotation's few operands first, then the K row, the validity byte and the V chunks, which only depend on the
    // position — so the rotation, its LDS hand-off and the barrier run under the K / V round trip.  Arithmetic unchanged.
    const int t63 = tid & 63;
    const float cs = cos_t[(size_t)pos * 64 + t63], sn = sin_t[(size_t)pos * 64 + t63];
    const uint16_t q0r = qp[t63], q1r = qp[t63 + 64], k0r = qp[Hq + t63], k1r = qp[Hq + t63 + 64], vr = qp[2 * Hq + ((tid - 64) & 127)];
    u32x4 vv[16], kk[16];
    uint8_t kvalid = 1;
    const int jmax = max(min(hi, pos) - 1, 0);                // last row this split may read (row 0 of the cache if there is none)
    // (round 5: a pass covers 256 keys but a split's range is often 128 — kv_len 257..512 — and then half of the 32 loads per thread
    // fetched a clamped row nobody uses: waves whose 64 keys lie past the range skip their K rows, a pass of at most 128 keys skips
    // the upper eight V chunks.  Wave- and workgroup-uniform branches; the values were masked before, the arithmetic is unchanged.)
    auto request = [&](int c0) {
        if (c0 + (tid & ~63) < hi) {                          // (wave-uniform)
            const u32x4* kr = (const u32x4*)(kbase + (size_t)min(c0 + tid, jmax) * 128);
#pragma unroll
            for (int c = 0; c < 16; ++c) kk[c] = kr[c];
        }
        if (kvld) kvalid = kvld[min(c0 + tid, hi - 1)];       // (uniform branch)
#pragma unroll
        for (int u = 0; u < 8; ++u) vv[u] = *(const u32x4*)(vbase + (size_t)min(c0 + (tid >> 4) + 16 * u, jmax) * 128 + 8 * (tid & 15));
        if (hi - c0 > 128) {                                  // (uniform)
#pragma unroll
            for (int u = 8; u < 16; ++u) vv[u] = *(const u32x4*)(vbase + (size_t)min(c0 + (tid >> 4) + 16 * u, jmax) * 128 + 8 * (tid & 15));
        } else {
#pragma unroll
            for (int u = 8; u < 16; ++u) vv[u] = u32x4{0u, 0u, 0u, 0u};
        }
    };
    request(lo);
    // ---- RoPE on q (every split), on k + append, v append (the owner): decode_fused_kernel's arithmetic
    if (tid < 64) {
        const float q0 = h2f(q0r), q1 = h2f(q1r);
        const float scale = 0.08838834764831845f * LOG2E;
        qs[tid] = h2f(f2h(rope_rot(q0, q1, cs, sn, -1.f))) * scale;
        qs[tid + 64] = h2f(f2h(rope_rot(q1, q0, cs, sn, 1.f))) * scale;
        if (owner) {
            const float k0 = h2f(k0r), k1 = h2f(k1r);
            const uint16_t r0 = f2h(rope_rot(k0, k1, cs, sn, -1.f)), r1 = f2h(rope_rot(k1, k0, cs, sn, 1.f));
            knew[tid] = h2f(r0);
            knew[tid + 64] = h2f(r1);
            kbase[(size_t)pos * 128 + tid] = r0;
            kbase[(size_t)pos * 128 + tid + 64] = r1;
        }
    } else if (tid < 192 && owner) {
        const int d = tid - 64;
        vnew[d] = h2f(vr);
        vbase[(size_t)pos * 128 + d] = vr;
    }
    __syncthreads();

    const int kg = tid >> 4, dc = tid & 15;                      // V role: key group (16), d chunk (8 dims)
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;
#pragma unroll 1
    for (int c0 = lo; c0 < hi; c0 += 256) {
        if (c0 != lo) request(c0);
#pragma unroll
        for (int u = 0; u < 16; ++u) {                          // rows past the range / the new position count as zero
            const int j = c0 + kg + 16 * u;
            if (!(j < hi && j < pos)) vv[u] = u32x4{0u, 0u, 0u, 0u};
        }
        const int jk = c0 + tid;
        float s = NEG_BIG;
        if (jk < hi && kvalid) {
            float a = 0.f;
            if (jk < pos) {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const f32x4 q0 = *(const f32x4*)(qs + 8 * c), q1 = *(const f32x4*)(qs + 8 * c + 4);
                    a = fmaf(h_lo(kk[c][0]), q0[0], a); a = fmaf(h_hi(kk[c][0]), q0[1], a);
                    a = fmaf(h_lo(kk[c][1]), q0[2], a); a = fmaf(h_hi(kk[c][1]), q0[3], a);
                    a = fmaf(h_lo(kk[c][2]), q1[0], a); a = fmaf(h_hi(kk[c][2]), q1[1], a);
                    a = fmaf(h_lo(kk[c][3]), q1[2], a); a = fmaf(h_hi(kk[c][3]), q1[3], a);
                }
            } else {                                             // the new token's key: still in LDS
#pragma unroll 8
                for (int d = 0; d < 128; ++d) a = fmaf(knew[d], qs[d], a);
            }
            s = a;
        }
        float mc = wave_max(s);
        if (lane == 0) red[wave] = mc;
        __syncthreads();
        mc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float m_new = fmaxf(m_run, mc);
        const float alpha = sm_exp2(m_run - m_new);
        const float p = s > 0.5f * NEG_BIG ? sm_exp2(s - m_new) : 0.f;   // masked keys never count, even in an all-masked chunk
        sc[tid] = p;
        float lc = wave_sum(p);
        if (lane == 0) red[4 + wave] = lc;
        __syncthreads();
        lc = red[4] + red[5] + red[6] + red[7];
        l_run = vly_mul_add(l_run, alpha, lc);
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] *= alpha;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int jl = kg + 16 * u;
            const float pj = sc[jl];
            // only the OWNER filled vnew: in another split the new position can fall into the padding of a 256-key pass
            // (kv_len = 337: split 1 covers [128, 256), c0 + jl reaches 336), where pj = 0 but 0 x stale LDS may be NaN
            if (owner && c0 + jl == pos) {
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = fmaf(pj, vnew[8 * dc + i], o[i]);
            } else {
                o[0] = fmaf(pj, h_lo(vv[u][0]), o[0]); o[1] = fmaf(pj, h_hi(vv[u][0]), o[1]);
                o[2] = fmaf(pj, h_lo(vv[u][1]), o[2]); o[3] = fmaf(pj, h_hi(vv[u][1]), o[3]);
                o[4] = fmaf(pj, h_lo(vv[u][2]), o[4]); o[5] = fmaf(pj, h_hi(vv[u][2]), o[5]);
                o[6] = fmaf(pj, h_lo(vv[u][3]), o[6]); o[7] = fmaf(pj, h_hi(vv[u][3]), o[7]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc_s[kg][8 * dc + i] = o[i];
    __syncthreads();
    if (tid < 128) {
        float t = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) t += acc_s[k2][tid];
        split_publish(part + 4 + tid, t, merged != nullptr);
    } else if (tid < 132) {
        split_publish(part + tid - 128, tid == 128 ? m_run : tid == 129 ? l_run : 0.f, merged != nullptr);
    }
    if (merged) split_merge_if_last(partials, merged, arrivals, heads, h, b, sc);
}

// Round 5, the launch at EIGHT rows (serving.ContinuousBatcher; 320 heads x 4 splits, ~61 MB of K / V per layer of the 13B step):
// 20.7 us per layer, 19.5 with the unused loads skipped (above) — 3.1 TB/s, against 23.0 for decode_fused_kernel's 320 workgroups.
// Three restructurings measured no better in the step (profiles/r05/r05_decode_b8_attention.txt) and are not kept:
//   * sixteen lanes per key for K as for V (a wave instruction reads four whole 256-byte rows instead of touching 64; 16-lane DPP
//     sum per score, probabilities stay in registers): 1273 vs 1288 tokens/s at eight requests, 215.4 vs 216.2 at one;
//   * two, three or four splits per head chosen by the launch's rounds of workgroups: 18.8 / 19.5 / 19.6 us;
//   * decode_step.hip's register-lean form (K row and V chunks in two halves, 128 registers, four workgroups per CU): 26 us.
// So neither the K access shape, nor the rounds, nor the occupancy bounds it; what is left is each workgroup's chain position ->
// K / V round trip -> two barriers -> write-through publish -> ticket, which more resident workgroups only stretch.

// ---- vly_llama_attention_probs: HF's ``output_attentions`` (round 4) -------------------------------------------------------
// The flash-style kernels above never hold a row of probabilities; a caller that asks for them (valley_model.py:281,324-330
// forwards the flag to HF's eager attention, hf:llama/modeling_llama.py:191-213) gets them from this separate pass over the
// ROTATED q (left in the q|k|v buffer by vly_rope_kv / the fused epilogue) and the K cache: one workgroup per (query, head,
// sequence), scores in LDS, fp32 softmax, fp32 out.  Not a hot path: it re-reads the K rows once per query.
template <typename T> VLY_DEVICE float probs_ld(const T* p);
template <> VLY_DEVICE float probs_ld<uint16_t>(const uint16_t* p) { return h2f(*p); }
template <> VLY_DEVICE float probs_ld<float>(const float* p) { return *p; }

template <typename T>
__global__ void __launch_bounds__(256) attn_probs_kernel(const T* __restrict__ qkv, const T* __restrict__ kc,
                                                         const uint8_t* __restrict__ key_valid, float* __restrict__ out, int S,
                                                         int heads, int past, int kv_stride, int ctx_max) {
    extern __shared__ float pr_sc[];                       // kv_len scores / probabilities
    __shared__ float qs[128];
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int kv_len = past + S, vis = past + i + 1, Hq = heads * 128;
    const T* q = qkv + ((size_t)b * S + i) * 3 * Hq + h * 128;
    const T* kb = kc + ((size_t)b * heads + h) * ctx_max * 128;
    const uint8_t* kvld = key_valid ? key_valid + (size_t)b * kv_stride : nullptr;
    if (tid < 128) qs[tid] = probs_ld(q + tid) * (0.08838834764831845f * LOG2E);
    __syncthreads();
    float m = NEG_BIG;
    for (int j = tid; j < kv_len; j += 256) {
        float s = NEG_BIG;
        if (j < vis && (!kvld || kvld[j])) {
            const T* kr = kb + (size_t)j * 128;
            float a = 0.f;
#pragma unroll 8
            for (int d = 0; d < 128; ++d) a = fmaf(probs_ld(kr + d), qs[d], a);
            s = a;
        }
        pr_sc[j] = s;
        m = fmaxf(m, s);
    }
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float l = 0.f;
    for (int j = tid; j < kv_len; j += 256) {
        const float s = pr_sc[j];
        const float p = s > 0.5f * NEG_BIG ? sm_exp2(s - m) : 0.f;
        pr_sc[j] = p;
        l += p;
    }
    l = wave_sum(l);
    if (lane == 0) red[4 + wave] = l;
    __syncthreads();
    l = red[4] + red[5] + red[6] + red[7];
    const float inv = l > 0.f ? 1.f / l : 0.f;               // a query with no visible key (left padding): a row of zeros
    float* o = out + (((size_t)b * heads + h) * S + i) * kv_len;
    for (int j = tid; j < kv_len; j += 256) o[j] = pr_sc[j] * inv;
}

// the kernel that is NOT the default of its op (A/B runs and the default's bit-identity witness: VLY_LLAMA_ATTN=1 -> llama_attn_kernel)
// lives in the EXPERIMENTAL library only (libvalley_hip_exp.so, -DVLY_EXPERIMENTAL=1, valley_amd/build.py)
#ifndef VLY_EXPERIMENTAL
#define VLY_EXPERIMENTAL 0
#endif
#if VLY_EXPERIMENTAL
#include "attention_ab.inc"
#endif

}  // namespace

#if VLY_VIT_TIMING
extern "C" int vlydbg_vit_timing_read(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vit_ts), sizeof(unsigned long long) * VTS_WG * VNW * VTS_N); }
#endif

// The persistent kernel (one 16-wave workgroup per CU walking its heads, the next head's K / V in flight by LDS-DMA: attention_vit_persist.inc)
// at every frame count: one kernel per op keeps the encode bit-identical however many frames share a launch (tests/test_scale_gpu.py::
// test_c4_shape_tower_chunk_invariance).  Against one workgroup per (frame, head) (vit_attn_kernel, rounds 2-5's default), A/B, 100
// repetitions, same box (profiles/r06/r06_vit_attn_persist_ab.txt, r06_vit_attn_persist_small.txt): 4 frames 12.4 -> 11.2 us, 32 frames
// 23.2 -> 21.7, 64 frames 46.7 -> 43.7, 128 frames 88.2 -> 72.5, 256 frames 169.1 -> 131.2.  VLY_VIT_ATTN=1 runs vit_attn_kernel (A/B runs,
// its tests), VLY_VIT_ATTN=5 the two-workgroups-per-CU form (attention_vit_pp.inc: bit-identical, measured 6 % behind); -DVLY_VIT_PERSIST_MIN=n
// builds a library that uses vit_attn_kernel below n frames.
#ifndef VLY_VIT_PERSIST_MIN
#define VLY_VIT_PERSIST_MIN 1
#endif
#ifndef VLY_VIT_PP
#define VLY_VIT_PP 0                // 1: vit_attn_pp_kernel instead of vit_attn_persist_kernel (A/B builds)
#endif
constexpr int VIT_PERSIST_MIN_FRAMES = VLY_VIT_PERSIST_MIN;
extern "C" int vly_vit_attention(const void* qkv, void* out, int F, void* stream) {
    // (out: the kernels store 16 bytes per lane since round 3)
    if (F <= 0 || ((uintptr_t)qkv & 15) || ((uintptr_t)out & 15)) { vly_set_error("vly_vit_attention: bad args F=%d (qkv and out 16-byte aligned)", F); return -22; }
    const char* sel = getenv("VLY_VIT_ATTN");           // (read per call: the tests compare the kernels inside one process)
    const int ver = sel ? atoi(sel) : 0;
    if (ver == 4 || ver == 5 || (ver != 1 && F >= VIT_PERSIST_MIN_FRAMES)) {
        static const int cus = [] {
            int dev = 0, n = 0;
            (void)hipGetDevice(&dev);
            (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
            return n > 0 ? n : 256;
        }();
        const int nheads = F * 16;
        if ((size_t)F * VN * VLD * 2 >= ((size_t)1 << 32)) { vly_set_error("vly_vit_attention: F=%d exceeds the 4 GB buffer descriptor", F); return -22; }
        if (ver == 5 || VLY_VIT_PP) {                    // two persistent 8-wave workgroups per CU (attention_vit_pp.inc)
            hipLaunchKernelGGL(vit_attn_pp_kernel, dim3(nheads < 2 * cus ? nheads : 2 * cus), dim3(PPW * 64), 0, (hipStream_t)stream,
                               (const uint16_t*)qkv, (uint16_t*)out, nheads);
            return vly_check_launch("vly_vit_attention");
        }
        hipLaunchKernelGGL(vit_attn_persist_kernel, dim3(nheads < cus ? nheads : cus), dim3(VPW * 64), 0, (hipStream_t)stream,
                           (const uint16_t*)qkv, (uint16_t*)out, nheads);
        return vly_check_launch("vly_vit_attention");
    }
    hipLaunchKernelGGL(vit_attn_kernel, dim3(F * 16), dim3(VNW * 64), 0, (hipStream_t)stream, (const uint16_t*)qkv, (uint16_t*)out);
    return vly_check_launch("vly_vit_attention");
}

extern "C" int vly_llama_attention(const void* qkv, const void* kcache, const void* vcache, const uint8_t* key_valid,
                                   int key_valid_stride, void* out, int B, int S, int heads, int past_len,
                                   const int32_t* past_len_dev, int ctx_max, void* stream) {
    if (B <= 0 || S <= 0 || heads <= 0 || past_len < 0 || past_len + S > ctx_max || B > 65535 || heads > 65535 ||
        ((uintptr_t)qkv & 15) || ((uintptr_t)kcache & 15) || ((uintptr_t)vcache & 15) || ((uintptr_t)out & 7)) {
        vly_set_error("vly_llama_attention: bad args B=%d S=%d heads=%d past=%d ctx_max=%d", B, S, heads, past_len, ctx_max);
        return -22;
    }
    if (key_valid && key_valid_stride < past_len + S) {
        vly_set_error("vly_llama_attention: key_valid_stride %d < kv_len %d", key_valid_stride, past_len + S);
        return -22;
    }
    if (S == 1 && ctx_max <= DEC_MAX_CTX) {
        hipLaunchKernelGGL(decode_attn_kernel, dim3(heads, B), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)qkv,
                           (const uint16_t*)kcache, (const uint16_t*)vcache, key_valid, (uint16_t*)out, heads, past_len,
                           past_len_dev, key_valid_stride, ctx_max);
        return vly_check_launch("vly_llama_attention(decode)");
    }
    // llama_attn2_kernel (LDS-DMA tiles, two workgroups per CU: 59.3 -> 45.0 us per 13B layer at B = 8, S = 336; 77 -> 53 at S = 1024,
    // profiles/history/r03/r03_llama_attn2.txt).  VLY_LLAMA_ATTN=1 keeps the register-staged llama_attn_kernel (A/B runs and its tests).
#if VLY_EXPERIMENTAL
    static const int ver = getenv("VLY_LLAMA_ATTN") ? atoi(getenv("VLY_LLAMA_ATTN")) : 2;
#else
    constexpr int ver = 2;
#endif
    if ((size_t)ctx_max * 256 >= ((size_t)1 << 32)) {
        vly_set_error("vly_llama_attention: ctx_max %d exceeds the 4 GB a (batch, head) cache row may span", ctx_max);
        return -22;
    }
    if (ver == 2) {                                                       // (per-head descriptors: no limit on the cache as a whole)
        hipLaunchKernelGGL(llama_attn2_kernel, dim3(heads, B, (S + 16 * LNW - 1) / (16 * LNW)), dim3(LNW * 64), 0, (hipStream_t)stream,
                           (const uint16_t*)qkv, (const uint16_t*)kcache, (const uint16_t*)vcache, key_valid, (uint16_t*)out,
                           S, heads, past_len, past_len_dev, key_valid_stride, ctx_max);
        return vly_check_launch("vly_llama_attention");
    }
#if VLY_EXPERIMENTAL
    hipLaunchKernelGGL(llama_attn_kernel, dim3(heads, B, (S + 16 * LNW - 1) / (16 * LNW)), dim3(LNW * 64), 0, (hipStream_t)stream,
                       (const uint16_t*)qkv, (const uint16_t*)kcache, (const uint16_t*)vcache, key_valid, (uint16_t*)out,
                       S, heads, past_len, past_len_dev, key_valid_stride, ctx_max);
    return vly_check_launch("vly_llama_attention");
#else
    return -22;                                                           // (unreachable: ver == 2)
#endif
}

extern "C" int vly_decode_attention(const void* qkv, void* kcache, void* vcache, const float* cos_table, const float* sin_table,
                                    const uint8_t* key_valid, int key_valid_stride, void* out, int B, int heads, int past_len,
                                    const int32_t* past_len_dev, int ctx_max, void* stream) {
    if (B <= 0 || heads <= 0 || past_len < 0 || past_len + 1 > ctx_max || B > 65535 || heads > 65535 ||
        ((uintptr_t)qkv & 15) || ((uintptr_t)kcache & 15) || ((uintptr_t)vcache & 15) || ((uintptr_t)out & 7) || !cos_table || !sin_table) {
        vly_set_error("vly_decode_attention: bad args B=%d heads=%d past=%d ctx_max=%d", B, heads, past_len, ctx_max);
        return -22;
    }
    if (key_valid && key_valid_stride < past_len + 1) {
        vly_set_error("vly_decode_attention: key_valid_stride %d < kv_len %d", key_valid_stride, past_len + 1);
        return -22;
    }
    hipLaunchKernelGGL(decode_fused_kernel, dim3(heads, B), dim3(512), 0, (hipStream_t)stream, (const uint16_t*)qkv,
                       (uint16_t*)kcache, (uint16_t*)vcache, cos_table, sin_table, key_valid, (uint16_t*)out, heads, past_len,
                       past_len_dev, key_valid_stride, ctx_max, 0);
    return vly_check_launch("vly_decode_attention");
}

extern "C" int vly_decode_attention_rows(const void* qkv, void* kcache, void* vcache, const float* cos_table, const float* sin_table,
                                         const uint8_t* key_valid, int key_valid_stride, void* out, int B, int heads,
                                         const int32_t* past_len_rows, int ctx_max, void* stream) {
    if (B <= 0 || heads <= 0 || !past_len_rows || B > 65535 || heads > 65535 || ((uintptr_t)qkv & 15) || ((uintptr_t)kcache & 15) ||
        ((uintptr_t)vcache & 15) || ((uintptr_t)out & 7) || !cos_table || !sin_table || (key_valid && key_valid_stride < ctx_max)) {
        vly_set_error("vly_decode_attention_rows: bad args B=%d heads=%d ctx_max=%d key_valid_stride=%d", B, heads, ctx_max, key_valid_stride);
        return -22;
    }
    hipLaunchKernelGGL(decode_fused_kernel, dim3(heads, B), dim3(512), 0, (hipStream_t)stream, (const uint16_t*)qkv,
                       (uint16_t*)kcache, (uint16_t*)vcache, cos_table, sin_table, key_valid, (uint16_t*)out, heads, 0,
                       past_len_rows, key_valid_stride, ctx_max, 1);
    return vly_check_launch("vly_decode_attention_rows");
}

static int decode_split_launch(const char* name, const void* qkv, void* kcache, void* vcache, const float* cos_table,
                               const float* sin_table, const uint8_t* key_valid, int key_valid_stride, float* partials, int B,
                               int heads, int past_len, const int32_t* past_len_dev, int past_len_dev_stride, int ctx_max,
                               void* merged, unsigned* arrivals, void* stream);

#if VLY_EXPERIMENTAL      // round 3's form (partials merged in the o GEMV's prologue): the merged launch's bit-identity witness
extern "C" int vly_decode_attention_split(const void* qkv, void* kcache, void* vcache, const float* cos_table, const float* sin_table,
                                          const uint8_t* key_valid, int key_valid_stride, float* partials, int B, int heads,
                                          int past_len, const int32_t* past_len_dev, int past_len_dev_stride, int ctx_max,
                                          void* stream) {
    return decode_split_launch("vly_decode_attention_split", qkv, kcache, vcache, cos_table, sin_table, key_valid, key_valid_stride,
                               partials, B, heads, past_len, past_len_dev, past_len_dev_stride, ctx_max, nullptr, nullptr, stream);
}
#endif

extern "C" int vly_decode_attention_merged(const void* qkv, void* kcache, void* vcache, const float* cos_table, const float* sin_table,
                                           const uint8_t* key_valid, int key_valid_stride, float* partials, void* out,
                                           uint32_t* arrivals, int B, int heads, int past_len, const int32_t* past_len_dev,
                                           int past_len_dev_stride, int ctx_max, void* stream) {
    if (!out || ((uintptr_t)out & 7) || !arrivals || ((uintptr_t)arrivals & 3)) {
        vly_set_error("vly_decode_attention_merged: out (8-byte aligned) and arrivals (B * heads zeroed uint32) are required");
        return -22;
    }
    return decode_split_launch("vly_decode_attention_merged", qkv, kcache, vcache, cos_table, sin_table, key_valid, key_valid_stride,
                               partials, B, heads, past_len, past_len_dev, past_len_dev_stride, ctx_max, out, arrivals, stream);
}

static int decode_split_launch(const char* name, const void* qkv, void* kcache, void* vcache, const float* cos_table,
                               const float* sin_table, const uint8_t* key_valid, int key_valid_stride, float* partials, int B,
                               int heads, int past_len, const int32_t* past_len_dev, int past_len_dev_stride, int ctx_max,
                               void* merged, unsigned* arrivals, void* stream) {
    if (B <= 0 || heads <= 0 || past_len < 0 || past_len + 1 > ctx_max || B > 65535 || heads > 65535 || ((uintptr_t)qkv & 15) ||
        ((uintptr_t)kcache & 15) || ((uintptr_t)vcache & 15) || !partials || ((uintptr_t)partials & 15) || !cos_table || !sin_table ||
        past_len_dev_stride < 0 || past_len_dev_stride > 1 || (past_len_dev_stride == 1 && !past_len_dev)) {
        vly_set_error("%s: bad args B=%d heads=%d past=%d ctx_max=%d", name, B, heads, past_len, ctx_max);
        return -22;
    }
    if (key_valid && key_valid_stride < (past_len_dev ? ctx_max : past_len + 1)) {
        vly_set_error("%s: key_valid_stride %d too short", name, key_valid_stride);
        return -22;
    }
    static const bool spread = getenv("VLY_DECODE_SPLIT_SPREAD") && atoi(getenv("VLY_DECODE_SPLIT_SPREAD")) != 0 && VLY_DECODE_SPLITS != 1;
    // (the stress order needs B != SPLITS to be told apart by the kernel: B = 4 keeps the production order)
    const dim3 grid = spread && B != VLY_DECODE_SPLITS ? dim3(VLY_DECODE_SPLITS, heads, B) : dim3(heads, B, VLY_DECODE_SPLITS);
    hipLaunchKernelGGL(decode_split_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)qkv,
                       (uint16_t*)kcache, (uint16_t*)vcache, cos_table, sin_table, key_valid, partials, heads, past_len, past_len_dev,
                       key_valid_stride, ctx_max, past_len_dev_stride, (uint16_t*)merged, arrivals);
    return vly_check_launch(name);
}

extern "C" int vly_llama_attention_probs(const void* qkv, const void* kcache, const uint8_t* key_valid, int key_valid_stride,
                                         float* out, int B, int S, int heads, int past_len, int ctx_max, int inputs_f32,
                                         void* stream) {
    const int kv_len = past_len + S;
    if (B <= 0 || S <= 0 || heads <= 0 || past_len < 0 || kv_len > ctx_max || kv_len > 16000 || B > 65535 || heads > 65535 || !qkv ||
        !kcache || !out) {
        vly_set_error("vly_llama_attention_probs: bad args B=%d S=%d heads=%d past=%d ctx_max=%d (kv_len <= 16000)", B, S, heads,
                      past_len, ctx_max);
        return -22;
    }
    if (key_valid && key_valid_stride < kv_len) {
        vly_set_error("vly_llama_attention_probs: key_valid_stride %d < kv_len %d", key_valid_stride, kv_len);
        return -22;
    }
    const dim3 grid(S, heads, B), block(256);
    const size_t lds = (size_t)kv_len * 4;
    if (inputs_f32)
        hipLaunchKernelGGL(attn_probs_kernel<float>, grid, block, lds, (hipStream_t)stream, (const float*)qkv, (const float*)kcache,
                           key_valid, out, S, heads, past_len, key_valid_stride, ctx_max);
    else
        hipLaunchKernelGGL(attn_probs_kernel<uint16_t>, grid, block, lds, (hipStream_t)stream, (const uint16_t*)qkv,
                           (const uint16_t*)kcache, key_valid, out, S, heads, past_len, key_valid_stride, ctx_max);
    return vly_check_launch("vly_llama_attention_probs");
}
