// One persistent launch for ALL decoder layers of a batch-1/2 decode step (BASELINE.json configs[4];
// serve/model_worker.py:380-394 -> hf LlamaDecoderLayer.forward x L).
//
// Round 3's step was five launches per layer (norm+q|k|v GEMV, split attention, merge+o GEMV, norm+gate|up GEMV, down GEMV):
// 118 us per 13B layer of which the weight stream itself is 92 us at the 6.9 TB/s the longest GEMV reaches — the rest is
// five kernel boundaries (MI355X_MICROARCH "boundary": 1.2-1.9 us each) and, at every one of them, a drained memory pipe that
// has to fill again.  Here ONE 8-wave workgroup per CU (256 registers per lane) walks the five phases of every layer,
// separated by a grid barrier: a workgroup is two 256-thread groups, each of which owns a strided sequence of weight UNITS (a
// row pair, or one row of the K = I down projection) and keeps NS = 3 units in flight in registers (7 x 16 bytes per thread
// each: 172 KB per CU); the first units of the NEXT phase are requested in the last round of the current one (they depend on
// nothing a barrier orders).
//
// MEASURED (13B, 256 tokens, profiles/r04/r04_decode_persistent_*): 198.7 tokens/s against the launches' 208-212 — NOT the
// default (VALLEY_DECODE_PERSISTENT=1 selects it).  Its weight loops do stream at 6.9-7 TB/s (82 of 127 us per layer), but a
// phase boundary inside the launch costs no less than a kernel boundary: workgroups arrive 3.5-8 us apart (static work split),
// the barrier itself takes 1.5-2 us, the activation hand-off + norm 1.4-2.8 us, the store drain 1-2 us; and the requests
// issued ahead of the barrier land before it ends, so they shorten the next loop instead of covering the gap
// (tools/decode_phase_times.py prints the anatomy).  What it would take: dynamic unit claims against the arrival skew and an
// LDS-DMA ring that keeps requesting across the barrier (MI355X_MICROARCH "engine-vs-launches").
//
// Arithmetic: per output element exactly that of the launches it replaces — the same chunk-to-thread mapping, the same wave
// sums, the same fixed-order sum over four waves, norm_row_kernel's norm, decode_split_kernel's attention,
// gemv_norm_kernel's merge — so the step is BIT-IDENTICAL to the five-launch step (tests/test_decode_persistent_gpu.py).
//
// Hand-offs between workgroups (cdna_hip_programming §6 Guideline 16, form R1): every value another CU reads inside the
// launch (q|k|v, attention partials, the residual stream, the MLP intermediate: 10-84 KB per phase) is stored WRITE-THROUGH
// (sc1, agent-scope relaxed atomic stores of 4-8 bytes), the storing waves drain vmcnt before the workgroup arrives at the
// barrier, and readers use sc1 loads after it: no release / acquire fences (a fence per phase would cost 1.7 us x 200).
// The barrier is the XCD-hierarchical counter barrier of MI355X_MICROARCH "barrier-xcd" over LOGICAL groups (block % 8:
// placement only affects speed): per-group counter -> top counter -> per-group generation word, relaxed sc1 polls by one
// lane with s_sleep, every spin BOUNDED: a workgroup that gives up sets the abort word, everyone leaves at their next
// barrier, and the host reads the word.  The sync words are zeroed ONCE by the caller and count on from launch to launch (every
// launch passes the same number of barriers; see the kernel), and again by the caller after an abort.
// EXPERIMENTAL: compiled into libvalley_hip_exp.so only (-DVLY_EXPERIMENTAL=1, valley_amd/build.py) — measured behind the five launches
// it replaces, so the shipped libraries and include/valley_hip.h's default section do not carry it.
#ifndef VLY_EXPERIMENTAL
#define VLY_EXPERIMENTAL 0
#endif
#if VLY_EXPERIMENTAL
#include "common.hpp"
#include "../../include/valley_hip.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;
constexpr int SPLITS = VLY_DECODE_SPLITS;
constexpr int WG_THREADS = 512;          // two 256-thread groups, 256 registers per lane
constexpr int NGRP = WG_THREADS / 256;
#ifndef VLY_DL_SLOTS
#define VLY_DL_SLOTS 3
#endif
constexpr int NS = VLY_DL_SLOTS;           // weight units in flight per group (7 x 16 bytes per thread each)
constexpr int ATTN_SCRATCH_FLOATS = 256 + 3 * 128 + 8 + 16 * 128;      // sc, qs, knew, vnew, red, acc_s (per 256-thread group)
constexpr unsigned SPIN_LIMIT = 1u << 20;                                  // ~1 s of polling before a workgroup gives up

// sync words (uint32), each polled word on its own 64-byte line
constexpr int SY_CNT = 0;            // [8] x 16
constexpr int SY_TOP = 8 * 16;
constexpr int SY_GEN = 9 * 16;       // [8] x 16
constexpr int SY_ABORT = 17 * 16;
static_assert(SY_ABORT + 16 <= VLY_DECODE_SYNC_WORDS, "sync area");

struct Args {
    const vly_decode_layer* layers;
    int n_layers;
    float* h;                   // [B, H] fp32 residual stream (in / out)
    uint32_t* qkv;              // [B, 3H] 16-bit, as pairs
    float* partials;            // [B, heads, SPLITS, 132]
    float* mlp;                 // [B, I] fp32 (silu(gate) * up before its 16-bit rounding)
    const float* cos_t;
    const float* sin_t;
    const uint8_t* key_valid;
    int kv_stride;
    const int32_t* pos_dev;
    int pos_stride;
    int B, H, heads, I;
    float eps;
    int ctx_max;
    uint32_t* sync;
    unsigned long long* timing; // debugging aid (vlydbg_decode_timing): [workgroup][layer][5 phases][5 stamps] of s_memrealtime, or nullptr
    int stop;                   // debugging aid (VLY_DL_STOP): leave after grid barrier `stop` of the first layer (0: never)
};

// ---- cross-CU accesses: write-through stores, L1-bypassing loads ------------------------------------------------------
#ifndef VLY_DL_SCOPE
#define VLY_DL_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
VLY_DEVICE void st_cc(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, VLY_DL_SCOPE); }
VLY_DEVICE void st_cc(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, VLY_DL_SCOPE); }
VLY_DEVICE void st_cc2(float* p, float a, float b) {            // 8 bytes, 8-byte aligned
    const unsigned long long v = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
    __hip_atomic_store((unsigned long long*)p, v, __ATOMIC_RELAXED, VLY_DL_SCOPE);
}
VLY_DEVICE uint32_t ld_cc(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, VLY_DL_SCOPE); }
VLY_DEVICE float ld_cc(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, VLY_DL_SCOPE); }
VLY_DEVICE u32x2 ld_cc2(const void* p) {
    const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, VLY_DL_SCOPE);
    return u32x2{(uint32_t)v, (uint32_t)(v >> 32)};
}
VLY_DEVICE float4 ld_cc4(const float* p) {                       // 16 bytes as two 8-byte agent-scope loads
    const u32x2 a = ld_cc2(p), b = ld_cc2(p + 2);
    return make_float4(__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(b[0]), __uint_as_float(b[1]));
}

// Pointers that come out of the layer table are generic to the compiler (it only knows kernel ARGUMENTS to be global):
// loads through them would be flat_load with a 64-bit per-lane address, counted on lgkmcnt AND vmcnt.  Everything read
// through a table pointer goes through these.
#define VLY_GLOBAL __attribute__((address_space(1)))
template <typename T>
VLY_DEVICE const VLY_GLOBAL T* as_global(const void* p) { return (const VLY_GLOBAL T*)(uintptr_t)p; }
template <typename T>
VLY_DEVICE VLY_GLOBAL T* as_global_rw(void* p) { return (VLY_GLOBAL T*)(uintptr_t)p; }

// The layer table is read with vector loads (the kernel stores to memory, so hipcc does not scalarise them) and its pointers
// arrive in VGPRs: a buffer descriptor built from one gets a WATERFALL loop around every load that uses it.  They are
// workgroup-uniform by construction: say so.
template <typename T>
VLY_DEVICE T* uniform_ptr(T* p) {
    const uintptr_t v = (uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (T*)((uintptr_t)lo | ((uintptr_t)hi << 32));
}
VLY_DEVICE vly_decode_layer load_layer(const vly_decode_layer* t, int l) {
    const vly_decode_layer r = t[l];
    return vly_decode_layer{uniform_ptr(r.w_qkv), uniform_ptr(r.w_o), uniform_ptr(r.w_gu), uniform_ptr(r.w_down),
                            uniform_ptr(r.ln1), uniform_ptr(r.ln2), uniform_ptr(r.kcache), uniform_ptr(r.vcache)};
}

// Lane-constant addresses are loop-invariant across the LAYER loop: left alone, hipcc hoists every set-up function's address
// arithmetic (a dozen 64-bit lane values each) to kernel entry and carries ~120 registers through the unit loops, next to the
// slots.  Every phase re-derives its lane index from a value the optimiser cannot see through.
VLY_DEVICE int opaque_i32(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

VLY_DEVICE float dot8(const u32x4& w, const u32x4& a) { return vly_dot8(w, a); }      // common.hpp: one definition for both files

// ---- the weight stream ------------------------------------------------------------------------------------------------
// A UNIT is NR weight rows (NR = 2: a row pair of q|k|v / o / gate|up — gate and up of one SwiGLU output; NR = 1: one row
// of the down projection, whose K = I is 2.7 x longer); a thread of the owning group holds chunks tid + 256 i of each row:
// NR x CH 16-byte loads per unit and thread (6 or 7 at the 13B shapes), plus the residual words of its outputs where the
// phase adds to the residual stream.  Group g owns units g, g + G, g + 2 G, ... and keeps NS of them in flight.
enum { K_QKV = 0, K_RES2 = 1, K_SWIGLU = 2, K_RES1 = 3, K_NONE = -1 };

template <int MR>
struct Slot {
    u32x4 r[7];
    u32x2 res[MR];
};

struct WPhase {
    const uint16_t* W;
    int units;              // live units (rows / NR); a unit index beyond them fetches nothing
};

// what a lane contributes to every request of the kernel: its 16-byte column inside a 4096-byte row segment, and the same
// with the ragged last chunk of an H-wide / I-wide row masked out
struct Lane {
    uint32_t vfull, vlastH, vlastI;
};

// The requests go through a BUFFER descriptor of the weight matrix (base, bytes): the row offset is a scalar (soffset), the
// lane offset a register shared by the whole phase (16 tid; chunk i adds 4096 i on the scalar side), and a lane that has
// nothing to fetch — the ragged last chunk, a unit past the end — presents an offset beyond the descriptor's range: the
// hardware returns zeros without touching memory.  No 64-bit lane arithmetic and no branch: hipcc's vmcnt bookkeeping stays
// exact (a branch around a load makes it wait vmcnt(0): gemv_bf16.hip load_pair), ~20 instructions per unit.
constexpr uint32_t OOB = 0x80000000u;                        // beyond any matrix (all are < 2 GB)
template <int MR, int CHH, int CHI, int KIND>
VLY_DEVICE void issue(Slot<MR>& s, const WPhase& p, int unit, const Lane& ln, const Args& a) {
    constexpr bool down = KIND == K_RES1;
    constexpr int NR = down ? 1 : 2, CH = down ? CHI : CHH;
    const uint32_t row_bytes = 2u * (uint32_t)(down ? a.I : a.H);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.W), 0, (uint32_t)p.units * NR * row_bytes, 0x00020000);
    const bool live = unit < p.units;                        // workgroup-uniform
    const uint32_t base = live ? (uint32_t)unit * NR * row_bytes : 0u;
    const uint32_t vf = live ? ln.vfull : OOB, vl = live ? (down ? ln.vlastI : ln.vlastH) : OOB;
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int i = 0; i < CH; ++i)
            s.r[r * CH + i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, i == CH - 1 ? vl : vf,
                                                                                               base + r * row_bytes + 4096u * i, 2 /* nt */));
    if constexpr (KIND == K_RES2 || KIND == K_RES1) {
        // the residual words this unit's outputs are added to: final since the last barrier (h is only written by the phase
        // that owns the row), one address for the whole group (one request per wave)
        const int n0 = unit * NR;
#pragma unroll
        for (int m = 0; m < MR; ++m) s.res[m] = ld_cc2(a.h + (live ? (size_t)m * a.H + (n0 & ~1) : (size_t)0));
    }
}

// dot products of one unit against the activation rows in LDS, wave sums, the four waves' partials through LDS (fixed order),
// epilogue and write-through store by thread 0 of the group.  ONE workgroup barrier per unit (rd alternates between units).
template <int MR, int CHH, int CHI, int KIND>
VLY_DEVICE void consume(Slot<MR>& s, const uint16_t* xs, int units, int unit, int tid, float* rd, const Args& a) {
    constexpr bool down = KIND == K_RES1;
    constexpr int NR = down ? 1 : 2, CH = down ? CHI : CHH;
    const int K = down ? a.I : a.H, nch = K >> 3;
    const int lane = tid & 63, wave = tid >> 6;
    const int tq = opaque_i32(tid);                          // (the clamped LDS offsets below are recomputed per unit, not carried)
    // The dot products only depend on registers: nothing orders them behind the previous unit's barrier, and hipcc computes the
    // sums of ALL slots at the top of a round, behind one wait for every slot in flight — the stream would drain once per
    // round.  An empty asm on the slot's registers pins this unit's arithmetic (and the wait for its loads) here.  (The
    // residual words too: a register the phase never reads is dead on arrival, the allocator hands it to temporaries, and
    // every write to one waits for the load still in flight into it.)
#pragma unroll
    for (int i = 0; i < NR * CH; ++i) {
        u32x4 q = s.r[i];
        asm volatile("" : "+v"(q));
        s.r[i] = q;
    }
    if constexpr (KIND == K_RES2 || KIND == K_RES1) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            u32x2 q = s.res[m];
            asm volatile("" : "+v"(q));
            s.res[m] = q;
        }
    }
    float acc[NR][MR];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[r][m] = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {                           // chunk order per thread: tid, tid + 256, ... (gemv_kernel<.., 4, .>)
        // a chunk past the row holds ZEROS (issue: out-of-range lanes fetch nothing) and adds + 0 to the sums — the launches skip
        // it, same bits; its activation chunk is clamped to one that exists (0 x finite, never 0 x stale LDS)
        const int c = min(tq + 256 * i, nch - 1);
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const u32x4 av = *(const u32x4*)(xs + (size_t)m * K + 8 * c);
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[r][m] += dot8(s.r[r * CH + i], av);
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[r][m] = wave_sum(acc[r][m]);
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int r = 0; r < NR; ++r) rd[(wave * MR + m) * NR + r] = acc[r][m];
    }
    __syncthreads();
    if (tid != 0 || unit >= units) return;
    const int n0 = unit * NR;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        if (m >= a.B) break;
        float v[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float t = rd[m * NR + r];                        // fixed order: wave 0 + 1 + 2 + 3
#pragma unroll
            for (int wv = 1; wv < 4; ++wv) t += rd[(wv * MR + m) * NR + r];
            v[r] = t + 0.f;                                  // (+ bias, absent on this path: the launches' own expression)
        }
        if constexpr (KIND == K_QKV) {
            st_cc(a.qkv + ((size_t)m * 3 * a.H + n0) / 2, pack_h2(v[0], v[1]));
        } else if constexpr (KIND == K_SWIGLU) {
            st_cc(a.mlp + (size_t)m * a.I + (n0 >> 1), x_sigmoid(v[0], 1.f) * v[1]);
        } else if constexpr (KIND == K_RES2) {
            st_cc2(a.h + (size_t)m * a.H + n0, v[0] + __uint_as_float(s.res[m][0]), v[1] + __uint_as_float(s.res[m][1]));
        } else {
            st_cc(a.h + (size_t)m * a.H + n0, v[0] + __uint_as_float((n0 & 1) ? s.res[m][1] : s.res[m][0]));
        }
    }
}

// The units of one phase: trips = ceil(units / G) passes for every group (a group whose last unit does not exist makes the
// pass with zeros and stores nothing: the workgroup's barriers stay matched).  Full rounds re-request a slot for the same
// phase as soon as it is consumed; in the LAST round (1 .. NS units) a consumed slot is handed to the first units of the
// NEXT phase, which depend on nothing a barrier orders: they are in flight under the barrier and the next set-up.
template <int MR, int CHH, int CHI, int KIND, int NEXT>
VLY_DEVICE void run_phase(Slot<MR> (&S)[NS], const WPhase& cur, const WPhase& nxt, const uint16_t* xs, int gid, int G,
                          int tid, const Lane& ln, float* rd0, float* rd1, const Args& a) {
    const int trips = (cur.units + G - 1) / G;               // >= 1
    const int full = (trips - 1) / NS;                       // rounds before the last one
#pragma unroll 1
    for (int rnd = 0; rnd < full; ++rnd) {
        const int t = rnd * NS;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            consume<MR, CHH, CHI, KIND>(S[k], xs, cur.units, gid + G * (t + k), tid, ((t + k) & 1) ? rd1 : rd0, a);
            issue<MR, CHH, CHI, KIND>(S[k], cur, gid + G * (t + k + NS), ln, a);      // (beyond the last pass: fetches nothing)
        }
    }
    const int t = full * NS, rem = trips - t;                // 1 .. NS passes left
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        if (k < rem) consume<MR, CHH, CHI, KIND>(S[k], xs, cur.units, gid + G * (t + k), tid, ((t + k) & 1) ? rd1 : rd0, a);
        if constexpr (NEXT != K_NONE) issue<MR, CHH, CHI, NEXT>(S[k], nxt, gid + G * k, ln, a);
    }
}

// ---- activation set-up of a phase (after the barrier): the rows every unit is multiplied with, 16-bit, into LDS ----------
// RMSNorm of the fp32 residual stream: norm_row_kernel's arithmetic operation for operation, by group 0; the other groups
// meet it at the same 2 MR barriers (gemv_norm_kernel's prologue).  gamma comes from LDS (`gs`, staged a phase or more ahead
// by stage_gamma: two slots of weights are in flight in registers here, 24 more for gamma do not fit beside the row)
template <int MR, int CH>
VLY_DEVICE void setup_norm(const float* H, const float* gs, float eps, uint16_t* xs, float* nred, int K, int M, int grp, int tid_) {
    const int tid = opaque_i32(tid_);
    const int nvec = K >> 2, lane = tid & 63, wave = tid >> 6;
    if (grp == 0) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            float4 v[2 * CH];
            const float* hr = H + (size_t)min(m, M - 1) * K;
#pragma unroll
            for (int i = 0; i < 2 * CH; ++i) {
                const int c = tid + 256 * i;
                const float4 t = ld_cc4(hr + 4 * min(c, nvec - 1));
                v[i] = (c < nvec) ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 2 * CH; ++i) s += vly_sumsq4(v[i].x, v[i].y, v[i].z, v[i].w);
            s = wave_sum(s);
            if (lane == 0) nred[wave] = s;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            s = nred[0] + nred[1] + nred[2] + nred[3];
            const float rstd = rsqrtf(s / (float)K + eps);
#pragma unroll
            for (int i = 0; i < 2 * CH; ++i) {
                const int c = tid + 256 * i;
                if (c >= nvec) continue;
                const float4 gm = *(const float4*)(gs + 4 * c);
                float4 o;
                o.x = gm.x * (v[i].x * rstd); o.y = gm.y * (v[i].y * rstd);
                o.z = gm.z * (v[i].z * rstd); o.w = gm.w * (v[i].w * rstd);
                u32x2 pk;
                pk[0] = pack_h2(o.x, o.y);
                pk[1] = pack_h2(o.z, o.w);
                *(u32x2*)(xs + (size_t)m * K + 4 * c) = pk;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    } else {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            asm volatile("s_barrier" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
    }
}

// An RMSNorm weight vector on its way into LDS: requested (two float4 per thread cover K <= 8192) before a phase's unit loop,
// written after it — the loads complete under the loop, nothing waits for them
struct GammaRegs {
    f32x4 g[4];                                              // 512 threads x 4 float4 cover K <= 8192
};
VLY_DEVICE void stage_gamma_issue(GammaRegs& r, const float* gamma, int K) {
    const int nvec = K >> 2, t = opaque_i32(threadIdx.x);
#pragma unroll
    for (int i = 0; i < 4; ++i) r.g[i] = as_global<f32x4>(gamma)[min(t + WG_THREADS * i, nvec - 1)];
}
VLY_DEVICE void stage_gamma_commit(const GammaRegs& r, float* gs, int K) {
    const int nvec = K >> 2, t = opaque_i32(threadIdx.x);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = t + WG_THREADS * i;
        if (c < nvec) *(f32x4*)(gs + 4 * c) = r.g[i];
    }
}

// merge of the attention partials (gemv_norm_kernel PRO = 1): all threads, unit u = dims 4 (u & 31) .. + 3 of head u >> 5
template <int MR>
VLY_DEVICE void setup_merge(const float* partials, uint16_t* xs, int heads, int M) {
    const int K = heads * 128, nvec = K >> 2, t0 = opaque_i32(threadIdx.x);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const float* pb_ = partials + (size_t)min(m, M - 1) * heads * (SPLITS * 132);
#pragma unroll 2
        for (int u = t0; u < nvec; u += WG_THREADS) {
            const float* hb = pb_ + (size_t)(u >> 5) * (SPLITS * 132);
            float ms[SPLITS], ls[SPLITS];
            float4 os[SPLITS];
#pragma unroll
            for (int sp = 0; sp < SPLITS; ++sp) {
                const u32x2 ml = ld_cc2(hb + sp * 132);
                ms[sp] = __uint_as_float(ml[0]);
                ls[sp] = __uint_as_float(ml[1]);
                os[sp] = ld_cc4(hb + sp * 132 + 4 + 4 * (u & 31));
            }
            float mx = ms[0];
#pragma unroll
            for (int sp = 1; sp < SPLITS; ++sp) mx = fmaxf(mx, ms[sp]);
            float L = 0.f;
            float4 O = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int sp = 0; sp < SPLITS; ++sp) {            // split order: deterministic
                const float w = exp2f(ms[sp] - mx);
                L = fmaf(ls[sp], w, L);
                O.x = fmaf(os[sp].x, w, O.x); O.y = fmaf(os[sp].y, w, O.y);
                O.z = fmaf(os[sp].z, w, O.z); O.w = fmaf(os[sp].w, w, O.w);
            }
            u32x2 pk;
            pk[0] = pack_h2(O.x / L, O.y / L);
            pk[1] = pack_h2(O.z / L, O.w / L);
            *(u32x2*)(xs + (size_t)m * K + 4 * u) = pk;
        }
    }
    __syncthreads();
}

// the MLP intermediate: fp32 silu(gate) * up -> its 16-bit rounding (what vly_gemv's SwiGLU epilogue stores)
template <int MR>
VLY_DEVICE void setup_mlp(const float* mlp, uint16_t* xs, int I, int M) {
    const int nvec = I >> 2, t0 = opaque_i32(threadIdx.x);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const float* src = mlp + (size_t)min(m, M - 1) * I;
#pragma unroll 4
        for (int c = t0; c < nvec; c += WG_THREADS) {
            const float4 v = ld_cc4(src + 4 * c);
            u32x2 pk;
            pk[0] = pack_h2(v.x, v.y);
            pk[1] = pack_h2(v.z, v.w);
            *(u32x2*)(xs + (size_t)m * I + 4 * c) = pk;
        }
    }
    __syncthreads();
}

// ---- attention of the new token, one (batch, head, key split) per 256-thread group: decode_split_kernel's arithmetic ---------
// Register-lean form for a 1024-thread workgroup (128 registers per lane): a thread's 256-byte K row and its sixteen V chunks
// arrive in two halves each (same accumulation order).  `trips` is the workgroup's (uniform) count of 256-key passes:
// 3 trips + 2 workgroup barriers, matched one for one by the groups that hold no item (attn_idle).
VLY_DEVICE void attn_item(const Args& a, const vly_decode_layer& Lp, int b, int h, int sp, int tid_, float* scratch, int trips) {
    const int tid = opaque_i32(tid_);
    float* sc = scratch;
    float* qs = sc + 256;
    float* knew = qs + 128;
    float* vnew = knew + 128;
    float* red = vnew + 128;
    float* acc_s = red + 8;                                  // [16][128]
    const int lane = tid & 63, wave = tid >> 6;
    const int heads = a.heads, Hq = heads * 128, ctx_max = a.ctx_max;
    const int pos = min(a.pos_dev[(size_t)b * a.pos_stride], ctx_max - 1), kv_len = pos + 1;
    const int chunk = (((kv_len + SPLITS - 1) / SPLITS) + 63) & ~63;
    const int lo = sp * chunk, hi = min(lo + chunk, kv_len);
    const bool owner = pos >= lo && pos < hi;
    const uint32_t* qp32 = a.qkv + ((size_t)b * 3 * Hq + h * 128) / 2;
    VLY_GLOBAL uint16_t* kbase = as_global_rw<uint16_t>(Lp.kcache) + ((size_t)b * heads + h) * ctx_max * 128;
    VLY_GLOBAL uint16_t* vbase = as_global_rw<uint16_t>(Lp.vcache) + ((size_t)b * heads + h) * ctx_max * 128;
    const uint8_t* kvld = a.key_valid ? a.key_valid + (size_t)b * a.kv_stride : nullptr;
    float* part = a.partials + (((size_t)b * heads + h) * SPLITS + sp) * 132;
    auto half_of = [](uint32_t w, int i) -> uint16_t { return (uint16_t)((i & 1) ? (w >> 16) : (w & 0xffffu)); };
    if (tid < 64) {
        const float cs = a.cos_t[(size_t)pos * 64 + tid], sn = a.sin_t[(size_t)pos * 64 + tid];
        const float q0 = h2f(half_of(ld_cc(qp32 + (tid >> 1)), tid)), q1 = h2f(half_of(ld_cc(qp32 + ((tid + 64) >> 1)), tid));
        const float scale = 0.08838834764831845f * LOG2E;
        qs[tid] = h2f(f2h(rope_rot(q0, q1, cs, sn, -1.f))) * scale;
        qs[tid + 64] = h2f(f2h(rope_rot(q1, q0, cs, sn, 1.f))) * scale;
        if (owner) {
            const float k0 = h2f(half_of(ld_cc(qp32 + ((Hq + tid) >> 1)), tid)), k1 = h2f(half_of(ld_cc(qp32 + ((Hq + tid + 64) >> 1)), tid));
            const uint16_t r0 = f2h(rope_rot(k0, k1, cs, sn, -1.f)), r1 = f2h(rope_rot(k1, k0, cs, sn, 1.f));
            knew[tid] = h2f(r0);
            knew[tid + 64] = h2f(r1);
            kbase[(size_t)pos * 128 + tid] = r0;
            kbase[(size_t)pos * 128 + tid + 64] = r1;
        }
    } else if (tid < 192 && owner) {
        const int d = tid - 64;
        const uint16_t v = half_of(ld_cc(qp32 + ((2 * Hq + d) >> 1)), d);
        vnew[d] = h2f(v);
        vbase[(size_t)pos * 128 + d] = v;
    }
    __syncthreads();

    const int kg = tid >> 4, dc = tid & 15;
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;
#pragma unroll 1
    for (int it = 0; it < trips; ++it) {
        const int c0 = lo + 256 * it;                        // (passes beyond this item's range leave m, l, o as they are)
        u32x4 va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = c0 + kg + 16 * u;
            va[u] = (j < hi && j < pos) ? *(const VLY_GLOBAL u32x4*)(vbase + (size_t)j * 128 + 8 * dc) : u32x4{0u, 0u, 0u, 0u};
        }
        const int jk = c0 + tid;
        float s = NEG_BIG;
        if (jk < hi && (!kvld || kvld[jk])) {
            float ac = 0.f;
            if (jk < pos) {
                const VLY_GLOBAL u32x4* kr = (const VLY_GLOBAL u32x4*)(kbase + (size_t)jk * 128);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    u32x4 kk[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) kk[c] = kr[8 * hf + c];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const f32x4 q0 = *(const f32x4*)(qs + 8 * (8 * hf + c)), q1 = *(const f32x4*)(qs + 8 * (8 * hf + c) + 4);
                        ac = fmaf(h_lo(kk[c][0]), q0[0], ac); ac = fmaf(h_hi(kk[c][0]), q0[1], ac);
                        ac = fmaf(h_lo(kk[c][1]), q0[2], ac); ac = fmaf(h_hi(kk[c][1]), q0[3], ac);
                        ac = fmaf(h_lo(kk[c][2]), q1[0], ac); ac = fmaf(h_hi(kk[c][2]), q1[1], ac);
                        ac = fmaf(h_lo(kk[c][3]), q1[2], ac); ac = fmaf(h_hi(kk[c][3]), q1[3], ac);
                    }
                }
            } else {                                             // the new token's key: still in LDS
#pragma unroll 8
                for (int d = 0; d < 128; ++d) ac = fmaf(knew[d], qs[d], ac);
            }
            s = ac;
        }
        __builtin_amdgcn_sched_barrier(0);                   // (not above the K rows: va + vb + kk would be 96 registers)
#pragma unroll
        for (int u = 0; u < 8; ++u) {                        // second half of the V rows: in flight under the softmax reductions
            const int j = c0 + kg + 16 * (u + 8);
            vb[u] = (j < hi && j < pos) ? *(const VLY_GLOBAL u32x4*)(vbase + (size_t)j * 128 + 8 * dc) : u32x4{0u, 0u, 0u, 0u};
        }
        float mc = wave_max(s);
        if (lane == 0) red[wave] = mc;
        __syncthreads();
        mc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float m_new = fmaxf(m_run, mc);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        const float p = s > 0.5f * NEG_BIG ? __builtin_amdgcn_exp2f(s - m_new) : 0.f;
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
            const u32x4 vv = u < 8 ? va[u & 7] : vb[u & 7];
            if (owner && c0 + jl == pos) {
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = fmaf(pj, vnew[8 * dc + i], o[i]);
            } else {
                o[0] = fmaf(pj, h_lo(vv[0]), o[0]); o[1] = fmaf(pj, h_hi(vv[0]), o[1]);
                o[2] = fmaf(pj, h_lo(vv[1]), o[2]); o[3] = fmaf(pj, h_hi(vv[1]), o[3]);
                o[4] = fmaf(pj, h_lo(vv[2]), o[4]); o[5] = fmaf(pj, h_hi(vv[2]), o[5]);
                o[6] = fmaf(pj, h_lo(vv[3]), o[6]); o[7] = fmaf(pj, h_hi(vv[3]), o[7]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc_s[kg * 128 + 8 * dc + i] = o[i];
    __syncthreads();
    if (tid < 128) {
        float t = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) t += acc_s[k2 * 128 + tid];
        st_cc(part + 4 + tid, t);
    } else if (tid < 132) {
        st_cc(part + (tid - 128), tid == 128 ? m_run : tid == 129 ? l_run : 0.f);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the partials have left before this workgroup arrives at the barrier
}

VLY_DEVICE void attn_idle(int trips) {
    const int nb = 3 * trips + 2;
#pragma unroll 1
    for (int i = 0; i < nb; ++i) asm volatile("s_barrier" ::: "memory");
}

VLY_DEVICE void stamp(const Args& a, int layer, int phase, int k) {
    if (a.timing != nullptr && threadIdx.x == 0)
        a.timing[(((size_t)blockIdx.x * a.n_layers + layer) * 5 + phase) * 5 + k] = __builtin_amdgcn_s_memrealtime();
}

// ---- grid barrier (MI355X_MICROARCH "barrier-xcd" over logical groups block % 8; bounded) -------------------------------
VLY_DEVICE bool grid_barrier(const Args& a, int layer, int phase, unsigned epoch, int* flag, bool storing_wave) {
    uint32_t* sync = a.sync;
    stamp(a, layer, phase, 2);
    if (storing_wave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have completed
    __syncthreads();
    stamp(a, layer, phase, 3);
    if (threadIdx.x == 0) {
        const unsigned nwg = gridDim.x, g = blockIdx.x & 7u, ngr = nwg < 8u ? nwg : 8u, members = (nwg - g + 7u) / 8u;
        int ok = 1;
#if VLY_DL_RELEASE
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        if (ld_cc(sync + SY_ABORT) != 0u) ok = 0;
        else {
            const unsigned old = __hip_atomic_fetch_add(sync + SY_CNT + 16 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1u == members * epoch) {
                const unsigned o2 = __hip_atomic_fetch_add(sync + SY_TOP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (o2 + 1u == ngr * epoch)
                    for (unsigned gg = 0; gg < ngr; ++gg) st_cc(sync + SY_GEN + 16 * gg, epoch);
            }
            unsigned spins = 0;
            while ((int)(ld_cc(sync + SY_GEN + 16 * g) - epoch) < 0) {
                if (++spins > SPIN_LIMIT || ((spins & 255u) == 0u && ld_cc(sync + SY_ABORT) != 0u)) {
                    st_cc(sync + SY_ABORT, 1u | (epoch << 1));        // (never 0, whatever the epoch)
                    ok = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
#if VLY_DL_ACQUIRE
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        *flag = ok;
    }
    __syncthreads();
    stamp(a, layer, phase, 4);
    return *flag != 0;
}

template <int MR, int CHH, int CHI>
__global__ void __launch_bounds__(WG_THREADS) decode_layers_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char dl_smem[];
    const int H = a.H, I = a.I, B = a.B;
    const int Kmax = I > H ? I : H;
    uint16_t* xs = (uint16_t*)dl_smem;                                   // [MR][Kmax]
    float* fl = (float*)(dl_smem + (((size_t)MR * Kmax * 2 + 15) & ~(size_t)15));
    float* gs1 = fl;                                                     // [H] input_layernorm weight of the layer ahead
    float* gs2 = gs1 + H;                                                // [H] post_attention_layernorm weight
    float* red = gs2 + H;                                                // [2][NGRP][4 * MR * 2]
    float* nred = red + 2 * NGRP * (4 * MR * 2);                         // [4]
    int* flag = (int*)(nred + 4);                                        // [4]
    float* attn = nred + 8;                                              // [NGRP][ATTN_SCRATCH_FLOATS]
    const int tid = threadIdx.x & 255;
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    const bool storing_wave = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    const int nwg = gridDim.x, G = nwg * NGRP, gid = (int)blockIdx.x * NGRP + grp;
    float* rd0 = red + grp * (4 * MR * 2);
    float* rd1 = red + (NGRP + grp) * (4 * MR * 2);
    // The barrier words are never reset: every launch passes the same number of barriers, so the generation word of this
    // workgroup's group still holds the last epoch of the previous launch (complete by stream order) and the counters are
    // whole multiples of it; this launch counts on from there, in wrap-around arithmetic.  (A memset node ahead of the launch,
    // the other way to re-initialise, was observed to race with the kernel under hipGraph replay.)  After an ABORT the words
    // are inconsistent: the host zeroes them (DecodeSession.check).
    unsigned epoch = ld_cc(a.sync + SY_GEN + 16 * ((int)blockIdx.x & 7));

    // workgroup-uniform pass count of the attention phase (longest row; per-row positions allowed)
    int trips = 1;
    for (int b = 0; b < B; ++b) {
        const int kvl = min(a.pos_dev[(size_t)b * a.pos_stride], a.ctx_max - 1) + 1;
        const int chunk = (((kvl + SPLITS - 1) / SPLITS) + 63) & ~63;
        trips = max(trips, (chunk + 255) >> 8);
    }
    const int items = B * a.heads * SPLITS;

    const vly_decode_layer* Lt = a.layers;
    const vly_decode_layer L0 = load_layer(Lt, 0);
    Lane ln;
    {
        const int t = opaque_i32(tid);
        ln.vfull = 16u * (uint32_t)t;
        ln.vlastH = t + 256 * (CHH - 1) < (H >> 3) ? ln.vfull : OOB;
        ln.vlastI = t + 256 * (CHI - 1) < (I >> 3) ? ln.vfull : OOB;
    }
    Slot<MR> S[NS];
    {
        const WPhase pq{(const uint16_t*)L0.w_qkv, 3 * H / 2};
#pragma unroll
        for (int k = 0; k < NS; ++k) issue<MR, CHH, CHI, K_QKV>(S[k], pq, gid + G * k, ln, a);
        GammaRegs g1;
        stage_gamma_issue(g1, L0.ln1, H);                             // (the first norm of the step waits for its weights once)
        stage_gamma_commit(g1, gs1, H);
        __syncthreads();
    }
#pragma unroll 1
    for (int l = 0; l < a.n_layers; ++l) {
        const vly_decode_layer L = load_layer(Lt, l);
        const bool more = l + 1 < a.n_layers;
        const vly_decode_layer* Lnp = Lt + (more ? l + 1 : l);
        const void* nxt_wqkv = uniform_ptr(Lnp->w_qkv);
        const float* nxt_ln1 = uniform_ptr(Lnp->ln1);
        const WPhase pq{(const uint16_t*)L.w_qkv, 3 * H / 2}, po{(const uint16_t*)L.w_o, H / 2}, pg{(const uint16_t*)L.w_gu, I},
            pd{(const uint16_t*)L.w_down, H}, nq{(const uint16_t*)nxt_wqkv, more ? 3 * H / 2 : 0};
        GammaRegs gr;
        // ---- input_layernorm + q|k|v --------------------------------------------------------------------------------
        stamp(a, l, 0, 0);
        setup_norm<MR, CHH>(a.h, gs1, a.eps, xs, nred, H, B, grp, tid);
        stamp(a, l, 0, 1);
        stage_gamma_issue(gr, L.ln2, H);
        run_phase<MR, CHH, CHI, K_QKV, K_RES2>(S, pq, po, xs, gid, G, tid, ln, rd0, rd1, a);
        stage_gamma_commit(gr, gs2, H);
        if (!grid_barrier(a, l, 0, ++epoch, flag, storing_wave)) return;
        // ---- RoPE + KV append + attention of the new token (items on the first groups), o-projection weights on their way -----
        stamp(a, l, 1, 0);
        {
            const int item = grp * nwg + (int)blockIdx.x;
            // (the first units of the o projection were requested at the end of the q|k|v phase, by every group: they landed under
            // the barrier, the attention's three dependent round trips see a quiet memory system, and 80 % of the o projection
            // is on chip when the phase ends.  Requested HERE, by the idle groups, 58 MB of weights queued in front of the
            // attention's loads: 14.8 us for a phase whose stand-alone kernel takes 6.9)
            if (item < items) {
                const int sp = item % SPLITS, hh = (item / SPLITS) % a.heads, bb = item / (SPLITS * a.heads);
                attn_item(a, L, bb, hh, sp, tid, attn + grp * ATTN_SCRATCH_FLOATS, trips);
            } else {
                attn_idle(trips);
            }
        }
        if (!grid_barrier(a, l, 1, ++epoch, flag, false)) return;
        // ---- merge + o projection + residual ---------------------------------------------------------------------------
        stamp(a, l, 2, 0);
        setup_merge<MR>(a.partials, xs, a.heads, B);
        stamp(a, l, 2, 1);
        run_phase<MR, CHH, CHI, K_RES2, K_SWIGLU>(S, po, pg, xs, gid, G, tid, ln, rd0, rd1, a);
        if (!grid_barrier(a, l, 2, ++epoch, flag, storing_wave)) return;
        if (a.stop == 3) return;
        // ---- post_attention_layernorm + gate|up + SwiGLU ----------------------------------------------------------------
        stamp(a, l, 3, 0);
        setup_norm<MR, CHH>(a.h, gs2, a.eps, xs, nred, H, B, grp, tid);
        stamp(a, l, 3, 1);
        stage_gamma_issue(gr, nxt_ln1, H);
        run_phase<MR, CHH, CHI, K_SWIGLU, K_RES1>(S, pg, pd, xs, gid, G, tid, ln, rd0, rd1, a);
        stage_gamma_commit(gr, gs1, H);
        if (!grid_barrier(a, l, 3, ++epoch, flag, storing_wave)) return;
        // ---- down projection + residual; the next layer's q|k|v rows follow it in the stream ----------------------------
        stamp(a, l, 4, 0);
        setup_mlp<MR>(a.mlp, xs, I, B);
        stamp(a, l, 4, 1);
        run_phase<MR, CHH, CHI, K_RES1, K_QKV>(S, pd, nq, xs, gid, G, tid, ln, rd0, rd1, a);
        if (more && !grid_barrier(a, l, 4, ++epoch, flag, storing_wave)) return;
    }
}

int cu_count() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus;
}

template <int MR, int CHH, int CHI>
int launch(const Args& a, size_t lds, hipStream_t st) {
    static const hipError_t attr = hipFuncSetAttribute((const void*)decode_layers_kernel<MR, CHH, CHI>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) {
        vly_set_error("vly_decode_layers: hipFuncSetAttribute: %s", hipGetErrorString(attr));
        return -(1000 + (int)attr);
    }
    // every workgroup must be resident for the grid barrier: one workgroup per CU, never more than the CUs
    hipLaunchKernelGGL((decode_layers_kernel<MR, CHH, CHI>), dim3(cu_count()), dim3(WG_THREADS), lds, st, a);
    return vly_check_launch("vly_decode_layers");
}

unsigned long long* g_timing = nullptr;

}  // namespace

// debugging aid, not part of the ABI (tools/decode_phase_times.py): per-workgroup phase time stamps of the next launches
extern "C" void vlydbg_decode_timing(void* buf) { g_timing = (unsigned long long*)buf; }

extern "C" int vly_decode_layers_supported(int B, int H, int heads, int I) {
    const int chh = (H + 2047) / 2048, chi = (I + 2047) / 2048;
    return B >= 1 && B <= 2 && heads * 128 == H && H % 8 == 0 && I % 8 == 0 && ((chh == 2 && chi == 6) || (chh == 3 && chi == 7)) ? 1 : 0;
}

extern "C" int vly_decode_layers(const vly_decode_layer* layers_dev, int n_layers, float* h, void* qkv_scratch, float* partials,
                                 float* mlp_scratch, const float* cos_table, const float* sin_table, const uint8_t* key_valid,
                                 int key_valid_stride, const int32_t* pos_dev, int pos_stride, int B, int H, int heads, int I, float eps,
                                 int ctx_max, uint32_t* sync, void* stream) {
    if (!layers_dev || n_layers <= 0 || !h || !qkv_scratch || !partials || !mlp_scratch || !cos_table || !sin_table || !pos_dev || !sync ||
        pos_stride < 0 || pos_stride > 1 || ctx_max <= 0 || ((uintptr_t)h & 15) || ((uintptr_t)qkv_scratch & 15) || ((uintptr_t)partials & 15) ||
        ((uintptr_t)mlp_scratch & 15) || ((uintptr_t)sync & 63) || (key_valid && key_valid_stride < ctx_max)) {
        vly_set_error("vly_decode_layers: bad arguments (B=%d H=%d heads=%d I=%d ctx_max=%d)", B, H, heads, I, ctx_max);
        return -22;
    }
    if (!vly_decode_layers_supported(B, H, heads, I)) {
        vly_set_error("vly_decode_layers: unsupported shape B=%d H=%d heads=%d I=%d (B <= 2, heads * 128 = H, (H, I) in the 7B / 13B "
                      "classes: 2048 < H <= 4096 with 10240 < I <= 12288, or 4096 < H <= 6144 with 12288 < I <= 14336)", B, H, heads, I);
        return -22;
    }
    hipStream_t st = (hipStream_t)stream;
    static const int dl_stop = getenv("VLY_DL_STOP") ? atoi(getenv("VLY_DL_STOP")) : 0;
    Args a{layers_dev, n_layers, h, (uint32_t*)qkv_scratch, partials, mlp_scratch, cos_table, sin_table, key_valid, key_valid_stride,
           pos_dev, pos_stride, B, H, heads, I, eps, ctx_max, sync, g_timing, dl_stop};
    const int Kmax = I > H ? I : H;
    const int chh = (H + 2047) / 2048;
#define VLY_DL(MR)                                                                                                    \
    do {                                                                                                              \
        const size_t lds = (((size_t)MR * Kmax * 2 + 15) & ~(size_t)15) + ((size_t)2 * H + 2 * NGRP * (4 * MR * 2) + 8 + NGRP * ATTN_SCRATCH_FLOATS) * 4; \
        return chh == 2 ? launch<MR, 2, 6>(a, lds, st) : launch<MR, 3, 7>(a, lds, st);                               \
    } while (0)
    if (B == 1) VLY_DL(1);
    VLY_DL(2);
#undef VLY_DL
}

#endif  // VLY_EXPERIMENTAL
