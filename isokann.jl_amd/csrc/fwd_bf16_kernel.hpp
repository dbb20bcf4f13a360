// Split-bf16 ("bf16x3") variant of the fused MLP forward for gfx950.
//
// gfx950's f32-input MFMA runs at 1/16 of the bf16 MFMA rate.  Layer 1 (99 % of the flops) is therefore
// computed as an error-compensated bf16 product: every f32 operand is split into three bf16 planes
//      a = a0 + a1 + a2,   a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)      (24 mantissa bits)
// and a.b is accumulated in f32 from the six largest cross terms
//      a0b0 + (a0b1 + a1b0) + (a0b2 + a2b0 + a1b1)          dropped: a1b2, a2b1, a2b2  (<= 2^-24 relative)
// i.e. 6 x v_mfma_f32_16x16x32_bf16 per 16x16x32 block = 16/6 = 2.67x the f32 MFMA throughput at
// f32-equivalent accuracy (measured on K = 1712 dot products: rms error 6.0e-7 vs 5.1e-7 for an f32 BLAS).
//
// Structure: the three weight planes of a 32-feature chunk (3*NT KiB, fragment-linear, pre-split by
// fold_bf16_kernel) are staged through LDS once per workgroup and shared by its 8 waves (two per SIMD); the sample
// rows stay private to a wave: f32 global -> VGPR in fragment shape, split into planes in registers.  Tail layers,
// bias and activation are the f32 code shared with the f32 kernel (fwd_common.hpp).  The main loop is scheduled by
// hand -- see the comment above PipeGeom for what the first version of this kernel lost and why.
//
// PRE = true ("split-f16 x2", forward_mode 3): the same pipeline with BOTH operands pre-split into two scaled f16 planes
//      a * 2^s = hi + lo,  hi = f16(a 2^s), lo = f16(a 2^s - hi)                              (22 mantissa bits)
// -- the samples by the featurizer (FeatArgs.planes: per-row scale, 128 B per 32-feature chunk = [hi | lo], the same
// bytes as the f32 row), the weights by fold_f16_kernel (per-tensor scale) -- and a.b accumulated from THREE
// v_mfma_f32_16x16x32_f16 (lo.hi + hi.lo + hi.hi; dropped lo.lo <= 2^-24 relative), rescaled once in the epilogue.
// Half the matrix work of the bf16 split and no conversion VALU in the loop: the kernel becomes HBM-bound on the
// sample stream (4*Fp bytes per row).
#pragma once
#include <cstdlib>
#include <utility>
#include "fwd_common.hpp"

typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct Planes {
    bf16x8 p0, p1, p2;
};

#ifndef ISK_ABLATE
#define ISK_ABLATE 0
#endif
__device__ __forceinline__ Planes split3(const f32x8 v) {
    Planes s;
    s.p0 = __builtin_convertvector(v, bf16x8);
    const f32x8 r1 = v - __builtin_convertvector(s.p0, f32x8);
    s.p1 = __builtin_convertvector(r1, bf16x8);
    const f32x8 r2 = r1 - __builtin_convertvector(s.p1, f32x8);
    s.p2 = __builtin_convertvector(r2, bf16x8);
    return s;
}

#ifndef ISK_STAMP
#define ISK_STAMP 0
#endif
#if ISK_STAMP
#define STAMP(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); \
                      g_ph[k] += tn_ - g_t; g_t = tn_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------------
// The main loop.  In-kernel cycle stamps of the first version of this kernel (same data flow, compiler-scheduled:
// all of a stage's loads right after its barrier, conversion before the first MFMA; diagnostic build -DISK_STAMP)
// showed where a wave's time went: 35 % issuing MFMAs, 24 % blocked ISSUING the eight global
// loads every wave fires at once right after the barrier (the CU's vector-memory front end drains 64 B/clk),
// 15 % in the barrier, 13 % converting samples to planes before the first MFMA of a stage.  Both waves of a
// SIMD are phase-locked by the barrier, so none of that hides behind the other wave's MFMAs.  Here a stage is
// scheduled by hand, tile by tile (one tile = 16 outputs = 6 MFMAs per unit):
//   * the sample loads of chunk c+2, the LDS writes of weight chunk c+1, the global loads of weight chunk
//     c+2 and the f32 -> 3 x bf16 conversion of sample chunk c+1 are spread over the tiles of stage c, one or
//     two per tile, pinned with sched_barrier so the compiler cannot cluster them again;
//   * weight fragments are double-buffered in registers one tile ahead;
//   * the loop body is branch-free (vector-memory loads retire in order; with conditional loads in the body
//     hipcc falls back to s_waitcnt vmcnt(0) at the loop header): the chunk count is padded to an even number
//     (zero weight planes; sample loads clamped to the last real chunk), staging loads are clamped instead
//     of predicated and the clamped duplicates land in an LDS pad.
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT, int U, int NTHR, bool EARLY = false, int NPL = 3>
struct PipeGeom {
    static constexpr int STAGE16 = NPL * NT * 64;                         // 16-byte units staged per chunk
    static constexpr int PER_T = (STAGE16 + NTHR - 1) / NTHR;
    static constexpr int SLOTS = PER_T * NTHR;                          // LDS slots per buffer (>= STAGE16)
    static constexpr int NX = 2 * U;                                    // 16-byte sample loads / half-unit splits per chunk
    static constexpr int H = NT / 2;
    static constexpr int clampt(int t) { return t < NT ? t : NT - 1; }
    static constexpr int tile_x(int i) { return clampt((i * NT) / (2 * NX)); }                         // first half
    static constexpr int tile_dw(int i) { return clampt(((i + 1) * NT) / (2 * (PER_T + 1))); }          // first half
    static constexpr int tile_wl(int i) { return clampt(H + (i * (NT - H)) / PER_T); }                  // second half
    // sample conversion: second half of the stage; first half for the SIMD's other wave (EARLY), so that the
    // two waves' VALU bursts do not coincide and each runs under the other's MFMAs
    static constexpr int tile_sp(int i) { return EARLY ? clampt(((i + 1) * H) / (NX + 1)) : clampt(H + ((i + 1) * (NT - H)) / (NX + 1)); }
};

// half h (4 features) of a sample fragment -> the three plane halves
template <int HALF, int NPL = 3>
__device__ __forceinline__ void split_half(Planes& d, const f32x4v v) {
    const bf16x4 a0 = __builtin_convertvector(v, bf16x4);
    if (NPL == 1) {                                      // plain bf16 mode: the leading plane only
        if (HALF == 0) d.p0 = __builtin_shufflevector(a0, __builtin_shufflevector(d.p0, d.p0, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5, 6, 7);
        else d.p0 = __builtin_shufflevector(__builtin_shufflevector(d.p0, d.p0, 0, 1, 2, 3), a0, 0, 1, 2, 3, 4, 5, 6, 7);
        return;
    }
    const f32x4v r1 = v - __builtin_convertvector(a0, f32x4v);
    const bf16x4 a1 = __builtin_convertvector(r1, bf16x4);
    const f32x4v r2 = r1 - __builtin_convertvector(a1, f32x4v);
    const bf16x4 a2 = __builtin_convertvector(r2, bf16x4);
    if (HALF == 0) {
        d.p0 = __builtin_shufflevector(a0, __builtin_shufflevector(d.p0, d.p0, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5, 6, 7);
        d.p1 = __builtin_shufflevector(a1, __builtin_shufflevector(d.p1, d.p1, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5, 6, 7);
        d.p2 = __builtin_shufflevector(a2, __builtin_shufflevector(d.p2, d.p2, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5, 6, 7);
    } else {
        d.p0 = __builtin_shufflevector(__builtin_shufflevector(d.p0, d.p0, 0, 1, 2, 3), a0, 0, 1, 2, 3, 4, 5, 6, 7);
        d.p1 = __builtin_shufflevector(__builtin_shufflevector(d.p1, d.p1, 0, 1, 2, 3), a1, 0, 1, 2, 3, 4, 5, 6, 7);
        d.p2 = __builtin_shufflevector(__builtin_shufflevector(d.p2, d.p2, 0, 1, 2, 3), a2, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

template <int... Ts, class F>
__device__ __forceinline__ void static_for_tiles(std::integer_sequence<int, Ts...>, F&& f) {
    (f(std::integral_constant<int, Ts>{}), ...);
}

#define PX_CH 32
#define PX_H 4
// the sample stream is read once per launch (1.45 GB): non-temporal loads keep it from evicting the weight planes and
// the training working set (measured: step 1.647 -> 1.628 ms, cached-features mode 1.191 -> 1.163 ms)
#ifndef ISK_X_NT
#define ISK_X_NT 1
#endif
#if ISK_X_NT
#define X_LD(p) __builtin_nontemporal_load(p)
#else
#define X_LD(p) (*(p))
#endif
template <int NT, int U, int NTHR, bool EARLY, int NPL, bool PRE>
__device__ __forceinline__ void pipe_pass(const MlpFwdArgs& a, const u32x4* __restrict__ W16, u32x4* wbuf, const float* const (&xp)[U],
                                          int nch, int lane, f32x4 (&acc)[U][NT]
#if ISK_STAMP
                                          , unsigned long long (&g_ph)[8], unsigned long long& g_t
#endif
                                          ) {
    typedef PipeGeom<NT, U, NTHR, EARLY, NPL> G;
    constexpr int STAGE16 = G::STAGE16, PER_T = G::PER_T, NX = G::NX, SLOTS = G::SLOTS;
    constexpr int TBAR = G::tile_dw(PER_T - 1);          // the stage's barrier follows the tile with the last LDS write
    constexpr bool XPF = TBAR < NT - 1;                  // next stage's first fragments can be fetched before this stage ends
    // float offsets of a lane's second 16-byte load (the lo plane / the second half) and of the next chunk.  Pre-split
    // samples live unit-blocked (isk_plane_off): chunk = 2 KiB, lo plane 1 KiB behind hi
    constexpr int PXH = PRE ? 256 : PX_H;
    constexpr int XCH = PRE ? 512 : PX_CH;
    static_assert(!PRE || NPL == 2, "pre-split samples come as two f16 planes");
    const int tid = threadIdx.x;
    const int last = nch - 1;
    const int nchp = (nch + 1) & ~1;
    // global image: [chunk][plane][nt_total tiles][64]; the block stages tiles [tile0, tile0 + NT) of each plane
    const int ntt = a.dense_mode ? a.w_nt_total : NT;
    const int gchunk = NPL * ntt * 64;
    int goff[PER_T];
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
        int idx = i * NTHR + tid;
        if (idx > STAGE16 - 1) idx = STAGE16 - 1;                     // clamped duplicate, lands in the LDS pad
        const int p = idx / (NT * 64), rem = idx - p * (NT * 64);
        goff[i] = (p * ntt + (a.dense_mode ? a.w_tile0 : 0)) * 64 + rem;
    }
    // both prologue weight chunks in flight together: one exposed memory round trip per pass instead of two
    u32x4 wreg[PER_T], wfirst[PER_T];
#pragma unroll
    for (int i = 0; i < PER_T; ++i) wfirst[i] = W16[goff[i]];
#pragma unroll
    for (int i = 0; i < PER_T; ++i) wreg[i] = W16[gchunk + goff[i]];  // nchp >= 2
    f32x4v xr[2][U][2];                                              // [ring slot = chunk & 1][unit][half]
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            xr[0][u][h] = *(const f32x4v*)(xp[u] + PXH * h);
            xr[1][u][h] = *(const f32x4v*)(xp[u] + (last > 0 ? XCH : 0) + PXH * h);
        }
    __syncthreads();                                     // the previous pass's epilogue scratch aliases the buffers
#pragma unroll
    for (int i = 0; i < PER_T; ++i) wbuf[i * NTHR + tid] = wfirst[i];
    __syncthreads();
    Planes cur[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (PRE) {                                       // the loaded bytes ARE the operands
            cur[u].p0 = __builtin_bit_cast(bf16x8, xr[0][u][0]);
            cur[u].p1 = __builtin_bit_cast(bf16x8, xr[0][u][1]);
        } else {
            split_half<0, NPL>(cur[u], xr[0][u][0]);
            split_half<1, NPL>(cur[u], xr[0][u][1]);
        }
        nxt[u] = cur[u];
    }
    // consume every prologue load here, so the loop is entered with a clean vmcnt state and the waits inside
    // are the counted ones of the back edge
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" :: "v"(xr[1][u][0]), "v"(xr[1][u][1]));
#pragma unroll
    for (int i = 0; i < PER_T; ++i) asm volatile("" :: "v"(wreg[i]));
    // three weight buffers in a ring: stage c reads ring[c % 3] and fills ring[(c + 1) % 3]; ONE barrier per
    // stage, in the middle (after the last LDS write), so a stage boundary has no barrier and the first
    // fragments of the next stage are prefetched across it
    int rdi = 0;                                                     // c % 3
    bf16x8 wf[2][NPL];
    {
        const bf16x8* wl = (const bf16x8*)wbuf;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) wf[0][pl] = wl[(pl * NT) * 64 + lane];
    }
    STAMP(0);
    for (int c0 = 0; c0 < nchp; c0 += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int c = c0 + s;
            const int cx = c + 2 < last ? c + 2 : last;             // sample chunk fetched in this stage -> slot s
            const int cw = c + 2 < nchp - 1 ? c + 2 : nchp - 1;     // weight chunk fetched in this stage
            const u32x4* wc = W16 + (long long)cw * gchunk;
            const int wri = rdi == 2 ? 0 : rdi + 1;
            const bf16x8* wl = (const bf16x8*)(wbuf + rdi * SLOTS);
            const bf16x8* wln = (const bf16x8*)(wbuf + wri * SLOTS);
            u32x4* wr = wbuf + wri * SLOTS;
            if (!XPF) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) wf[(s * NT) & 1][pl] = wl[(pl * NT) * 64 + lane];
            }
            static_for_tiles(std::make_integer_sequence<int, NT>{}, [&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value;
                const int fb = (s * NT + t) & 1;
                if (t + 1 < NT) {
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) wf[fb ^ 1][pl] = wl[(pl * NT + t + 1) * 64 + lane];
                } else if (XPF) {
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) wf[fb ^ 1][pl] = wln[(pl * NT) * 64 + lane];
                }
                if (PRE) {
                    const f16x8 wh = __builtin_bit_cast(f16x8, wf[fb][0]), wl = __builtin_bit_cast(f16x8, wf[fb][NPL > 1 ? 1 : 0]);
#define F_(P_, W_) _Pragma("unroll") for (int u = 0; u < U; ++u) acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, cur[u].P_), W_, acc[u][t], 0, 0, 0);
#if ISK_ABLATE == 16          /* diagnostic: two of the three MFMAs, every operand still loaded (wrong results; what does the matrix work cost?) */
                    F_(p1, wl) F_(p0, wh)
#elif ISK_ABLATE == 17        /* diagnostic: one of the three, every operand still loaded */
                    F_(p0, wh)
                    _Pragma("unroll") for (int u = 0; u < U; ++u) asm volatile("" :: "v"(cur[u].p1));
                    asm volatile("" :: "v"(wl));
#else
                    F_(p1, wh) F_(p0, wl) F_(p0, wh)
#endif
#undef F_
                } else if (NPL == 1) {
#pragma unroll
                    for (int u = 0; u < U; ++u) acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[u].p0, wf[fb][0], acc[u][t], 0, 0, 0);
                } else {
                    const bf16x8 w0 = wf[fb][0], w1 = wf[fb][NPL > 1 ? 1 : 0], w2 = wf[fb][NPL > 2 ? 2 : 0];
                    // smallest terms first (an order with longer runs of MFMAs sharing the B operand measured the same)
#define T_(P_, W_) _Pragma("unroll") for (int u = 0; u < U; ++u) acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[u].P_, W_, acc[u][t], 0, 0, 0);
                    T_(p1, w1) T_(p2, w0) T_(p0, w2) T_(p1, w0) T_(p0, w1) T_(p0, w0)
#undef T_
                }
                // ---- this tile's share of the stage's side work
#pragma unroll
                for (int i = 0; i < NX; ++i)
                    if (G::tile_x(i) == t && ISK_ABLATE != 12) xr[s][i >> 1][i & 1] = X_LD((const f32x4v*)(xp[i >> 1] + cx * XCH + PXH * (i & 1)));
#pragma unroll
                for (int i = 0; i < PER_T; ++i)
                    if (G::tile_dw(i) == t && ISK_ABLATE != 14) wr[i * NTHR + tid] = wreg[i];
#pragma unroll
                for (int i = 0; i < PER_T; ++i)
                    if (G::tile_wl(i) == t && ISK_ABLATE != 14) wreg[i] = wc[goff[i]];
#pragma unroll
                for (int i = 0; i < NX; ++i)
                    if (!PRE && G::tile_sp(i) == t && ISK_ABLATE != 13) {
                        if (i & 1) split_half<1, NPL>(nxt[i >> 1], xr[s ^ 1][i >> 1][1]);
                        else split_half<0, NPL>(nxt[i >> 1], xr[s ^ 1][i >> 1][0]);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (t == TBAR) {
                    STAMP(1);
                    if (ISK_ABLATE != 11) __syncthreads();
                    STAMP(2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (PRE) {                                   // chunk c + 1 landed in the other ring slot during stage c - 1
                    cur[u].p0 = __builtin_bit_cast(bf16x8, xr[s ^ 1][u][0]);
                    cur[u].p1 = __builtin_bit_cast(bf16x8, xr[s ^ 1][u][1]);
                } else cur[u] = nxt[u];
            }
            rdi = wri;
        }
    }
    STAMP(3);
    __syncthreads();                                     // every wave is done reading before the epilogue scratch overwrites the ring
    STAMP(4);
}

template <int NT, int U, int NWV, bool EARLY, int NPL, bool DENSE, bool PRE>
__device__ __forceinline__ void pipe_run(const MlpFwdArgs& a, const uint4* __restrict__ W16, float* smem, long long u0, int nact, int col_shift
#if ISK_STAMP
                                         , unsigned long long (&g_ph)[8], unsigned long long& g_t
#endif
                                         ) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    u32x4* wbuf = (u32x4*)(smem + a.tailw_floats);                  // ring of three chunk buffers
    // the per-wave epilogue scratch ALIASES the weight buffers: it is only used between the last barrier
    // of a pass and the first barrier of the next one
    float* scrA = smem + a.tailw_floats + (size_t)wave * 16 * (a.stride_a + a.stride_b);
    float* scrB = scrA + 16 * a.stride_a;
    const float* xp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        long long r = (u0 + u) * 16 + li;
        if (r > a.S - 1) r = a.S - 1;
        if (r < 0) r = 0;
        if (a.rows) r = a.rows[r];
        // pre-split planes: the lane's 16 bytes of row r inside the unit's KiB (16 rows x 64 B) -- the 64 lanes of a load
        // cover that KiB exactly: 8 whole lines
        xp[u] = PRE ? a.X + isk_plane_off(a.x_row0 + r, a.Fp) + lq * 4 : a.X + r * (long long)a.Fp + lq * 8;
    }
    // pre-split operands: the inverse scales of this lane's four accumulator rows (row = 4 lq + r), fetched before the
    // pass so the epilogue does not wait on them
    float rsc[U][4];
    if (PRE) {
        const float wsc = a.wscale[0];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                long long r = (u0 + u) * 16 + lq * 4 + r4;
                if (r > a.S - 1) r = a.S - 1;
                if (r < 0) r = 0;
                if (a.rows) r = a.rows[r];
                rsc[u][r4] = a.xscale[r] * wsc;
            }
    }
    f32x4 acc[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[u][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // a wave without a unit in this pass computes on clamped rows and discards the result
#if ISK_STAMP
    pipe_pass<NT, U, NWV * 64, EARLY, NPL, PRE>(a, (const u32x4*)W16, wbuf, xp, a.Fp / 32, lane, acc, g_ph, g_t);
#else
    pipe_pass<NT, U, NWV * 64, EARLY, NPL, PRE>(a, (const u32x4*)W16, wbuf, xp, a.Fp / 32, lane, acc);
#endif
    if (PRE) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) acc[u][t][r4] *= rsc[u][r4];
    }
    // spelled out: with the unit index a loop variable hipcc may decline to unroll the (large) epilogue and then
    // keeps the whole accumulator array in scratch memory
    if (0 < nact) unit_tail<NT, DENSE>(a, acc[0], u0 * 16, scrA, scrB, smem, lane, col_shift);
    if (U > 1 && 1 < nact) unit_tail<NT, DENSE>(a, acc[U - 1], (u0 + 1) * 16, scrA, scrB, smem, lane, col_shift);
    static_assert(U <= 2, "two units per pass");
    STAMP(5);
}

template <int NT, int NWV, int NPL, bool DENSE, bool PRE = false>
__global__ __launch_bounds__(NWV * 64) void mlp_fwd_bf16x3p_kernel(MlpFwdArgs a, const uint4* __restrict__ W16, long long units,
                                                                   long long units_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    long long vblock = blockIdx.x;
    int col_shift = 0;                       // the argument block itself stays untouched (a modified copy would live in scratch)
    if (DENSE && a.dense_groups > 1) {
        // grouped column blocks: block id = xcd + 8 * (group + G * j)  ->  row share xcd + 8 j, column block `group`
        const int G = a.dense_groups;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int grp = slot % G;
        vblock = (long long)(slot / G) * 8 + xcd;
        col_shift = grp * a.nt1 * 16;
        W16 += (long long)grp * a.nt1 * 64;                          // = w_tile0 + grp * nt1 in the plane image
        if (vblock * NWV * units_per_wave >= units) return;          // workgroup-uniform: a share without rows
    }
    stage_tail_weights(a, smem, col_shift);
    __syncthreads();
    const long long gw = vblock * NWV + (threadIdx.x >> 6);
    const long long ustart = gw * units_per_wave;
    long long uend = ustart + units_per_wave;
    if (uend > units) uend = units;
    const long long npass = (units_per_wave + 1) / 2;              // identical for every wave: barriers inside
    // a workgroup's waves go to the SIMDs cyclically, so waves w and w + 4 share one: give them different schedules
    static_assert(NWV == 4 || NWV == 8, "wave count");
    const bool early = !PRE && NWV == 8 && (threadIdx.x >> 8) != 0 && a.sched_skew;    // PRE: no conversion work to skew
#if ISK_STAMP
    unsigned long long g_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long g_t0 = __builtin_amdgcn_s_memtime(), g_rt0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long g_t = g_t0;
    for (long long p = 0; p < npass; ++p) {
        const long long u = ustart + 2 * p;
        if (u + 1 < uend) {
            if (early) pipe_run<NT, 2, NWV, true, NPL, DENSE, PRE>(a, W16, smem, u, 2, col_shift, g_ph, g_t);
            else pipe_run<NT, 2, NWV, false, NPL, DENSE, PRE>(a, W16, smem, u, 2, col_shift, g_ph, g_t);
        } else pipe_run<NT, 1, NWV, false, NPL, DENSE, PRE>(a, W16, smem, u, u < uend ? 1 : 0, col_shift, g_ph, g_t);
    }
    if (a.dbg && (threadIdx.x & 63) == 0) {
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* d = a.dbg + gw * 12;
        for (int k = 0; k < 8; ++k) d[k] = g_ph[k];
        d[8] = __builtin_amdgcn_s_memtime() - g_t0; d[9] = rt1 - g_rt0;
    }
#else
    for (long long p = 0; p < npass; ++p) {
        const long long u = ustart + 2 * p;
        if (u + 1 < uend) {                                                            // wave-uniform
            if (early) pipe_run<NT, 2, NWV, true, NPL, DENSE, PRE>(a, W16, smem, u, 2, col_shift);
            else pipe_run<NT, 2, NWV, false, NPL, DENSE, PRE>(a, W16, smem, u, 2, col_shift);
        } else pipe_run<NT, 1, NWV, false, NPL, DENSE, PRE>(a, W16, smem, u, u < uend ? 1 : 0, col_shift);
    }
#endif
}

template <int NT, int NWV, int NPL, bool DENSE>
static size_t pipe_lds_bytes(const MlpFwdArgs& a) {
    // dense blocks store straight from the accumulators: no per-wave epilogue scratch
    const size_t wb = (size_t)3 * PipeGeom<NT, 2, NWV * 64, false, NPL>::SLOTS * 4;
    const size_t scr = DENSE ? 0 : (size_t)NWV * 16 * (a.stride_a + a.stride_b);
    return ((size_t)a.tailw_floats + (wb > scr ? wb : scr)) * sizeof(float);
}

template <int NT, int NWV, int NPL, bool DENSE, bool PRE = false>
static void launch_pipe_nw(const MlpFwdArgs& a, const void* W16, hipStream_t st) {
    const long long units = (a.S + 15) / 16;
    const int G = (DENSE && a.dense_groups > 1) ? a.dense_groups : 1;
    const long long waves_max = (long long)(ISK_NUM_CU / G) * NWV;
    long long upw = (units + waves_max - 1) / waves_max;
    if (upw < 1) upw = 1;
    const long long nwaves = (units + upw - 1) / upw;
    long long blocks = (nwaves + NWV - 1) / NWV;
    if (G > 1) blocks = (blocks + 7) / 8 * 8 * G;               // whole XCD rounds of row shares, G column blocks each
    const size_t lds = pipe_lds_bytes<NT, NWV, NPL, DENSE>(a);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)mlp_fwd_bf16x3p_kernel<NT, NWV, NPL, DENSE, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((mlp_fwd_bf16x3p_kernel<NT, NWV, NPL, DENSE, PRE>), dim3((unsigned)blocks), dim3(NWV * 64), lds, st, a, (const uint4*)W16, units, upw);
}

template <int NT, int NPL, bool DENSE>
static int launch_pipe_nt(const MlpFwdArgs& a, const void* W16, hipStream_t st) {
    const size_t cap = 158 * 1024;
    if (pipe_lds_bytes<NT, 8, NPL, DENSE>(a) <= cap) { launch_pipe_nw<NT, 8, NPL, DENSE>(a, W16, st); return 0; }
    if (NPL == 3 && !DENSE) {     // the fused split kernel's per-wave epilogue scratch can outgrow the ring with 8 waves
        if (pipe_lds_bytes<NT, 4, 3, false>(a) <= cap) { launch_pipe_nw<NT, 4, 3, false>(a, W16, st); return 0; }
    }
    return ISK_EUNSUPPORTED;
}

// split-f16 x2 (pre-split samples): fused networks only
template <int NT>
static int launch_pipe_f16(const MlpFwdArgs& a, const void* W16, hipStream_t st, bool dry = false) {
    const size_t cap = 158 * 1024;
    if (pipe_lds_bytes<NT, 8, 2, false>(a) <= cap) { if (!dry) launch_pipe_nw<NT, 8, 2, false, true>(a, W16, st); return 0; }
    if (pipe_lds_bytes<NT, 4, 2, false>(a) <= cap) { if (!dry) launch_pipe_nw<NT, 4, 2, false, true>(a, W16, st); return 0; }
    return ISK_EUNSUPPORTED;
}
