// Forward-path kernels for gfx950 (MI355X): featurizer (pdist + LayerNorm standardisation),
// parameter fold, fused MLP forward (f32 MFMA), K-mean, min/max, shift-scale, ISA apply.
//
// Reference semantics restated (files relative to the reference repo):
//   pdist            src/utils/pairdists.jl:109-127 (pdists), :6-24 (flatpairdists, halfinds order)
//   LayerNorm/Dense  src/models.jl:87-92 (Flux.LayerNorm, Flux.Dense)
//   K-mean           src/isotarget.jl:18 (expectation)
//   shift-scale      src/isotarget.jl:36-42
#include <cstdlib>
#include "isk_internal.h"

#include "fwd_common.hpp"

// =====================================================================================
// featurize: one wave per sample.  The sample's 3A coordinates are staged in LDS
// (coalesced HBM read of the ys slab), every lane gathers its atom pairs from LDS,
// the raw distances are stashed in LDS, the wave reduces mean / variance with
// shuffles (two-pass variance, as Flux's var(mean=mu)), then writes the standardised,
// zero-padded row with 16-B stores.
// Algorithmic bytes per sample: 4*D read + 4*Fp written.  HBM-bound.
// =====================================================================================
template <bool PDIST>
__global__ __launch_bounds__(256) void featurize_kernel(FeatArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Dp = PDIST ? ((a.D + 3) & ~3) : 0;
    float* coords = smem + wave * (Dp + a.Fp);
    float* stash = coords + Dp;
    const long long nw = (long long)gridDim.x * 4;
    for (long long s0 = (long long)blockIdx.x * 4; s0 < a.S; s0 += nw) {
        const long long s = s0 + wave;
        const bool active = s < a.S;                 // wave-uniform
        const float* row = a.in + (active ? s : 0) * (long long)a.D;
        if (PDIST) {
            if (active)
                for (int i = lane; i < a.D; i += WAVE) coords[i] = row[i];
        }
        __syncthreads();
        float sum = 0.f;
        if (active) {
            for (int f = lane; f < a.F; f += WAVE) {
                float d;
                if (PDIST) {
                    const int2 p = a.pairs[f];
                    const float dx = coords[3 * p.x] - coords[3 * p.y];
                    const float dy = coords[3 * p.x + 1] - coords[3 * p.y + 1];
                    const float dz = coords[3 * p.x + 2] - coords[3 * p.y + 2];
                    d = sqrtf(dx * dx + dy * dy + dz * dz);
                } else {
                    d = row[f];
                }
                stash[f] = d;
                sum += d;
            }
            for (int f = a.F + lane; f < a.Fp; f += WAVE) stash[f] = 0.f;
        }
        float mu = 0.f, rstd = 1.f;
        if (a.normalize) {
            mu = wave_sum(sum) / (float)a.F;
            float ss = 0.f;
            if (active)
                for (int f = lane; f < a.F; f += WAVE) {
                    const float t = stash[f] - mu;
                    ss += t * t;
                }
            const float var = wave_sum(ss) / (float)a.F;
            if (a.ln_mode == ISK_LN_VAR_EPS2) rstd = 1.0f / sqrtf(var + a.eps * a.eps);
            else if (a.ln_mode == ISK_LN_STD_EPS) rstd = 1.0f / (sqrtf(var) + a.eps);
            else rstd = 1.0f / sqrtf(var + a.eps);
        }
        __syncthreads();
        if (active) {
            float* orow = a.xhat + s * (long long)a.Fp;
            if (a.stats && lane == 0) { a.stats[2 * s] = mu; a.stats[2 * s + 1] = rstd; }          // kept for the coordinate pullback, as the
            for (int c = lane * 4; c < a.Fp; c += WAVE * 4) {                                         // register variant does
                f32x4 v = *(const f32x4*)(stash + c);
                if (a.raw) *(f32x4*)(a.raw + s * (long long)a.Fp + c) = v;                          // (un-normalised features, zero padding)
                if (a.normalize) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (c + j < a.F) ? (v[j] - mu) * rstd : 0.f;
                }
                *(f32x4*)(orow + c) = v;
                if (a.xhat32 && s >= a.x32_first) *(f32x4*)(a.xhat32 + (s - a.x32_first) * (long long)a.Fp + c) = v;
            }
        }
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------
// featurize, register-resident variant (the one that normally runs): persistent waves, one
// sample per wave per trip, NO workgroup barrier.  Each lane keeps its <= NFL (packed 16-bit)
// atom pairs and its NFL distances in registers across the whole sample loop; the sample's
// coordinates are staged in a wave-private LDS region padded to xyzw so a pair costs two
// ds_read_b96, kept several pairs deep in flight; mean / variance by DPP wavefront reductions; the standardised row
// goes out as coalesced 256-B buffer stores.
// -------------------------------------------------------------------------------------
// the coordinate reads and feature writes are pure streams (1.4 GB each per iteration): non-temporal, so they do not
// evict what the latency-bound training kernels keep re-reading (xs features, weights) from L2 / MALL -- measured:
// training phase beside the featurizer 0.96 -> 0.91 ms
#ifndef ISK_FEAT_NT
#define ISK_FEAT_NT 1
#endif
#define FEAT_AUX (ISK_FEAT_NT ? 2 : 0)         /* buffer-instruction cache policy: nt */
#if ISK_FEAT_NT
#define FEAT_LD(p) __builtin_nontemporal_load(p)
#define FEAT_ST(p, v) __builtin_nontemporal_store((v), (p))
#else
#define FEAT_LD(p) (*(p))
#define FEAT_ST(p, v) (*(p) = (v))
#endif
// 12-byte buffer load, declared the way composable_kernel does: this hipcc lowers the __builtin_amdgcn_raw_buffer_load_b64 /
// _b96 / _b128 builtins to a single-dword load (checked in the ISA), the intrinsic itself is fine
typedef int isk_i32x4 __attribute__((ext_vector_type(4)));
typedef float isk_f32x3 __attribute__((ext_vector_type(3)));
__device__ isk_f32x3 isk_raw_buffer_load_f32x3(isk_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v3f32");

typedef float isk_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 isk_f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 isk_bf16x2 __attribute__((ext_vector_type(2)));
__device__ void isk_raw_buffer_store_f32x2(isk_f32x2 v, isk_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2f32");
__device__ void isk_raw_buffer_store_u32(unsigned v, isk_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.i32");

// max over the 64 lanes of NON-NEGATIVE values (0 is the identity the DPP fill relies on), same DPP ladder as wave_sum_dpp
__device__ __forceinline__ float wave_max_dpp(float v) {
#define ISK_DPP_MAX(ctrl, rmask)                                                                                   \
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false)))
    ISK_DPP_MAX(0x111, 0xf);
    ISK_DPP_MAX(0x112, 0xf);
    ISK_DPP_MAX(0x114, 0xf);
    ISK_DPP_MAX(0x118, 0xf);
    ISK_DPP_MAX(0x142, 0xa);
    ISK_DPP_MAX(0x143, 0xc);
#undef ISK_DPP_MAX
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Feature <-> register mapping: lane l keeps the ADJACENT features f = 128*(i/2) + 2*l + (i%2), i < NFL, so a row goes
// out as 8-byte stores (512 contiguous bytes per wave instruction) and, in PLANES mode, an f16 pair is one dword.
//
// PLANES (forward_mode 3): the standardised row is written pre-split for the split-f16 forward GEMM
// (fwd_f16 path in fwd_bf16_kernel.hpp): per 32-feature chunk 64 B of hi and 64 B of lo with
//     xs = xhat * 2^s (row scale, maximum in [2^14, 2^15)),  hi = f16(xs),  lo = f16(xs - hi)      (22 mantissa bits)
// and 2^-s goes to xscale[row].  Same bytes per row as the f32 image (4*Fp); the forward kernel then loads MFMA A
// operands straight from HBM with no conversion work.  The rows are stored UNIT-BLOCKED (FeatArgs.row0, isk_plane_off):
// 16 consecutive rows share, per chunk, one 2 KiB block [hi: 16 x 64 B | lo: 16 x 64 B], because what the forward kernel
// pays for is the number of 128-byte lines one load instruction touches (measured with address-only ablation builds of
// the forward: 16 rows x 64 B of a row-major image = half of 16 lines per instruction 0.39 ms, the same bytes as one
// contiguous KiB = 8 whole lines 0.29 ms; quad-contiguous or same-DRAM-page variants that still touch 16 lines gain
// little).  A line of that image holds 64 B of TWO neighbouring rows: a wave therefore takes the rows in pairs
// (2j, 2j+1 in consecutive trips), parks the first row's plane dwords in a wave-private LDS stash and stores both rows
// together behind the second one -- the two halves of every line leave in consecutive store instructions of one wave,
// as the hi / lo halves of a row-major line did (stored a row apart -- 3.5 us under the throttle -- the half lines were
// written back separately: featurizer 0.73 -> 0.82 ms, training beside it 0.69 -> 0.78).
// TIGHT: F > 64 (NFL - 2), i.e. only a lane's LAST feature pair can lie past the end of the row -- the validity select
// (one per feature and use otherwise) is compiled for those two registers only.  The kernel runs throttled to one wave
// per SIMD beside the training kernels, where its time is its instruction count (~930 per row before this, the selects
// and the predicated coordinate writes ~150 of them).
template <int NFL, int NDL, bool PDIST, int PF = 1, bool PLANES = false, bool TIGHT = false>
__global__ __launch_bounds__(256) void featurize_reg_kernel(FeatArgs a, const unsigned* __restrict__ packed_pairs) {
    static_assert(NFL % 2 == 0, "a lane holds adjacent feature pairs");
    constexpr int NPB = NFL / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int natoms = a.D / 3;
    f32x4* cw = (f32x4*)smem + (size_t)wave * (natoms + 1);  // wave-private xyzw coordinates (+ one cell that takes the lanes past the last atom)
    // PLANES: behind the coordinates, per wave NPB x 64 (hi, lo) dword pairs of the pair's first row
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2* stash = (u32x2*)((f32x4*)smem + (PDIST ? 4 * (size_t)(natoms + 1) : 0)) + (size_t)wave * (NPB * 64);
    // feature i of this lane exists (padding features: pair (0, 0) / a load past the row's end, i.e. d == 0)
    auto ok = [&](int i) { return (TIGHT && i < NFL - 2) ? true : 128 * (i >> 1) + 2 * lane + (i & 1) < a.F; };
    unsigned pr[NFL];
    if (PDIST) {
#pragma unroll
        for (int i = 0; i < NFL; ++i) {
            const int f = 128 * (i >> 1) + 2 * lane + (i & 1);
            pr[i] = (f < a.F) ? packed_pairs[f] : 0u;
        }
    }
    const float invF = 1.0f / (float)a.F;
    const long long nw = (long long)gridDim.x * 4;
    const long long w0 = (long long)blockIdx.x * 4 + wave;
    // trip t of this wave -> row.  PLANES: rows in pairs (see above); else row w0 + t nw
    auto row_of = [&](long long t) { return PLANES ? 2 * (w0 + (t >> 1) * nw) + (t & 1) : w0 + t * nw; };
    // software pipeline: the next PF samples' rows (NDL or NFL dwords per lane each) are in flight in registers while
    // the current one is featurized, so a wave keeps HBM busy on its own (needed when the kernel runs
    // throttled beside the training kernels)
    // pairwise distances: a lane takes whole atoms (xyz = one 12-B load, one ds_write_b96 into its 16-B cell) -- a third
    // of the load and LDS-write instructions of a dword-per-lane row
    constexpr int NAL = (NDL * 64 / 3 + 63) / 64;                  // atoms per lane
    constexpr int NIN = PDIST ? 3 * NAL : NFL;
    typedef float f32x3 __attribute__((ext_vector_type(3)));
    float nxt[PF][NIN];
    // Rows are read and written through BUFFER instructions: a wave-uniform descriptor per row whose size is the
    // row length (0 for a row past the end), so lanes beyond the row read 0 and their stores are dropped by the
    // hardware.  That keeps the loop free of branches around memory instructions -- with them hipcc's wait-count
    // pass can only emit vmcnt(0), which made every wave sit out a full load + store round trip per sample (the
    // "prefetched" row included).
    auto row_rv = [&](const float* base, long long srow, long long ld, int len) {
        const unsigned long long p = (unsigned long long)(base + srow * ld);
        isk_i32x4 rv;
        rv[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
        rv[1] = __builtin_amdgcn_readfirstlane((int)((p >> 32) & 0xffffu));
        rv[2] = __builtin_amdgcn_readfirstlane(srow < a.S ? len * 4 : 0);
        rv[3] = 0x00020000;
        return rv;
    };
    auto fetch = [&](float (&buf)[NIN], long long srow) {
        if (PDIST) {
            const isk_i32x4 rv = row_rv(a.in, srow, a.D, a.D);
#pragma unroll
            for (int u = 0; u < NAL; ++u) {
                const isk_f32x3 v = isk_raw_buffer_load_f32x3(rv, (64 * u + lane) * 12, 0, FEAT_AUX);
                buf[3 * u] = v[0];
                buf[3 * u + 1] = v[1];
                buf[3 * u + 2] = v[2];
            }
        } else {
            // dword loads (a row of odd length would make the last 8-byte access straddle the descriptor's end)
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + srow * (long long)a.D), 0,
                                                                               srow < a.S ? a.D * 4 : 0, 0x00020000);
#pragma unroll
            for (int u = 0; u < NIN; ++u)
                buf[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (128 * (u >> 1) + 2 * lane + (u & 1)) * 4, 0, FEAT_AUX));
        }
    };
    auto plane_rv = [&](long long srow) {
        // descriptor over the rest of the row's unit block, starting at the row's 64-byte slot of chunk 0: chunks at or
        // past Fp/32 (the lanes' padding features) fall outside it and are dropped
        const long long g = a.row0 + srow;
        const unsigned long long p = (unsigned long long)(a.xhat + isk_plane_off(g, a.Fp));
        isk_i32x4 rv;
        rv[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
        rv[1] = __builtin_amdgcn_readfirstlane((int)((p >> 32) & 0xffffu));
        rv[2] = __builtin_amdgcn_readfirstlane(srow < a.S ? 64 * a.Fp - (int)(g & 15) * 64 : 0);
        rv[3] = 0x00020000;
        return rv;
    };
    auto one = [&](float (&buf)[NIN], long long srow, long long srow_refill, const int second) {
        float d[NFL];
        if (PDIST) {
#pragma unroll
            for (int u = 0; u < NAL; ++u) {                 // unpredicated: lanes past the last atom (they loaded zeros) share a spare cell
                const int at = 64 * u + lane;
                ((f32x3*)cw)[at < natoms ? at : natoms] = (f32x3){buf[3 * u], buf[3 * u + 1], buf[3 * u + 2]};
            }
        } else {
#pragma unroll
            for (int i = 0; i < NFL; ++i) d[i] = buf[i];
        }
        fetch(buf, srow_refill);                             // refill this slot: the sample PF trips ahead
        if (PDIST) {
            __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS writes have landed
            __builtin_amdgcn_wave_barrier();
            // the gathers are kept GW pairs deep in flight (a rolling register window, order pinned): left to itself
            // the compiler waits out the full LDS latency of every pair, 28 times per sample
            constexpr int GW = NFL < 6 ? NFL : 6;
            // 16-byte cells, xyz read (ds_read_b96)
            const f32x3* cw3 = (const f32x3*)cw;
            f32x3 qa[GW], qb[GW];
#pragma unroll
            for (int i = 0; i < GW; ++i) { qa[i] = cw3[pr[i] & 0xffffu]; qb[i] = cw3[pr[i] >> 16]; }
#pragma unroll
            for (int i = 0; i < NFL; ++i) {
                __builtin_amdgcn_sched_barrier(0);
                const f32x3 pa = qa[i % GW], pb = qb[i % GW];
                // (x, y) as one packed subtract and one packed multiply: 5 VALU operations per distance instead of 6
                const isk_f32x2 dxy = (isk_f32x2){pa[0], pa[1]} - (isk_f32x2){pb[0], pb[1]};
                const float dz = pa[2] - pb[2];
                if (i + GW < NFL) { qa[i % GW] = cw3[pr[i + GW] & 0xffffu]; qb[i % GW] = cw3[pr[i + GW] >> 16]; }
                const isk_f32x2 sq = dxy * dxy;
                d[i] = __builtin_amdgcn_sqrtf(fmaf(dz, dz, sq[0]) + sq[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_wave_barrier();             // all reads issued before the next trip's writes
        }
        float mu = 0.f, rstd = 1.f;
        float tt[NFL];                                       // d - mu, 0 for padding features
        if (a.normalize) {
            // a lane's feature pairs as packed operands: half the adds / fmas
            isk_f32x2 sum2 = {0.f, 0.f};
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) sum2 += (isk_f32x2){d[2 * pb], d[2 * pb + 1]};      // padding features are exactly 0
            mu = wave_sum_dpp(sum2[0] + sum2[1]) * invF;
            isk_f32x2 ss2 = {0.f, 0.f};
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
                isk_f32x2 t2 = (isk_f32x2){d[2 * pb], d[2 * pb + 1]} - (isk_f32x2){mu, mu};
                if (!ok(2 * pb)) t2[0] = 0.f;
                if (!ok(2 * pb + 1)) t2[1] = 0.f;
                tt[2 * pb] = t2[0]; tt[2 * pb + 1] = t2[1];
                ss2 += t2 * t2;
            }
            const float var = wave_sum_dpp(ss2[0] + ss2[1]) * invF;
            if (a.ln_mode == ISK_LN_VAR_EPS2) rstd = 1.0f / sqrtf(var + a.eps * a.eps);
            else if (a.ln_mode == ISK_LN_STD_EPS) rstd = 1.0f / (sqrtf(var) + a.eps);
            else rstd = 1.0f / sqrtf(var + a.eps);
        } else {
#pragma unroll
            for (int i = 0; i < NFL; ++i) tt[i] = d[i];
        }
        if (PLANES) {
            // xhat = tt * rstd; its row maximum is rstd * max|tt| (rounding is monotone), and the plane scale 2^s folds into
            // one factor with rstd (a power of two: the same bits as scaling xhat afterwards)
            float m = 0.f;
#pragma unroll
            for (int i = 0; i < NFL; ++i) m = fmaxf(m, fabsf(tt[i]));
            float sc, inv;
            isk_f16_scale(wave_max_dpp(m) * rstd, sc, inv);
            const float s2 = rstd * sc;
            // no arrays of converted values here: the kernel sits at the register limit that lets two 128-register training
            // waves share a SIMD with it (248 of 512), and 16 more registers cost the training chain 40 % beside it
            if (!second) {                                   // wave-uniform: first row of the pair -> stash
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb) {
                    const isk_f32x2 sv = {tt[2 * pb] * s2, tt[2 * pb + 1] * s2};
                    const isk_f16x2 hi = __builtin_convertvector(sv, isk_f16x2);
                    const isk_f16x2 lo = __builtin_convertvector(sv - __builtin_convertvector(hi, isk_f32x2), isk_f16x2);
                    stash[pb * 64 + lane] = (u32x2){__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo)};
                }
            } else {
                const isk_i32x4 ra = plane_rv(srow - 1), rb = plane_rv(srow);
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb) {
                    const u32x2 fa = stash[pb * 64 + lane];                            // same lane wrote it: no barrier needed
                    const isk_f32x2 sv = {tt[2 * pb] * s2, tt[2 * pb + 1] * s2};
                    const isk_f16x2 hi = __builtin_convertvector(sv, isk_f16x2);
                    const isk_f16x2 lo = __builtin_convertvector(sv - __builtin_convertvector(hi, isk_f32x2), isk_f16x2);
                    const int off = (4 * pb + (lane >> 4)) * 2048 + (lane & 15) * 4;  // chunk 4 pb + lane/16, pair lane%16
                    isk_raw_buffer_store_u32(fa[0], ra, off, 0, FEAT_AUX);
                    isk_raw_buffer_store_u32(__builtin_bit_cast(unsigned, hi), rb, off, 0, FEAT_AUX);
                    isk_raw_buffer_store_u32(fa[1], ra, off + 1024, 0, FEAT_AUX);
                    isk_raw_buffer_store_u32(__builtin_bit_cast(unsigned, lo), rb, off + 1024, 0, FEAT_AUX);
                }
            }
            if (lane == 0 && srow < a.S) a.xscale[srow] = inv;
        } else if (a.out_bf16) {                             // wave-uniform: bf16 rows (2 Fp bytes each) for the fused wide forward
            const isk_i32x4 ro = row_rv(a.xhat, srow, (a.ld16 ? a.ld16 : a.Fp) / 2, a.Fp / 2);
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
                const isk_bf16x2 h = __builtin_convertvector((isk_f32x2){tt[2 * pb] * rstd, tt[2 * pb + 1] * rstd}, isk_bf16x2);
                isk_raw_buffer_store_u32(__builtin_bit_cast(unsigned, h), ro, (128 * pb + 2 * lane) * 2, 0, FEAT_AUX);
            }
        } else {
            const isk_i32x4 ro = row_rv(a.xhat, srow, a.Fp, a.Fp);
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb)
                isk_raw_buffer_store_f32x2((isk_f32x2){tt[2 * pb] * rstd, tt[2 * pb + 1] * rstd}, ro, (128 * pb + 2 * lane) * 4, 0, FEAT_AUX);
        }
        if (a.xhat32 && srow >= a.x32_first) {               // wave-uniform: a second, f32 copy (the xs rows the training kernels read)
            const isk_i32x4 r32 = row_rv(a.xhat32, srow - a.x32_first, a.Fp, srow < a.S ? a.Fp : 0);
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb)
                isk_raw_buffer_store_f32x2((isk_f32x2){tt[2 * pb] * rstd, tt[2 * pb + 1] * rstd}, r32, (128 * pb + 2 * lane) * 4, 0, 0);
        }
        if (a.raw) {                                         // wave-uniform: kept for the coordinate pullback
            const isk_i32x4 rr = row_rv(a.raw, srow, a.Fp, a.Fp);
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
                const int f = 128 * pb + 2 * lane;
                isk_raw_buffer_store_f32x2((isk_f32x2){d[2 * pb], d[2 * pb + 1]}, rr, f * 4, 0, 0);     // padding features are exactly 0
            }
        }
        if (a.stats && lane == 0 && srow < a.S) { a.stats[2 * srow] = mu; a.stats[2 * srow + 1] = rstd; }
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) fetch(nxt[p], row_of(p));
    // rows past the end (the last trips) run on empty descriptors; PLANES: a pair is entered when its even row exists
    for (long long t = 0; row_of(PLANES ? (t & ~1LL) : t) < a.S; t += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) one(nxt[p], row_of(t + p), row_of(t + p + PF), (int)((t + p) & 1));
    }
}

template <int NFL, int NDL>
static void launch_feat_reg(const FeatArgs& a, const unsigned* packed, hipStream_t st) {
    const bool pd = a.pairs != nullptr;
    // wave-private coordinate cells + (planes) the stash of a row pair's first row
    const size_t lds = (pd ? (size_t)4 * (a.D / 3 + 1) * 16 : 0) + (a.planes ? (size_t)4 * (NFL / 2) * 64 * 8 : 0);
    const bool tight = a.F > 64 * (NFL - 2);     // only the last feature pair of a lane can be padding
    // waves per CU bounded by the wave-private LDS image (160 KiB) and by 16 (4 per SIMD)
    const int cap_env = a.bg_wg_per_cu > 0 ? a.bg_wg_per_cu : 1;       // engine option feat_wg_per_cu
    // background launches (beside the training kernels) run ONE self-pipelined workgroup per CU: measured
    // 1.74 ms/step against 1.87 with two
    int wg_per_cu = (a.background && cap_env > 0) ? cap_env : 4;
    if (lds > 0) {
        const int fit = (int)((150 * 1024) / lds);
        if (fit < wg_per_cu) wg_per_cu = fit < 1 ? 1 : fit;
    }
    long long blocks = (a.S + 3) / 4;
    long long cap = (long long)ISK_NUM_CU * wg_per_cu;
    const int bg_total = a.bg_wgs_total;                       // engine option feat_bg_wgs
    if (a.background && bg_total > 0) cap = bg_total;          // tuning knob: background workgroups in total
    if (a.background > 1) cap = a.background;                  // the caller names the workgroup count (early share beside the forward)
    if (blocks > cap) blocks = cap;
    void* args[] = {(void*)&a, (void*)&packed};
    if (pd) {
        // engine option feat_pf = 2: background launches keep two samples in flight per wave (default: one).  Measured on C3,
        // same box: 1.554 -> 1.533 ms/step; three in flight: 1.558
        const int pf = a.bg_pf > 0 ? a.bg_pf : 1;
        const void* fn = a.planes ? (tight ? (const void*)featurize_reg_kernel<NFL, NDL, true, 1, true, true> : (const void*)featurize_reg_kernel<NFL, NDL, true, 1, true>)
                                  : (const void*)featurize_reg_kernel<NFL, NDL, true, 1, false>;
        if (a.background && pf == 2)
            fn = a.planes ? (const void*)featurize_reg_kernel<NFL, NDL, true, 2, true> : (const void*)featurize_reg_kernel<NFL, NDL, true, 2, false>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(256), args, lds, st);
    } else {
        const void* fn = a.planes ? (const void*)featurize_reg_kernel<NFL, 1, false, 1, true> : (const void*)featurize_reg_kernel<NFL, 1, false, 1, false>;
        (void)hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(256), args, lds, st);
    }
}

template <int NFL>
static void launch_feat_nd(const FeatArgs& a, const unsigned* packed, hipStream_t st) {
    const int nd = (a.D + 63) / 64;
    if (a.pairs == nullptr || nd <= 8) launch_feat_reg<NFL, 8>(a, packed, st);
    else if (nd <= 16) launch_feat_reg<NFL, 16>(a, packed, st);
    else if (nd <= 28) launch_feat_reg<NFL, 28>(a, packed, st);
    else launch_feat_reg<NFL, 32>(a, packed, st);
}

// the register-resident featurizer covers rows of <= 2048 padded features and (pairwise distances) <= 682 atoms with
// 16-bit atom indices; everything else takes the LDS-stash fallback, which has no split-f16 (PLANES) output
bool isk_featurize_reg_ok(const FeatArgs& a) {
    const bool pd = a.pairs != nullptr;
    const int need = (a.Fp + 63) / 64;
    return need <= 32 && (!pd || (a.packed_pairs != nullptr && a.D <= 2048 && (size_t)4 * (a.D / 3) * 16 <= 150 * 1024));
}

// diagnostic (engine option debug_feat_copy): the background launch replaced by a plain stream of the same bytes (coordinates in,
// rows out, no LDS, no arithmetic): separates what the training kernels lose to the featurizer's MEMORY traffic from what
// they lose to its LDS / issue-slot use
__global__ __launch_bounds__(256) void debug_stream_kernel(const f32x4* __restrict__ in, long long nin, f32x4* __restrict__ out, long long nout) {
    const long long stride = (long long)gridDim.x * 256;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nin; i += stride) acc += __builtin_nontemporal_load(in + i);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nout; i += stride) __builtin_nontemporal_store(acc, out + i);
}

void isk_launch_featurize(const FeatArgs& a, hipStream_t st) {
    if (a.S <= 0) return;
    const int dbg_copy = a.dbg_copy;
    if (dbg_copy && a.background) {
        hipLaunchKernelGGL(debug_stream_kernel, dim3(256 * dbg_copy), dim3(256), 0, st, (const f32x4*)a.in, a.S * a.D / 4, (f32x4*)a.xhat, a.S * a.Fp / 4);
        return;
    }
    const bool pd = a.pairs != nullptr;
    const int need = (a.Fp + 63) / 64;
    if (isk_featurize_reg_ok(a)) {
        if (need <= 2) launch_feat_nd<2>(a, a.packed_pairs, st);
        else if (need <= 4) launch_feat_nd<4>(a, a.packed_pairs, st);
        else if (need <= 8) launch_feat_nd<8>(a, a.packed_pairs, st);
        else if (need <= 16) launch_feat_nd<16>(a, a.packed_pairs, st);
        else if (need <= 24) launch_feat_nd<24>(a, a.packed_pairs, st);
        else if (need <= 28) launch_feat_nd<28>(a, a.packed_pairs, st);
        else launch_feat_nd<32>(a, a.packed_pairs, st);
        return;
    }
    // generic fallback (any F, any atom count): LDS stash variant
    const int Dp = pd ? ((a.D + 3) & ~3) : 0;
    const size_t lds = (size_t)4 * (Dp + a.Fp) * sizeof(float);
    long long blocks = (a.S + 3) / 4;
    const long long cap = (long long)ISK_NUM_CU * 8;
    if (blocks > cap) blocks = cap;
    if (lds > 64 * 1024) {
        (void)hipFuncSetAttribute((const void*)featurize_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)featurize_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (pd) hipLaunchKernelGGL(featurize_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(featurize_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, st, a);
}

// =====================================================================================
// fold: theta (Flux order) -> fragment-linear weights, LayerNorm affine folded into layer 1:
//   W1g[o][k] = W1[o][k]*g[k],   bias1[o] = b1[o] + sum_k W1[o][k]*beta[k]
// so that  W1*(g.*xhat + beta) + b1 == W1g*xhat + bias1  and the forward kernel consumes the
// standardised rows directly.
// =====================================================================================
#define FOLD_NB 2048             /* output columns per pass of the transpose tile (16 x 2049 floats = 128 KiB of LDS) */
__global__ __launch_bounds__(256) void fold_kernel(FoldArgs a) {
    ISK_LATENCY_CHAIN_PRIO();
    extern __shared__ __attribute__((aligned(16))) float tile[];     // [16][Np + 1]
    const int layer = blockIdx.y;
    if (layer >= a.nlayers) return;
    if (a.bt && blockIdx.x == 0 && layer == 0 && threadIdx.x == 0) {      // layered path: the optimizer step before this fold is done
        a.bt[0] *= a.beta1;
        a.bt[1] *= a.beta2;
    }
    const int K = a.widths[layer], N = a.widths[layer + 1];
    if (a.native && layer == 0) {
        // fused path: theta keeps W1 fragment-linear (zero padded) and the LayerNorm blocks padded to Kp, so the folded
        // image is an elementwise product and the folded bias a strided dot product
        const int Kp0 = a.Kp[0], Np0 = (N + 15) & ~15, nt0 = Np0 / 16, nch0 = Kp0 / 16;
        const float* W = a.theta + a.off_W[0];
        const bool ln0 = a.off_lns >= 0;
        const int t0 = threadIdx.x;
        if ((int)blockIdx.x < nch0) {
            const int c = blockIdx.x;
            float m = 0.f;
            for (int idx = t0; idx < nt0 * 256; idx += 256) {
                const int j = idx & 3, ln_ = (idx >> 2) & 63;
                const long long e = (long long)c * nt0 * 256 + idx;
                const float w = W[e] * (ln0 ? a.theta[a.off_lns + c * 16 + 4 * (ln_ >> 4) + j] : 1.f);
                a.Wt[0][e] = w;
                m = fmaxf(m, fabsf(w));
            }
            if (a.wmax) {                                   // tensor maximum for the f16 plane scale: one atomic per workgroup
                if (c == 0 && t0 == 0) a.wmax_clear[0] = 0u;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, WAVE));
                if ((t0 & 63) == 0) tile[t0 >> 6] = m;
                __syncthreads();
                if (t0 == 0) atomicMax(a.wmax, __builtin_bit_cast(unsigned, fmaxf(fmaxf(tile[0], tile[1]), fmaxf(tile[2], tile[3]))));
            }
            return;
        }
        const int o = (int)blockIdx.x - nch0;
        if (o >= Np0) return;
        float sacc = 0.f;
        if (o < N && ln0) {
            const float* Wo = W + ((long long)(o >> 4) * 64 + (o & 15)) * 4;          // + ((c*nt)*64 + 16 kq)*4 + j
            for (int k = t0; k < K; k += 256)
                sacc += Wo[((long long)(k >> 4) * nt0 * 64 + ((k & 15) >> 2) * 16) * 4 + (k & 3)] * a.theta[a.off_lnb + k];
            sacc = wave_sum(sacc);
        }
        if ((t0 & 63) == 0) tile[t0 >> 6] = sacc;
        __syncthreads();
        if (t0 == 0) a.bias[0][o] = (o < N) ? a.theta[a.off_b[0] + o] + ((tile[0] + tile[1]) + (tile[2] + tile[3])) : 0.f;
        return;
    }
    const int Kp = a.Kp[layer], Np = (N + 15) & ~15;
    const int nt = Np / 16, nchunks = Kp / 16;
    const float* W = a.theta + a.off_W[layer];
    const float* b = a.theta + a.off_b[layer];
    const bool ln = (layer == 0 && a.off_lns >= 0);
    const float* g = ln ? a.theta + a.off_lns : nullptr;
    const float* beta = ln ? a.theta + a.off_lnb : nullptr;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < nchunks) {
        // one 16-feature chunk: coalesced read of W[:, 16c .. 16c+15] (Flux layout: o fastest),
        // LDS transpose, coalesced write of the chunk's nt fragment-linear KiB
        // (output columns in blocks of FOLD_NB: the transpose tile is [16][block + 1] floats -- a 4096-wide layer in one piece would
        // need 262 KB of LDS, the launch failed with "invalid argument"; up to FOLD_NB columns this is the single pass it always was)
        const int c = blockIdx.x;
        const int ld = (Np < FOLD_NB ? Np : FOLD_NB) + 1;
        const float* Wc = W + (long long)c * 16 * N;
        float* Wt = a.Wt[layer] + (long long)c * nt * 256;
        for (int ob = 0; ob < Np; ob += FOLD_NB) {
            const int nbw = (Np - ob < FOLD_NB) ? Np - ob : FOLD_NB;         // padded columns of this block (a multiple of 16)
            const int nv = (N - ob < nbw) ? N - ob : nbw;                   // valid ones (>= 1: ob <= Np - 16 < N)
            const int nel = 16 * nv;
            for (int e0 = 0; e0 < nel; e0 += 256 * 4) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * 256 + tid;
                    const int kl = e / nv;
                    v[u] = (e < nel && c * 16 + kl < K) ? Wc[(long long)kl * N + ob + (e - kl * nv)] * (ln ? g[c * 16 + kl] : 1.f) : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * 256 + tid;
                    if (e < nel) { const int kl = e / nv; tile[kl * ld + (e - kl * nv)] = v[u]; }
                }
            }
            __syncthreads();
            for (int idx = tid; idx < (nbw / 16) * 256; idx += 256) {
                const int j = idx & 3, lane = (idx >> 2) & 63, t = idx >> 8;
                const int ol = t * 16 + (lane & 15), kl = 4 * (lane >> 4) + j;
                Wt[(ob / 16) * 256 + idx] = (ob + ol < N) ? tile[kl * ld + ol] : 0.f;
            }
            __syncthreads();
        }
        return;
    }
    // bias blocks: one output row per block, its 4 waves split the feature axis (the column-major W1 makes this a gather
    // of one element per 64-B segment: the loads are all issued at once, the chain is one memory round trip)
    const int lane = tid & 63, wv4 = tid >> 6;
    const int o = (int)blockIdx.x - nchunks;
    if (o >= Np) return;
    float sacc = 0.f;
    if (o < N && ln) {
        const int per = ((K + 3) / 4 + 63) / 64 * 64;            // features per wave, multiple of 64
        const int kbeg = wv4 * per, kend = (kbeg + per < K) ? kbeg + per : K;
        for (int k0 = kbeg; k0 < kend; k0 += 64 * 8) {
            float wv[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * 64 + lane;
                const int kc = (k < kend) ? k : (kend - 1);      // clamped (branch-free loads), masked below
                wv[u] = W[o + (long long)kc * N];
                bv[u] = (k < kend) ? beta[kc] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) sacc += wv[u] * bv[u];
        }
        sacc = wave_sum(sacc);
    }
    if (lane == 0) tile[wv4] = sacc;
    __syncthreads();
    if (tid == 0) a.bias[layer][o] = (o < N) ? b[o] + ((tile[0] + tile[1]) + (tile[2] + tile[3])) : 0.f;
}

void isk_launch_fold(const FoldArgs& a, hipStream_t st) {
    int bx = 1, maxNp = 16;
    for (int l = 0; l < a.nlayers; ++l) {
        const int Kp = a.Kp[l], Np = isk_pad16(a.widths[l + 1]);
        bx = bx > (Kp / 16 + Np) ? bx : (Kp / 16 + Np);
        maxNp = maxNp > Np ? maxNp : Np;
    }
    const size_t lds = (size_t)16 * ((maxNp < FOLD_NB ? maxNp : FOLD_NB) + 1) * sizeof(float);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)fold_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fold_kernel, dim3(bx, a.nlayers), dim3(256), lds, st, a);
}

// =====================================================================================
// Fused MLP forward.  One wave owns 16-row "units" of the sample matrix and ALL output
// columns of layer 1; two units are processed per pass so every weight fragment feeds 8
// MFMAs.  Operands go global -> VGPR directly in MFMA fragment shape:
//   A (samples): lane (i = lane&15, q = lane>>4) loads 16 B = X[row i][16c + 4q .. +3]
//   B (weights): lane loads 16 B of the fragment-linear image (coalesced 1 KiB per wave)
// and MFMA step j of chunk c contracts k = 16c + 4q + j on both sides.  The f32 MFMA
// (v_mfma_f32_16x16x4_f32, 256 FLOP/clk/CU) is slow enough that neither operand needs
// LDS: X is private to the wave (read once from HBM), the layer-1 weights (<= 1 MB) stay
// L2-resident.  The narrow tail layers run as MFMA over an LDS transpose buffer
// (accumulator layout -> A-operand layout) inside the same kernel, so only nout floats
// per sample are written.
// Bound: f32 MFMA (2*Fp*H1p flop per sample vs 4*Fp bytes).
// =====================================================================================
template <int NT>
struct WFrags {
    f32x4 w[NT];
};
struct XFrags {
    f32x4 x0, x1;
};

template <int NT>
__device__ __forceinline__ void load_w(WFrags<NT>& f, const f32x4* __restrict__ Wp, int c, int nt_total) {
#pragma unroll
    for (int t = 0; t < NT; ++t) f.w[t] = Wp[(long long)(c * nt_total + t) * 64];
}
template <bool TWO>
__device__ __forceinline__ void load_x(XFrags& f, const float* __restrict__ x0, const float* __restrict__ x1, int c) {
    f.x0 = *(const f32x4*)(x0 + c * 16);
    if (TWO) f.x1 = *(const f32x4*)(x1 + c * 16);
}

template <int NT, bool TWO>
__device__ __forceinline__ void mma_frags(const XFrags& x, const WFrags<NT>& f, f32x4 (&acc0)[NT], f32x4 (&acc1)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x0[j], f.w[t][j], acc0[t], 0, 0, 0);
            if (TWO) acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x1[j], f.w[t][j], acc1[t], 0, 0, 0);
        }
    }
}

// Layer-1 GEMM of one pass.  Two prefetch distances: the sample fragments come from HBM and are
// fetched XPF chunks ahead (8 VGPRs per chunk in flight), the weight fragments are L2-resident and
// double-buffered (one chunk ahead, 4*NT VGPRs each).  The sched_barriers pin "issue the loads,
// THEN run this chunk's MFMAs": without them hipcc sinks the loads below the MFMA block (to save
// registers) and waits vmcnt(0) right after issuing them, exposing the memory latency every chunk.
template <int NT, bool TWO>
__device__ __forceinline__ void layer1_gemm(const MlpFwdArgs& a, const float* x0, const float* x1, int lane,
                                            f32x4 (&acc0)[NT], f32x4 (&acc1)[NT]) {
    constexpr int XPF = 5;                        // sample fragments: 5 chunks ahead, ring of 6
    constexpr int WPF = 2;                        // weight fragments: 2 chunks ahead, ring of 3
    constexpr int UNR = 6;                        // lcm(XPF + 1, WPF + 1): ring slots are compile-time
    const int ntt = a.dense_mode ? a.w_nt_total : NT;
    const f32x4* Wp = (const f32x4*)a.W1t + (long long)(a.dense_mode ? a.w_tile0 : 0) * 64 + lane;
    const int nch = a.nchunks;
    const int last = nch - 1;
    XFrags xr[XPF + 1];
    WFrags<NT> w[WPF + 1];
#pragma unroll
    for (int i = 0; i < XPF; ++i) load_x<TWO>(xr[i], x0, x1, i < last ? i : last);
#pragma unroll
    for (int i = 0; i < WPF; ++i) load_w<NT>(w[i], Wp, i < last ? i : last, ntt);
    for (int c = 0; c < nch; c += UNR) {
#pragma unroll
        for (int s = 0; s < UNR; ++s) {
            const int cc = c + s;
            load_x<TWO>(xr[(s + XPF) % (XPF + 1)], x0, x1, cc + XPF < last ? cc + XPF : last);
            load_w<NT>(w[(s + WPF) % (WPF + 1)], Wp, cc + WPF < last ? cc + WPF : last, ntt);
            __builtin_amdgcn_sched_barrier(0);
            if (cc < nch) mma_frags<NT, TWO>(xr[s % (XPF + 1)], w[s % (WPF + 1)], acc0, acc1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int NT>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(MlpFwdArgs a, long long units, long long units_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    float* scrA = smem + a.tailw_floats + (size_t)wave * 16 * (a.stride_a + a.stride_b);
    float* scrB = scrA + 16 * a.stride_a;
    // stage the (small) fragment-linear tail weights in LDS once: the per-unit tail then never waits on L2
    stage_tail_weights(a, smem);
    __syncthreads();
    __builtin_amdgcn_s_setprio(3);                 // MFMA waves win issue arbitration over co-resident streaming kernels
    const long long gw = (long long)blockIdx.x * 4 + wave;
    long long u = gw * units_per_wave;
    long long uend = u + units_per_wave;
    if (uend > units) uend = units;
    for (; u < uend; u += 2) {
        const bool two = (u + 1 < uend);            // wave-uniform
        long long r0 = u * 16 + li, r1 = r0 + 16;
        if (r0 > a.S - 1) r0 = a.S - 1;
        if (r1 > a.S - 1) r1 = a.S - 1;
        if (a.rows) { r0 = a.rows[r0]; r1 = a.rows[r1]; }
        const float* x0 = a.X + r0 * (long long)a.Fp + lq * 4;
        const float* x1 = a.X + r1 * (long long)a.Fp + lq * 4;
        f32x4 acc0[NT], acc1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { acc0[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        if (two) layer1_gemm<NT, true>(a, x0, x1, lane, acc0, acc1);
        else layer1_gemm<NT, false>(a, x0, x1, lane, acc0, acc1);
        if (a.dense_mode) {                                             // wave-uniform
            unit_tail<NT, true>(a, acc0, u * 16, scrA, scrB, smem, lane);
            if (two) unit_tail<NT, true>(a, acc1, (u + 1) * 16, scrA, scrB, smem, lane);
        } else {
            unit_tail<NT, false>(a, acc0, u * 16, scrA, scrB, smem, lane);
            if (two) unit_tail<NT, false>(a, acc1, (u + 1) * 16, scrA, scrB, smem, lane);
        }
    }
}

template <int NT>
static void launch_fwd_nt(const MlpFwdArgs& a, hipStream_t st) {
    const long long units = (a.S + 15) / 16;
    const int wps = a.waves_per_simd < 1 ? 1 : (a.waves_per_simd > 4 ? 4 : a.waves_per_simd);     // engine option fwd_waves_per_simd
    const long long waves_max = (long long)ISK_NUM_CU * 4 * wps;
    long long upw = (units + waves_max - 1) / waves_max;     // units per wave
    if (upw < 1) upw = 1;
    const long long nwaves = (units + upw - 1) / upw;
    const long long blocks = (nwaves + 3) / 4;
    const size_t lds = ((size_t)a.tailw_floats + (size_t)4 * 16 * (a.stride_a + a.stride_b)) * sizeof(float);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)mlp_fwd_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mlp_fwd_kernel<NT>, dim3((unsigned)blocks), dim3(256), lds, st, a, units, upw);
}

int isk_launch_mlp_fwd(const MlpFwdArgs& a, hipStream_t st) {
    if (a.S <= 0) return 0;
    switch (a.nt1) {
        case 1: launch_fwd_nt<1>(a, st); break;
        case 2: launch_fwd_nt<2>(a, st); break;
        case 3: launch_fwd_nt<3>(a, st); break;
        case 4: launch_fwd_nt<4>(a, st); break;
        case 5: launch_fwd_nt<5>(a, st); break;
        case 6: launch_fwd_nt<6>(a, st); break;
        case 7: launch_fwd_nt<7>(a, st); break;
        case 8: launch_fwd_nt<8>(a, st); break;
        case 9: launch_fwd_nt<9>(a, st); break;
        case 10: launch_fwd_nt<10>(a, st); break;
        default: return ISK_EUNSUPPORTED;
    }
    return 0;
}

// =====================================================================================
// K-mean: ks[n][o] = (sum_k out[n*K + k][o]) / K      (expectation, src/isotarget.jl:18)
// One wave per start point; lanes stride over the K*nout contiguous outputs of that
// start, then a shuffle reduction combines lanes that hold the same output index o.
// =====================================================================================
__global__ __launch_bounds__(256) void kmean_kernel(KmeanArgs a) {
    ISK_LATENCY_CHAIN_PRIO();
    const long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (a.zero2 && blockIdx.x == 0 && threadIdx.x == 0) { a.zero2[0] = 0.0; a.zero2[1] = 0.0; a.zero2[3] = 0.0; }   // [2] is the status word
    if (n >= a.N) return;
    const float* base = a.out_ys + n * (long long)a.K * a.nout;
    const float* w = a.w ? a.w + n * (long long)a.K : nullptr;
    const int tot = a.K * a.nout;
    // The sums run in DOUBLE and are rounded to float once: K chi is what the ISA target amplifies (its simplex inverse has a condition
    // number of a few hundred on villin-like data), and a float summation of K = 20 terms carries more rounding error than the
    // K-mean leaves of the forward's own (averaged-out) error.
    if (a.nout <= 64 && (64 % a.nout) == 0) {
        // lane l always sees output index l % nout (stride 64 is a multiple of nout)
        double s = 0.0;
        for (int i = lane; i < tot; i += WAVE) s += w ? (double)base[i] * (double)w[i / a.nout] : (double)base[i];
        for (int o = 32; o >= a.nout; o >>= 1) s += __shfl_xor(s, o, WAVE);
        if (lane < a.nout) {
            const double m = s / (double)a.K;
            a.ks[n * a.nout + lane] = (float)m;
            if (a.ks64) a.ks64[n * a.nout + lane] = m;
        }
    } else if (a.nout < 64) {
        // nout does not divide 64 (C3: nout = 3): the first W = 64 - 64 % nout lanes walk the K x nout block with stride W,
        // so a lane keeps one output index; the W / nout partial sums per output are combined through LDS
        __shared__ double part[4][64];
        const int wv = threadIdx.x >> 6;
        const int W = WAVE - WAVE % a.nout;
        double s = 0.0;
        if (lane < W)
            for (int i = lane; i < tot; i += W) s += w ? (double)base[i] * (double)w[i / a.nout] : (double)base[i];
        part[wv][lane] = s;
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the wave's own LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        if (lane < a.nout) {
            double t = 0.0;
            for (int j = lane; j < W; j += a.nout) t += part[wv][j];
            const double m = t / (double)a.K;
            a.ks[n * a.nout + lane] = (float)m;
            if (a.ks64) a.ks64[n * a.nout + lane] = m;
        }
    } else {
        for (int o = lane; o < a.nout; o += WAVE) {
            double s = 0.0;
            for (int k = 0; k < a.K; ++k) s += w ? (double)base[k * a.nout + o] * (double)w[k] : (double)base[k * a.nout + o];
            const double m = s / (double)a.K;
            a.ks[n * a.nout + o] = (float)m;
            if (a.ks64) a.ks64[n * a.nout + o] = m;
        }
    }
}

void isk_launch_kmean(const KmeanArgs& a, hipStream_t st) {
    if (a.N <= 0) return;
    hipLaunchKernelGGL(kmean_kernel, dim3((unsigned)((a.N + 3) / 4)), dim3(256), 0, st, a);
}

// min / max over n values: single workgroup, wavefront shuffles + one LDS exchange
__global__ __launch_bounds__(1024) void minmax_kernel(const float* v, long long n, float* mm) {
    ISK_LATENCY_CHAIN_PRIO();
    __shared__ float smin[16], smax[16];
    float lo = INFINITY, hi = -INFINITY;
    bool nan = false;
    // eight values per thread and pass, requested together from clamped addresses (a duplicate changes neither min nor max): one
    // value per pass was one memory round trip per 1024 values -- 12 of them, ~10 us, at the C5 shard's 12 500 start points
    for (long long i = threadIdx.x; i < n; i += 8 * 1024) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const long long j = i + u * 1024; x[u] = v[j < n ? j : n - 1]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            nan |= (x[u] != x[u]);
            lo = fminf(lo, x[u]);
            hi = fmaxf(hi, x[u]);
        }
    }
    if (nan) { lo = NAN; hi = NAN; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float l2 = __shfl_xor(lo, o, WAVE), h2 = __shfl_xor(hi, o, WAVE);
        lo = (lo != lo || l2 != l2) ? NAN : fminf(lo, l2);
        hi = (hi != hi || h2 != h2) ? NAN : fmaxf(hi, h2);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smin[w] = lo; smax[w] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) {
            lo = (lo != lo || smin[i] != smin[i]) ? NAN : fminf(lo, smin[i]);
            hi = (hi != hi || smax[i] != smax[i]) ? NAN : fmaxf(hi, smax[i]);
        }
        mm[0] = lo;
        mm[1] = hi;
    }
}

void isk_launch_minmax(const float* v, long long n, float* mm, hipStream_t st) {
    hipLaunchKernelGGL(minmax_kernel, dim3(1), dim3(1024), 0, st, v, n, mm);
}

__global__ __launch_bounds__(256) void shiftscale_kernel(const float* ks, const float* mm, float* target, long long n,
                                                         int* status) {
    ISK_LATENCY_CHAIN_PRIO();
    const float lo = mm[0], hi = mm[1];
    const bool ok = hi > lo;                       // false for NaN too (src/isotarget.jl:39)
    if (!ok && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, 1);
    const float inv = hi - lo;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        target[i] = (ks[i] - lo) / inv;
}

void isk_launch_shiftscale(const float* ks, const float* mm, float* target, long long n, int* status, hipStream_t st) {
    long long b = (n + 255) / 256;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(shiftscale_kernel, dim3((unsigned)b), dim3(256), 0, st, ks, mm, target, n, status);
}

// target = A' ks in Float64, rounded once to Float32 (src/isotarget.jl:104-106), like isa_target_kernel on the device path
__global__ __launch_bounds__(256) void isa_apply_kernel(const float* ks, const double* At, const int* rowperm, float* target,
                                                        long long N, int d) {
    ISK_LATENCY_CHAIN_PRIO();
    const long long tot = N * d;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long long)gridDim.x * 256) {
        const long long n = e / d;
        const int i = (int)(e % d);
        const int a = rowperm[i];
        double s = 0.0;
        for (int b = 0; b < d; ++b) s += At[a * d + b] * (double)ks[n * d + b];
        target[e] = (float)s;
    }
}

void isk_launch_isa_apply(const float* ks, const double* At, const int* rowperm, float* target, long long N, int d,
                          hipStream_t st) {
    long long b = (N * d + 255) / 256;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(isa_apply_kernel, dim3((unsigned)b), dim3(256), 0, st, ks, At, rowperm, target, N, d);
}

