// One training epoch of train_batch! (src/iso.jl:179-194 of the reference) as ONE persistent launch.
//
// Why.  A step of the four-kernel chain (train_fwd1n -> train_tail -> train_grad1n -> opt_step) costs 31-45 us on an idle chip for
// 8 us of matrix work: every launch has a floor of about 5 us (dispatch, first memory round trip, end-of-kernel write-back) and the
// kernels re-read what the previous step already had on the chip (the layer-1 weights and their optimizer state: 4 MB per step).
// Here a workgroup OWNS a block of the layer-1 weights for the whole epoch -- K slice ks (CPS chunks of 16 features) x output tile
// tg -- together with their Adam / Nesterov state, in LDS.  For that block the products of BOTH directions are local:
//     forward   z1[rows][tile tg]  +=  (g .* xhat[rows][slice] + beta) W1[slice][tile]'      (a K-slice partial, all batch rows)
//     backward  G[slice][tile]      =   xhat[rows][slice]' delta1[rows][tile]                 (complete: all batch rows)
// so the optimizer step and the next step's forward follow the backward with no communication at all.  What does cross workgroups
// is the transposition of ownership every training step has: K-slice partials of z1 -> the units' tails (seam A), delta1 -> the
// block owners (seam B), plus two small exchanges (LayerNorm-gradient partials between the nt owners of a K slice; the tail
// parameters, updated by slice owners, back to the tails: seam C).  Seams are in-launch hand-offs in the write-through form:
// payload stored sc1 by every wave, s_waitcnt vmcnt(0), workgroup barrier, ONE relaxed agent-scope add on a sharded monotonic
// counter; the consumer's wave 0 polls the shards with sc1 loads, and EVERY load of handed-off bytes is an sc1 load (no acquire
// fence, no L2 write-back).  Counters are zeroed by a memset node in front of the launch; every spin is bounded (ISK_EP_TIMEOUT
// ticks of the 100 MHz clock) and raises an abort word that ends all workgroups.
//
// Grid: G = KS x nt workgroups of 512 threads (<= the CU count: they must be co-resident), <= 128 registers so that a workgroup
// fits beside a wave of the side-stream featurizer on every SIMD.  Workgroups u < U (= ceil(B/16)) also run the tail of unit u
// (train_tailn.hpp, eight waves) and own a slice of the tail parameters; workgroups with ks = 0 own the layer-1 bias of their tile.
// Results: those of the four-kernel chain up to summation order (27 K-slice partials instead of 8, rows split over 8 waves).
#include "isk_internal.h"
#ifndef ISK_EXPERIMENTAL
// shipped library: the persistent epoch is not built (it measured slower); option train_persist does not exist there
bool isk_train_epoch_plan(const TrainShape&, int, int, int, EpochPlan*) { return false; }
int isk_launch_train_epoch(const TrainShape&, const EpochPlan&, EpochArgs, hipStream_t) { return -1; }
#else
#include "train_tailn.hpp"

#define EP_THREADS 512
#define EP_WAVES 8
#define ISK_EP_TIMEOUT 20000000ull          /* 0.2 s of s_memrealtime (100 MHz) */
#define EP_RLX __ATOMIC_RELAXED
#define EP_AGENT __HIP_MEMORY_SCOPE_AGENT

typedef unsigned ep_u32;

// counter block (zeroed before every launch): every counter word on a 128-byte line of its own (the adders of one line serialise,
// and a poll of a line under atomic traffic waits behind it).  ids: [0..8) seam A shards, [8..16) seam B, [16..24) seam C, 24 abort,
// 25 + ks: the tile owners of K slice ks
#define EP_LINE 32
#define EP_CNT_A 0
#define EP_CNT_B 8
#define EP_CNT_C 16
#define EP_CNT_ABORT 24
#define EP_CNT_SIB 25
#define EP_MAX_KS ((ISK_EP_CNT_WORDS / EP_LINE) - EP_CNT_SIB)
#define EP_UPW 8                             /* units per wave: batches of up to 16 * 8 * 8 = 1024 rows */

// wave 0 of the workgroup: lanes 0..7 poll the eight shards of up to two counters until both sums reach their targets
__device__ __forceinline__ bool ep_poll(ep_u32* cnt, int base0, ep_u32 want0, int base1, ep_u32 want1, int ln) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        ep_u32 v0 = 0, v1 = 0, ab = 0;
        if (ln < 8) {
            v0 = __hip_atomic_load(cnt + (base0 + ln) * EP_LINE, EP_RLX, EP_AGENT);
            if (base1 >= 0) v1 = __hip_atomic_load(cnt + (base1 + ln) * EP_LINE, EP_RLX, EP_AGENT);
        } else if (ln == 8) {
            ab = __hip_atomic_load(cnt + EP_CNT_ABORT * EP_LINE, EP_RLX, EP_AGENT);
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { v0 += __shfl_xor(v0, o, 64); v1 += __shfl_xor(v1, o, 64); ab += __shfl_xor(ab, o, 64); }
        v0 = __builtin_amdgcn_readfirstlane(v0); v1 = __builtin_amdgcn_readfirstlane(v1); ab = __builtin_amdgcn_readfirstlane(ab);
        if (ab) return false;
        if (v0 >= want0 && (base1 < 0 || v1 >= want1)) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > ISK_EP_TIMEOUT) {
            if (ln == 0) __hip_atomic_store(cnt + EP_CNT_ABORT * EP_LINE, 1u, EP_RLX, EP_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// one counter word (the sibling counters)
__device__ __forceinline__ bool ep_poll1(ep_u32* cnt, int id, ep_u32 want, int ln) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        ep_u32 v = 0;
        if (ln == 0) v = __hip_atomic_load(cnt + id * EP_LINE, EP_RLX, EP_AGENT);
        if (ln == 1) v = __hip_atomic_load(cnt + EP_CNT_ABORT * EP_LINE, EP_RLX, EP_AGENT);
        const ep_u32 ab = __builtin_amdgcn_readlane(v, 1);
        v = __builtin_amdgcn_readfirstlane(v);
        if (ab) return false;
        if (v >= want) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > ISK_EP_TIMEOUT) {
            if (ln == 0) __hip_atomic_store(cnt + EP_CNT_ABORT * EP_LINE, 1u, EP_RLX, EP_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// every wave has drained its write-through stores; one lane signals
#define EP_PUBLISH(id)                                                          \
    do {                                                                        \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        \
        __syncthreads();                                                        \
        if (tid == 0) __hip_atomic_fetch_add(a.cnt + (id) * EP_LINE, 1u, EP_RLX, EP_AGENT); \
    } while (0)
// wave 0 polls, everybody learns the outcome; a failed wait ends the workgroup (the abort word ends the others)
#define EP_WAIT(expr)                                                           \
    do {                                                                        \
        if (wave == 0) { const bool ok_ = (expr); if (ln == 0) s_ok = ok_ ? 1 : 0; } \
        __syncthreads();                                                        \
        if (!s_ok) { aborted = true; goto done; }                               \
    } while (0)

__device__ __forceinline__ float ep_ld(const float* p) {           // sc1 dword load of handed-off data
    return __builtin_bit_cast(float, __hip_atomic_load((const ep_u32*)p, EP_RLX, EP_AGENT));
}
__device__ __forceinline__ void ep_st(float* p, float v) {
    __hip_atomic_store((ep_u32*)p, __builtin_bit_cast(ep_u32, v), EP_RLX, EP_AGENT);
}

struct OptRule {
    int kind; float lr, reg, beta1, beta2, eps, rho, bc1, bc2;
    __device__ __forceinline__ void operator()(float& th, float gs, float& mi, float& vi) const {       // opt_step_kernel's update
        const float g = gs + reg * th;
        float d;
        if (kind == ISK_OPT_ADAM) {
            const float m = beta1 * mi + (1.f - beta1) * g;
            const float v = beta2 * vi + (1.f - beta2) * g * g;
            mi = m; vi = v;
            d = m / bc1 / (sqrtf(v / bc2) + eps) * lr;
        } else {
            d = -rho * rho * mi + (1.f + rho) * lr * g;
            mi = rho * mi - lr * g;
        }
        th -= d;
    }
};

template <int L, int CPS>
__global__ __launch_bounds__(EP_THREADS, 4) void train_epoch_kernel(EpochArgs a) {
    ISK_LATENCY_CHAIN_PRIO();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ int s_ok;
    __shared__ float s_lstep;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63, c = ln & 15, q = ln >> 4;
    const int w = blockIdx.x;
    const int nt = a.nt, H1p = nt * 16, H1 = a.d.H1;
    const int tg = w % nt, ks = w / nt;
    const int c0 = ks * CPS;                                       // first 16-feature chunk of this block
    const int U = a.U, B = a.B;
    const int nTail = U < a.G ? U : a.G;                            // workgroups w < nTail run the tails of units w, w + G, ...
    const bool tail_role = w < nTail, b1_role = ks == 0, slice_role = w < a.nS;
    const bool ln_on = a.ln != 0;
    bool aborted = false;
    // ---- LDS carve
    float* s_W = sm;                          // [CPS][256]  fragment-linear layer-1 weights of this block (master copy)
    float* s_mW = s_W + CPS * 256;
    float* s_vW = s_mW + CPS * 256;
    float* s_lnp = s_vW + CPS * 256;          // [6][CPS*16]: g, beta, m_g, v_g, m_beta, v_beta
    float* s_own = s_lnp + 6 * CPS * 16;      // [2][CPS*16]: this tile's shares of dg, dbeta
    float* s_D = s_own + 2 * CPS * 16;        // [16]
    float* s_zb = s_D + 16;                   // [16]  W1 beta over this K slice (the LayerNorm bias, folded out of the sample operand)
    float* s_Dred = s_zb + 16;                // [8][16]
    float* s_red = s_Dred + EP_WAVES * 16;    // [8][CPS][256]; [0] doubles as the summed G
    float* s_z1 = s_red + EP_WAVES * CPS * 256;          // [16][H1p]  tail: z1 image
    float* s_zred = s_z1 + 16 * H1p;                     // [8][256]
    float* s_tile = s_zred + EP_WAVES * 256;             // [8][16*TN_TS]
    float* s_wc = s_tile + EP_WAVES * 16 * TN_TS;        // [nsmall] tail role: copy of theta[off_b0 .. nparams)
    const int nsmall = (int)(a.nparams - a.off_b0);
    // ---- this block's parameters and optimizer state -> LDS (chunks past the end of the feature axis: zeros)
    for (int e = tid; e < CPS * 256; e += EP_THREADS) {
        const int cc = e >> 8;
        const bool ok = c0 + cc < a.nch;
        const long long gi = a.off_W0 + ((long long)(c0 + cc) * nt + tg) * 256 + (e & 255);
        s_W[e] = ok ? a.theta[gi] : 0.f;
        s_mW[e] = ok ? a.m[gi] : 0.f;
        s_vW[e] = ok ? a.v[gi] : 0.f;
    }
    for (int e = tid; e < CPS * 16; e += EP_THREADS) {
        const int k = c0 * 16 + e;
        const bool ok = ln_on && k < a.nch * 16;
        s_lnp[0 * CPS * 16 + e] = ok ? a.theta[a.off_lns + k] : 1.f;
        s_lnp[1 * CPS * 16 + e] = ok ? a.theta[a.off_lnb + k] : 0.f;
        s_lnp[2 * CPS * 16 + e] = ok ? a.m[a.off_lns + k] : 0.f;
        s_lnp[3 * CPS * 16 + e] = ok ? a.v[a.off_lns + k] : 0.f;
        s_lnp[4 * CPS * 16 + e] = ok ? a.m[a.off_lnb + k] : 0.f;
        s_lnp[5 * CPS * 16 + e] = ok ? a.v[a.off_lnb + k] : 0.f;
    }
    __syncthreads();
    OptRule rule;
    rule.kind = a.opt_kind; rule.lr = a.lr; rule.reg = a.reg; rule.beta1 = a.beta1; rule.beta2 = a.beta2; rule.eps = a.eps; rule.rho = a.rho;
    double bt1 = a.bt1, bt2 = a.bt2;
    double loss_sum = 0.0;
    bool collapsed = a.collapsed_in != 0;
    const __amdgpu_buffer_rsrc_t rzp = tn_rsrc(a.zp, 0x7fffffff);
    const __amdgpu_buffer_rsrc_t rd1 = tn_rsrc(a.delta1, 0x7fffffff);
    const int zslice = a.zrows * H1p * 4;                          // bytes between two K-slice partials
    // the slice of the tail parameters this workgroup updates (slice role): elements [slo, shi) of the partial vector
    const int PS = (int)a.d.PS, sbase = 1 + H1, stot = PS - sbase;
    const int slo = slice_role ? sbase + (int)((long long)stot * w / a.nS) : 0, shi = slice_role ? sbase + (int)((long long)stot * (w + 1) / a.nS) : 0;
    // this wave's units: u = wave + 8 i, i < nuw; the batch rows of the units' samples (lane c of unit i) for the coming step
    const int nuw = wave < U ? (U - wave + EP_WAVES - 1) / EP_WAVES : 0;
    int idxU[EP_UPW];
    auto load_idx = [&](int s2) {
        const int* pm = a.perm + (long long)s2 * B;
#pragma unroll
        for (int i = 0; i < EP_UPW; ++i) {
            const int r = 16 * (wave + EP_WAVES * i) + c;
            idxU[i] = pm[r < B ? r : B - 1];
        }
    };
    load_idx(0);
    const float* const Xq = a.X + 4 * q;
    const float* const Xc = a.X + c;
    int chk[CPS];                                                  // feature offsets of this block's chunks (clamped: padding chunks have zero weights)
#pragma unroll
    for (int cc = 0; cc < CPS; ++cc) chk[cc] = 16 * (c0 + cc < a.nch ? c0 + cc : a.nch - 1);

#define EP_STAMP(k) do { if (a.dbg && tid == 0 && (w == 0 || w == a.G - 1) && s == (a.nb > 1 ? 1 : 0)) a.dbg[(w == 0 ? 0 : 32) + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    for (int s = 0; s < a.nb; ++s) {
        const int* perm = a.perm + (long long)s * B;
        EP_STAMP(0);
        rule.bc1 = (float)(1.0 - bt1); rule.bc2 = (float)(1.0 - bt2);
        // ================================ P1: K-slice partial of z1' for every unit, tile tg ================================
        // z1 = W1 (g .* xhat + beta) = (W1 .* g) xhat + W1 beta: the scale rides on the weight fragments, the bias term is a vector
        if (tid < 16) {
            float zb = 0.f;
            for (int kl = 0; kl < CPS * 16; ++kl) zb += s_W[(kl >> 4) * 256 + (tid + 16 * ((kl & 15) >> 2)) * 4 + (kl & 3)] * s_lnp[CPS * 16 + kl];
            s_zb[tid] = zb;
        }
        __syncthreads();
        {
            f32x4 Wg[CPS];
#pragma unroll
            for (int cc = 0; cc < CPS; ++cc) Wg[cc] = *(const f32x4*)(s_W + cc * 256 + ln * 4) * *(const f32x4*)(s_lnp + cc * 16 + 4 * q);
            const f32x4 zbv = *(const f32x4*)(s_zb + 4 * q);
#pragma unroll
            for (int i0 = 0; i0 < EP_UPW; i0 += 4) {
                if (i0 < nuw) {                                     // wave-uniform
                    f32x4 xa[4][CPS];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float* x = Xq + (long long)idxU[i0 + j] * a.Fp;
#pragma unroll
                        for (int cc = 0; cc < CPS; ++cc) xa[j][cc] = *(const f32x4*)(x + chk[cc]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (i0 + j < nuw) {
                            f32x4 acc0 = zbv, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int cc = 0; cc < CPS; ++cc) {
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj) {
                                    if (jj & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Wg[cc][jj], xa[j][cc][jj], acc1, 0, 0, 0);
                                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Wg[cc][jj], xa[j][cc][jj], acc0, 0, 0, 0);
                                }
                            }
                            const f32x4 z = acc0 + acc1;           // RL: lane (sample c, q) holds z1[sample][16 tg + 4q + r]
                            const int u = wave + EP_WAVES * (i0 + j);
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tn_u32x4, z), rzp, ((16 * u + c) * H1p + 16 * tg + 4 * q) * 4,
                                                                   ks * zslice, 16);
                        }
                    }
                }
            }
        }
        EP_STAMP(1);
        EP_PUBLISH(EP_CNT_A + (w & 7));
        EP_STAMP(2);
        // ================================ P2: the tails (workgroups w < nTail) ================================
        if (tail_role) {
            // the tail parameters of this step (published by their owners behind the previous step): one coalesced copy into LDS
            if (s > 0) EP_WAIT(ep_poll(a.cnt, EP_CNT_C, (ep_u32)a.nC * s, -1, 0, ln));
            for (int e = tid; e < nsmall; e += EP_THREADS) s_wc[e] = ep_ld(a.theta + a.off_b0 + e);
            EP_WAIT(ep_poll(a.cnt, EP_CNT_A, (ep_u32)a.G * (s + 1), -1, 0, ln));
            EP_STAMP(3);
            for (int u = w; u < U; u += a.G) {
                // z1 of the unit = sum of the KS partials (slice order), by all threads, into the LDS image
                const int nq = H1p >> 2;
                for (int i4 = tid; i4 < 16 * nq; i4 += EP_THREADS) {
                    const int row = i4 / nq, col4 = i4 - row * nq;
                    const int off = ((16 * u + row) * H1p + 4 * col4) * 4;
                    // sixteen partials in flight per round trip; summed in double: exact to float rounding whatever their number
                    double z0 = 0.0, z1d = 0.0, z2 = 0.0, z3 = 0.0;
                    for (int k2 = 0; k2 < a.KS; k2 += 16) {
                        f32x4 pz[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            pz[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rzp, off, (k2 + i < a.KS ? k2 + i : 0) * zslice, 16));
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            if (k2 + i < a.KS) { z0 += (double)pz[i][0]; z1d += (double)pz[i][1]; z2 += (double)pz[i][2]; z3 += (double)pz[i][3]; }
                    }
                    *(f32x4*)(s_z1 + row * H1p + 4 * col4) = (f32x4){(float)z0, (float)z1d, (float)z2, (float)z3};
                }
                __syncthreads();
                EP_STAMP(4);
                TailNIO io;
                io.theta = a.theta; io.target = a.target; io.lossw = a.lossw; io.cot_mode = 0;
                io.a1 = nullptr; io.zstride = 0; io.KS = 0; io.z1img = s_z1; io.wlds = s_wc;
                io.delta1 = a.delta1; io.part_out = a.part + (long long)u * PS; io.inv_div = a.inv_div;
                const int r = 16 * u + c;
                const int prow = perm[r < B ? r : B - 1];
                (void)tailn_unit<L, true, EP_WAVES, true>(a.d, io, u, B, prow, s_zred, s_tile, nullptr, wave, ln);
                __syncthreads();                                     // (the LDS images are reused by the next unit)
            }
            EP_STAMP(5);
            EP_PUBLISH(EP_CNT_B + (w & 7));
            EP_STAMP(6);
        }
        // ================================ P3: G = xhat' delta1 for this block (all batch rows) ================================
        // operands in the gradient product's layout: lane (feature c, q) holds xhat[row 4 sx + q][feature]; the rows' indices
        // come out of idxU by lane shuffles.  The first four units' samples are requested BEFORE the wait for the tails.
        {
            f32x4 acc[CPS];
#pragma unroll
            for (int cc = 0; cc < CPS; ++cc) acc[cc] = (f32x4){0.f, 0.f, 0.f, 0.f};
            float dsum = 0.f;
            float av[4][4][CPS];
            auto loadx3 = [&](int i0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx) {
                        const int pr = __shfl(idxU[i0 + j], 4 * sx + q, 64);
                        const float* x = Xc + (long long)pr * a.Fp;
#pragma unroll
                        for (int cc = 0; cc < CPS; ++cc) av[j][sx][cc] = x[chk[cc]];
                    }
                }
            };
            if (nuw > 0) loadx3(0);
            EP_WAIT(ep_poll(a.cnt, EP_CNT_B, (ep_u32)nTail * (s + 1), -1, 0, ln));
            EP_STAMP(7);
#pragma unroll
            for (int i0 = 0; i0 < EP_UPW; i0 += 4) {
                if (i0 < nuw) {                                     // wave-uniform
                    float bq[4][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int u = wave + EP_WAVES * (i0 + j < nuw ? i0 + j : 0);
#pragma unroll
                        for (int sx = 0; sx < 4; ++sx)
                            bq[j][sx] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd1, ((16 * u + 4 * sx + q) * H1p + 16 * tg + c) * 4, 0, 16));
                    }
                    if (i0 > 0) loadx3(i0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (i0 + j < nuw) {
#pragma unroll
                            for (int sx = 0; sx < 4; ++sx) {
                                dsum += bq[j][sx];
#pragma unroll
                                for (int cc = 0; cc < CPS; ++cc) acc[cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][sx][cc], bq[j][sx], acc[cc], 0, 0, 0);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int cc = 0; cc < CPS; ++cc) *(f32x4*)(s_red + ((wave * CPS + cc) * 64 + ln) * 4) = acc[cc];
            dsum += __shfl_xor(dsum, 16, 64);
            dsum += __shfl_xor(dsum, 32, 64);
            if (q == 0) s_Dred[wave * 16 + c] = dsum;
        }
        EP_STAMP(8);
        if (s + 1 < a.nb) load_idx(s + 1);                           // the next step's batch rows: under the update phases
        // the batch loss: the units' partials, summed in one fixed order by every workgroup (the same decision everywhere)
        if (wave == 0) {
            float lp = 0.f;
            for (int u = ln; u < U; u += 64) lp += ep_ld(a.part + (long long)u * PS);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) lp += __shfl_xor(lp, o, 64);
            if (ln == 0) s_lstep = lp;
        }
        __syncthreads();
        const float lstep = s_lstep;
        const bool bad = collapsed || !isfinite(lstep);
        if (bad) collapsed = true; else loss_sum += (double)lstep;
        // E1: sum the eight waves' partial tiles (fixed order); D = column sums of delta1
        for (int e = tid; e < CPS * 256; e += EP_THREADS) {
            float G = s_red[e];
#pragma unroll
            for (int wv2 = 1; wv2 < EP_WAVES; ++wv2) G += s_red[wv2 * CPS * 256 + e];
            s_red[e] = G;
        }
        if (tid < 16) {
            float D = s_Dred[tid];
#pragma unroll
            for (int wv2 = 1; wv2 < EP_WAVES; ++wv2) D += s_Dred[wv2 * 16 + tid];
            s_D[tid] = D;
        }
        __syncthreads();
        EP_STAMP(9);
        // E2 (LayerNorm): this tile's share of dg[k] = sum_o W1[o,k] G[k,o], dbeta[k] = sum_o W1[o,k] D[o]
        if (ln_on && tid < CPS * 16) {
            float dg_own = 0.f, db_own = 0.f;
            const int cc = tid >> 4, kq = (tid & 15) >> 2, j = tid & 3;
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                const int e = cc * 256 + (o + 16 * kq) * 4 + j;
                dg_own += s_W[e] * s_red[e];
                db_own += s_W[e] * s_D[o];
            }
            if (nt > 1) {
                float* lp = a.lnpart + ((long long)(ks * nt + tg) * 2) * CPS * 16;
                ep_st(lp + tid, dg_own);
                ep_st(lp + CPS * 16 + tid, db_own);
            } else {
                s_own[tid] = dg_own; s_own[CPS * 16 + tid] = db_own;
            }
        }
        if (ln_on && nt > 1) EP_PUBLISH(EP_CNT_SIB + ks);
        else if (ln_on) __syncthreads();                             // (E2 read the weights E4 is about to overwrite)
        EP_STAMP(10);
        // E4: optimizer step on this block's weights (LayerNorm unfold: dW1 = g .* G + beta * D')
        if (!bad) {
            for (int e = tid; e < CPS * 256; e += EP_THREADS) {
                const int cc = e >> 8, lane2 = (e >> 2) & 63, j = e & 3;
                const int kl = cc * 16 + 4 * (lane2 >> 4) + j, o = lane2 & 15;
                const float G = s_red[e];
                const float gW = ln_on ? s_lnp[kl] * G + s_lnp[CPS * 16 + kl] * s_D[o] : G;
                float th = s_W[e], mi = s_mW[e], vi = s_vW[e];
                rule(th, gW, mi, vi);
                s_W[e] = th; s_mW[e] = mi; s_vW[e] = vi;
            }
        }
        // E6: layer-1 bias of this tile (ks == 0) and this workgroup's slice of the tail parameters: reduce, update, publish
        if (b1_role && tid < 16 && !bad) {
            const int o = 16 * tg + tid;
            if (o < H1) {
                const long long gi = a.off_b0 + o;
                float th = ep_ld(a.theta + gi), mi = ep_ld(a.m + gi), vi = ep_ld(a.v + gi);
                rule(th, s_D[tid], mi, vi);
                ep_st(a.theta + gi, th); ep_st(a.m + gi, mi); ep_st(a.v + gi, vi);
            }
        }
        if (slice_role && !bad) {
            // one element per 16-lane group, the lanes split the units' partials (four independent loads per lane and pass)
            for (int e = slo + (tid >> 4); e < shi; e += EP_THREADS / 16) {
                // partial-vector element -> parameter: [1 + H1 ..) = for each tail layer W (Flux order) then b
                long long gi = -1;
                int rem = e - sbase;
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const int nw = a.d.lay[l].K * a.d.lay[l].N, nb2 = a.d.lay[l].N;
                    if (gi < 0) {
                        if (rem < nw) gi = a.d.lay[l].gW + rem;
                        else if (rem < nw + nb2) gi = a.d.lay[l].gb + (rem - nw);
                        else rem -= nw + nb2;
                    }
                }
                const long long gic = gi >= 0 ? gi : a.off_b0;
                float th = ep_ld(a.theta + gic), mi = ep_ld(a.m + gic), vi = ep_ld(a.v + gic);      // (requested with the partials)
                float gsum = 0.f;
                for (int u0 = c; u0 < U; u0 += 64) {
                    float pv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) pv[i] = ep_ld(a.part + (long long)(u0 + 16 * i < U ? u0 + 16 * i : 0) * PS + e);
#pragma unroll
                    for (int i = 0; i < 4; ++i) gsum += u0 + 16 * i < U ? pv[i] : 0.f;
                }
#pragma unroll
                for (int sft = 8; sft > 0; sft >>= 1) gsum += __shfl_xor(gsum, sft, 64);
                if (gi >= 0 && c == 0) {
                    rule(th, gsum, mi, vi);
                    ep_st(a.theta + gi, th); ep_st(a.m + gi, mi); ep_st(a.v + gi, vi);
                }
            }
        }
        EP_STAMP(11);
        if (slice_role || b1_role) EP_PUBLISH(EP_CNT_C + (w & 7));
        EP_STAMP(12);
        // E5 (LayerNorm): the nt owners of this K slice exchange their shares, every one updates its own copy of g, beta
        if (ln_on) {
            if (nt > 1) EP_WAIT(ep_poll1(a.cnt, EP_CNT_SIB + ks, (ep_u32)nt * (s + 1), ln));
            else __syncthreads();                                   // (E4 read g, beta)
            if (tid < 2 * CPS * 16 && !bad) {
                const int which = tid >= CPS * 16, kl = tid - which * CPS * 16;
                float gsum;
                if (nt > 1) {
                    float pv[10];                                    // nt <= 10: every share requested before the first is used
#pragma unroll
                    for (int t2 = 0; t2 < 10; ++t2)
                        pv[t2] = ep_ld(a.lnpart + ((long long)(ks * nt + (t2 < nt ? t2 : 0)) * 2 + which) * CPS * 16 + kl);
                    gsum = 0.f;
#pragma unroll
                    for (int t2 = 0; t2 < 10; ++t2) gsum += t2 < nt ? pv[t2] : 0.f;
                } else {
                    gsum = s_own[which * CPS * 16 + kl];
                }
                float th = s_lnp[which * CPS * 16 + kl], mi = s_lnp[(2 + 2 * which) * CPS * 16 + kl], vi = s_lnp[(3 + 2 * which) * CPS * 16 + kl];
                if (c0 * 16 + kl < a.nch * 16) rule(th, gsum, mi, vi);
                s_lnp[which * CPS * 16 + kl] = th; s_lnp[(2 + 2 * which) * CPS * 16 + kl] = mi; s_lnp[(3 + 2 * which) * CPS * 16 + kl] = vi;
            }
        }
        __syncthreads();                                             // the LDS master copies are complete before the next P1 reads them
        EP_STAMP(13);
        bt1 *= (double)a.beta1; bt2 *= (double)a.beta2;              // (every step, as the host copy does)
    }
done:
    // ---- write the block back (also after an abort: what was updated stays consistent per element)
    __syncthreads();
    for (int e = tid; e < CPS * 256; e += EP_THREADS) {
        const int cc = e >> 8;
        if (c0 + cc < a.nch) {
            const long long gi = a.off_W0 + ((long long)(c0 + cc) * nt + tg) * 256 + (e & 255);
            a.theta[gi] = s_W[e]; a.m[gi] = s_mW[e]; a.v[gi] = s_vW[e];
        }
    }
    if (ln_on && tg == 0) {
        for (int e = tid; e < CPS * 16; e += EP_THREADS) {
            const int k = c0 * 16 + e;
            if (k < a.nch * 16) {
                a.theta[a.off_lns + k] = s_lnp[e]; a.theta[a.off_lnb + k] = s_lnp[CPS * 16 + e];
                a.m[a.off_lns + k] = s_lnp[2 * CPS * 16 + e]; a.v[a.off_lns + k] = s_lnp[3 * CPS * 16 + e];
                a.m[a.off_lnb + k] = s_lnp[4 * CPS * 16 + e]; a.v[a.off_lnb + k] = s_lnp[5 * CPS * 16 + e];
            }
        }
    }
    if (w == 0 && tid == 0) {
        // the epoch's result, as the last opt_step publishes it: [loss sum, collapsed flag (2: the launch was aborted), status word]
        const double acc0 = a.loss_acc[0] + loss_sum;
        a.loss_acc[0] = acc0;
        if (collapsed) a.loss_acc[1] = 1.0;
        if (a.host_pub) {
            a.host_pub[0] = acc0;
            a.host_pub[1] = aborted ? 2.0 : (collapsed ? 1.0 : 0.0);
            ((int*)(a.host_pub + 2))[0] = ((const int*)(a.loss_acc + 2))[0];
            a.host_pub[3] = a.loss_acc[3];
            __threadfence_system();
        }
    }
    if (aborted && tid == 0 && a.host_pub) { a.host_pub[1] = 2.0; __threadfence_system(); }
}

// ---------------------------------------------------------------------------------------------------------------------------
static int ep_cps(int nch, int nt, int ncu) {
    if (((nch + 3) / 4) * nt <= ncu) return 4;
    return 0;
}
size_t isk_train_epoch_lds(int cps, int H1p) {
    return (size_t)(3 * cps * 256 + 8 * cps * 16 + 32 + EP_WAVES * 16 + EP_WAVES * cps * 256 + 16 * H1p + EP_WAVES * 256 + EP_WAVES * 16 * TN_TS) * sizeof(float);
}

// plan of the persistent epoch for a batch of B rows; false: the shape is not covered (the caller keeps the four-kernel chain)
bool isk_train_epoch_plan(const TrainShape& s, int Fp, int B, int ncu, EpochPlan* p) {
    if (!isk_train_tail_narrow_ok(s)) return false;
    if (s.act != ISK_ACT_SIGMOID || s.lastact != ISK_ACT_IDENTITY) return false;        // (the inlined-activation build of the tail)
    const int nch = Fp / 16, nt = isk_pad16(s.widths[1]) / 16;
    const int cps = ep_cps(nch, nt, ncu);
    if (!cps) return false;
    p->cps = cps; p->KS = (nch + cps - 1) / cps; p->nt = nt; p->G = p->KS * nt;
    p->U = (B + 15) / 16;
    if (p->U < 1 || p->U > EP_WAVES * EP_UPW) return false;        // (eight units per wave: batches of at most 1024 rows)
    if (p->KS > EP_MAX_KS) return false;                           // (one counter line per K slice)
    p->zrows = p->U * 16;
    p->nS = p->G < 64 ? p->G : 64;                                 // owners of the tail-parameter slices
    p->nC = 0;
    for (int w = 0; w < p->G; ++w) p->nC += (w < p->nS || w / nt == 0) ? 1 : 0;
    long long nsmall = s.nparams - s.off_b[0];
    p->lds = isk_train_epoch_lds(cps, nt * 16) + (size_t)nsmall * sizeof(float);
    if (p->lds > 120 * 1024) return false;
    return true;
}

int isk_launch_train_epoch(const TrainShape& s, const EpochPlan& p, EpochArgs a, hipStream_t st) {
    a.d = isk_tailn_desc(s);
    a.cps = p.cps; a.KS = p.KS; a.nt = p.nt; a.G = p.G; a.U = p.U; a.zrows = p.zrows; a.nC = p.nC; a.nS = p.nS;
    a.nparams = s.nparams;
    const int L = s.nlayers - 1;
    if (hipMemsetAsync(a.cnt, 0, ISK_EP_CNT_WORDS * sizeof(unsigned), st) != hipSuccess) return 1;
#define EPCASE(L_, C_)                                                                                                                          \
    {                                                                                                                                           \
        if (p.lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)train_epoch_kernel<L_, C_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds); \
        hipLaunchKernelGGL((train_epoch_kernel<L_, C_>), dim3(p.G), dim3(EP_THREADS), p.lds, st, a);                                           \
    }
    if (L == 1) EPCASE(1, 4) else if (L == 2) EPCASE(2, 4) else EPCASE(3, 4)
#undef EPCASE
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
#endif  // ISK_EXPERIMENTAL
