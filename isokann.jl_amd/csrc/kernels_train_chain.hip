// One training epoch of train_batch! (src/iso.jl:179-194 of the reference) for SMALL batches (B <= 128: the reference's default
// minibatch is 100) as ONE persistent launch on the workgroups of ONE XCD -- VERDICT round 4, item 2.
//
// Why one XCD.  A step of the launch chain costs 28.5 us at minibatch 100, all of it launch floors (three dependent launches).
// Round 4's persistent epoch replaced them by in-launch seams across 243 workgroups and lost: such a seam costs 23 us
// (scripts/ablation/seam_cost.hip).  Among <= 32 workgroups that share an XCD -- and therefore an L2 -- it costs 1.9 us: plain
// stores, s_waitcnt vmcnt(0), a workgroup barrier, ONE workgroup-scope atomic add (performed in that L2), polls and payload loads
// that bypass the reader's L1 (nt).  That form is valid ONLY among workgroups of one XCD, so the kernel does not assume where it
// runs: 256 workgroups are launched, each reads HW_REG_XCC_ID, the first to arrive names its XCD, the workgroups of that XCD draw
// tickets 0 .. P-1 and stay, all others leave at once.  (Round-robin dispatch put 32 workgroups on every XCD in every launch
// measured; if fewer than P turn up the first wait times out BEFORE anything has been updated and the launch reports it.)
//
// Ownership.  Workgroup p owns the K slice [64 p, 64 p + 64) of the layer-1 weights -- ALL nt output tiles of it, fragment-linear in
// LDS for the whole epoch -- with its LayerNorm scale / bias slice; the optimizer state of the slice stays in global memory (only its
// owner touches it).  For that slice BOTH layer-1 products are local:
//     forward   z1[rows][:]   += (g .* xhat[rows][slice] + beta) W1[slice][:]'        a K-slice partial for every batch row
//     backward  G[slice][:]     = xhat[rows][slice]' delta1[rows][:]                    complete
// so the LayerNorm unfold (dW1 = g .* G + beta D', dg = sum_o W1 .* G, dbeta = W1' D: all of them sums over the slice's own rows of W1)
// and the optimizer step need no communication.  Workgroups u < U (= ceil(B / 16)) also run the tail of unit u (train_tailn.hpp) and
// keep a PRIVATE copy of the small parameters (layer-1 bias + tail layers) with their optimizer state in LDS, which each of them
// updates itself from the units' partial sums -- the same bits everywhere.  Hence TWO seams per step, not three:
//     A   K-slice partials (all P)  ->  the U tail workgroups
//     B   delta1 + the units' small-gradient partials + loss partials (the U tail workgroups)  ->  all P
// Results: those of the launch chain up to summation order (P K-slice partials instead of <= 8; summed in double).
#include "isk_internal.h"
#ifndef ISK_EXPERIMENTAL
bool isk_train_chain_plan(const TrainShape&, int, int, EpochPlan*) { return false; }
int isk_launch_train_chain(const TrainShape&, const EpochPlan&, EpochArgs, hipStream_t) { return -1; }
#else
#include "train_tailn.hpp"

#define CH_THREADS 512
#define CH_WAVES 8
#define CH_CPS 4                              /* 16-feature chunks per K slice */
#define CH_LINE 32                            /* counter words are a 128-byte line apart */
#define CH_TICKET 0
#define CH_XCC 1
#define CH_ABORT 2
#define CH_SEAM_A 3
#define CH_SEAM_B 4
#define CH_TIMEOUT 20000000ull                /* 0.2 s of s_memrealtime (100 MHz) */
#define CH_MAXNT 10
#define CH_TPW 5                              /* gradient tiles per wave: ceil(CH_MAXNT / 2) */
#define CH_BIG(H1p) ((128 * (H1p)) > (CH_WAVES * 256 + CH_WAVES * 16 * TN_TS) ? (128 * (H1p)) : (CH_WAVES * 256 + CH_WAVES * 16 * TN_TS))   /* floats of s_big */
#define CH_DLOADS 10                          /* float4 loads per thread that bring delta1 of 128 rows x 160 columns to LDS */

typedef unsigned ch_u32;
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // a float4 load from a 4-byte aligned address

__device__ __forceinline__ ch_u32 ch_xcc_id() {
    ch_u32 v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
// a load that bypasses this CU's L1 and is served by the XCD's L2, never hoisted or merged by the compiler
__device__ __forceinline__ ch_u32 ch_poll_word(const ch_u32* p) {
    ch_u32 v;
    asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// lane 0 of wave 0: until counter `id` has seen `want` adds (or the launch is aborted / the wait times out)
__device__ __forceinline__ bool ch_wait(ch_u32* cnt, int id, ch_u32 want) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        if (ch_poll_word(cnt + id * CH_LINE) >= want) return true;
        if (ch_poll_word(cnt + CH_ABORT * CH_LINE)) return false;
        if (__builtin_amdgcn_s_memrealtime() - t0 > CH_TIMEOUT) {
            __hip_atomic_store(cnt + CH_ABORT * CH_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// every wave has drained its stores (they are in the XCD's L2); one lane signals
#define CH_PUBLISH(id)                                                                                                  \
    do {                                                                                                                \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                \
        __syncthreads();                                                                                                \
        if (tid == 0) __hip_atomic_fetch_add(a.cnt + (id) * CH_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
    } while (0)
#define CH_WAIT(id, want)                                                                                               \
    do {                                                                                                                \
        if (tid == 0) s_ok = ch_wait(a.cnt, (id), (want)) ? 1 : 0;                                                      \
        __syncthreads();                                                                                                \
        asm volatile("" ::: "memory");                                                                                  \
        CH_ACQUIRE();                                                                                                   \
        if (!s_ok) { aborted = true; goto done; }                                                                       \
    } while (0)

// a workgroup barrier for LDS hand-offs only: __syncthreads() also drains vmcnt -- every outstanding global load (the next step's
// rows) and store (the optimizer state) -- which is what the step must NOT wait for
// payload written by OTHER workgroups of the XCD is read with nt loads (served by the L2, past this CU's L1).  The alternative --
// invalidate the L1 after the wait (buffer_inv sc1), then plain loads -- was measured: the 64 KB delta1 image read by all 27
// workgroups took 13.5 us instead of 1.9 us, the K reduce 6.4 instead of 5.2 (-DCH_PLAIN_PAYLOAD rebuilds that variant)
#ifdef CH_PLAIN_PAYLOAD
#define CH_LOAD(p) (*(p))
#define CH_ACQUIRE() asm volatile("buffer_inv sc1" ::: "memory")
#else
#define CH_LOAD(p) __builtin_nontemporal_load(p)
#define CH_ACQUIRE() do {} while (0)
#endif
#define CH_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct ChRule {
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

template <int L>
__global__ __launch_bounds__(CH_THREADS) void train_chain_kernel(EpochArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ int s_me, s_ok;
    __shared__ float s_lstep;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63, c = ln & 15, q = ln >> 4;
    // ---- rendezvous: the workgroups of ONE XCD (the first arriver's) take tickets; everybody else leaves
    if (tid == 0) {
        const ch_u32 x = ch_xcc_id() + 1;
        ch_u32 chosen = __hip_atomic_load(a.cnt + CH_XCC * CH_LINE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (chosen == 0) {
            ch_u32 expected = 0;
            __hip_atomic_compare_exchange_strong(a.cnt + CH_XCC * CH_LINE, &expected, x, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            chosen = expected == 0 ? x : expected;
        }
        s_me = chosen == x ? (int)__hip_atomic_fetch_add(a.cnt + CH_TICKET * CH_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
    }
    __syncthreads();
    const int p = s_me;
    const int P = a.G;
    if (p < 0 || p >= P) return;
    const int nt = a.nt, H1p = nt * 16, H1 = a.d.H1;
    const int c0 = p * CH_CPS;                                       // first 16-feature chunk of this slice
    const int U = a.U, B = a.B;
    const int nTail = U < P ? U : P;                                  // workgroups p < nTail run the tails of units p, p + P, ...
    const bool tail_role = p < nTail, ln_on = a.ln != 0;
    const int PS = (int)a.d.PS;
    const int nsmall = (int)(a.nparams - a.off_b0);
    bool aborted = false;
    // ---- LDS carve
    float* s_W = sm;                                 // [CH_CPS][nt][256] fragment-linear layer-1 weights of this slice (master copy)
    float* s_g = s_W + CH_CPS * nt * 256;            // [64] LayerNorm scale of the slice (1 without LayerNorm)
    float* s_be = s_g + 64;                          // [64] LayerNorm bias
    float* s_D = s_be + 64;                          // [H1p] column sums of delta1
    float* s_lnr = s_D + H1p;                        // [CH_WAVES][2][16]: the waves' partial dg / dbeta (by chunk-local feature)
    int* s_perm = (int*)(s_lnr + CH_WAVES * 32);     // [3][128] the batch rows of this step and the next two (row index clamped to the batch)
    float* s_z1 = (float*)(s_perm + 384);            // tail role: [16][H1p] z1 image
    float* s_big = s_z1 + 16 * H1p;                  // [128][H1p]: delta1 of the batch (phase C); before that the tails' scratch:
                                                     //   [8 waves][64 nt] float4 partial sums of the K reduce, then s_zred / s_tile
    float* s_zred = s_big;                           // [8][256]
    float* s_tile = s_zred + CH_WAVES * 256;         // [8][16 * TN_TS]
    float* s_wc = s_big + CH_BIG(H1p);              // [nsmall] small parameters (layer-1 bias, tail layers): private copy
    float* s_wm = s_wc + nsmall;                     // [nsmall] their first moment / velocity
    float* s_wv = s_wm + nsmall;                     // [nsmall] their second moment
    for (int e = tid; e < CH_CPS * nt * 256; e += CH_THREADS) {
        const int cc = e / (nt * 256);
        s_W[e] = c0 + cc < a.nch ? a.theta[a.off_W0 + (long long)c0 * nt * 256 + e] : 0.f;
    }
    for (int e = tid; e < 64; e += CH_THREADS) {
        const int k = c0 * 16 + e;
        const bool ok = ln_on && k < a.nch * 16;
        s_g[e] = ok ? a.theta[a.off_lns + k] : 1.f;
        s_be[e] = ok ? a.theta[a.off_lnb + k] : 0.f;
    }
    if (tid < 256 && (tid < 128 || a.nb > 1)) s_perm[tid] = a.perm[(long long)(tid >> 7) * B + ((tid & 127) < B ? (tid & 127) : B - 1)];
    if (tail_role) {
        for (int e = tid; e < nsmall; e += CH_THREADS) { s_wc[e] = a.theta[a.off_b0 + e]; s_wm[e] = a.m[a.off_b0 + e]; s_wv[e] = a.v[a.off_b0 + e]; }
    }
    __syncthreads();
    ChRule rule;
    rule.kind = a.opt_kind; rule.lr = a.lr; rule.reg = a.reg; rule.beta1 = a.beta1; rule.beta2 = a.beta2; rule.eps = a.eps; rule.rho = a.rho;
    double bt1 = a.bt1, bt2 = a.bt2;
    double loss_sum = 0.0;
    bool collapsed = a.collapsed_in != 0;
    const int zslice = a.zrows * H1p * 4;                             // bytes between two K-slice partials
    // the gradient tiles of this wave: chunk cc = wave & 3, output tiles t = (wave >> 2) + 2 i
    const int gcc = wave & 3, gt0 = wave >> 2;
    const bool gch_ok = c0 + gcc < a.nch;
    const int gchunk = 16 * (gch_ok ? c0 + gcc : a.nch - 1);           // (a padding chunk re-reads the last one: its weights are 0, its gradient is dropped)

    // the rows of step 0 for the forward (unit = wave, lane (sample c, q)); every later step requests its rows one step ahead
    f32x4 xraw[CH_CPS];
    if (wave < U) {
        const int r = 16 * wave + c;
        const float* x = a.X + (long long)a.perm[r < B ? r : B - 1] * a.Fp + 4 * q;
#pragma unroll
        for (int cc = 0; cc < CH_CPS; ++cc) xraw[cc] = *(const f32x4*)(x + 16 * (c0 + cc < a.nch ? c0 + cc : a.nch - 1));
    }
#define CH_STAMP(k) do { if (a.dbg && tid == 0 && (p == 0 || p == P - 1) && s == (a.nb > 1 ? 1 : 0)) a.dbg[(p == 0 ? 0 : 32) + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    for (int s = 0; s < a.nb; ++s) {
        // thread coordinates made opaque once per step: as invariants of the step loop every address derived from them (the inlined
        // tail's included) was precomputed outside it and held in registers -- some 200 of them, the rest spilled
        int tix = tid;
        asm volatile("" : "+v"(tix));
        const int lx = tix & 63, cx = lx & 15, qx = lx >> 4;
        const int* perm = s_perm + 128 * (s % 3);
        const int* perm_next = s_perm + 128 * ((s + 1) % 3);
        int* perm_next2 = s_perm + 128 * ((s + 2) % 3);
        int pnext = 0;
        if (tix < 128 && s + 2 < a.nb) pnext = a.perm[(long long)(s + 2) * B + (tix < B ? tix : B - 1)];     // in LDS by the middle of this step
        // the rows of the NEXT step are pulled into this XCD's L2 now (three touches cover the 256-byte slice of a row wherever it
        // starts in a line); the value is never used -- it is kept live until the next vmcnt(0) so that its register is not reused
        float touch = 0.f;
        if (s + 1 < a.nb && wave < U) {
            const int off = (qx * 32) < 63 ? qx * 32 : 63;
            touch = a.X[(long long)perm_next[16 * wave + cx] * a.Fp + c0 * 16 + (c0 * 16 + off < a.Fp ? off : 0)];
        }
        CH_STAMP(0);
        rule.bc1 = (float)(1.0 - bt1); rule.bc2 = (float)(1.0 - bt2);
        // ================================ A: K-slice partial of z1 for unit `wave`, every tile ================================
        // (the rows of this step were requested during the previous step's optimizer phase: xraw)
        if (wave < U) {
            const int u = wave;
            f32x4 xa[CH_CPS];
#pragma unroll
            for (int cc = 0; cc < CH_CPS; ++cc)                       // the LayerNorm affine on the sample operand (one fma per element, as train_fwd1n does)
                xa[cc] = xraw[cc] * *(const f32x4*)(s_g + 16 * cc + 4 * qx) + *(const f32x4*)(s_be + 16 * cc + 4 * qx);
            for (int t = 0; t < nt; ++t) {
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
                for (int cc = 0; cc < CH_CPS; ++cc) {
                    const f32x4 wf = *(const f32x4*)(s_W + (cc * nt + t) * 256 + lx * 4);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        if (jj & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[jj], xa[cc][jj], acc1, 0, 0, 0);
                        else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[jj], xa[cc][jj], acc0, 0, 0, 0);
                    }
                }
                // RL: lane (sample cx, qx) holds z1[sample][16 t + 4 qx + r]
                *(f32x4*)(a.zp + (size_t)p * a.zrows * H1p + (size_t)(16 * u + cx) * H1p + 16 * t + 4 * qx) = acc0 + acc1;
            }
        }
        CH_STAMP(1);
        CH_PUBLISH(CH_SEAM_A);
        asm volatile("" :: "v"(touch));
        CH_STAMP(2);
        // ================================ B: the tails of units p, p + P, ... (workgroups p < min(U, P)) ================================
        if (tail_role) {
            CH_WAIT(CH_SEAM_A, (ch_u32)P * (ch_u32)(s + 1));
            CH_STAMP(3);
            for (int u = p; u < U; u += P) {
                // z1 of the unit = sum of the P partials.  The unit's block of a partial is 64 nt contiguous float4 items: wave w sums
                // partials w, w + 8, w + 16, w + 24 for every item (lane + 64 r), the eight wave sums meet in LDS.
                const char* zb = (const char*)(a.zp + (size_t)(16 * u) * H1p) + lx * 16;
                for (int r0 = 0; r0 < nt; r0 += 3) {
                    f32x4 pz[3][4];
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {                 // (unconditional, from a clamped address: see av below)
                            const int k = wave + 8 * kk, r = r0 + rr;
                            pz[rr][kk] = CH_LOAD((const f32x4*)(zb + (size_t)(k < P ? k : 0) * zslice + (size_t)(r < nt ? r : 0) * 1024));
                        }
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) {
                        f32x4 sum = pz[rr][0];                              // (partial `wave` exists: P >= 8 or the wave's sum is dropped below)
#pragma unroll
                        for (int kk = 1; kk < 4; ++kk)
                            if (wave + 8 * kk < P) sum += pz[rr][kk];
                        if (wave >= P) sum = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (r0 + rr < nt) *(f32x4*)(s_big + ((size_t)wave * nt + r0 + rr) * 256 + lx * 4) = sum;
                    }
                }
                CH_LDS_BARRIER();
                for (int i = tix; i < 64 * nt; i += CH_THREADS) {
                    f32x4 w8[CH_WAVES];
#pragma unroll
                    for (int w = 0; w < CH_WAVES; ++w) w8[w] = *(const f32x4*)(s_big + (size_t)w * nt * 256 + 4 * i);
                    *(f32x4*)(s_z1 + 4 * i) = ((w8[0] + w8[1]) + (w8[2] + w8[3])) + ((w8[4] + w8[5]) + (w8[6] + w8[7]));
                }
                CH_LDS_BARRIER();
                CH_STAMP(4);
                TailNIO io;
                io.theta = a.theta; io.target = a.target; io.lossw = a.lossw; io.cot_mode = 0;
                io.a1 = nullptr; io.zstride = 0; io.KS = 0; io.z1img = s_z1; io.wlds = s_wc;
                io.delta1 = a.delta1; io.part_out = a.part + (long long)u * PS; io.inv_div = a.inv_div;
                const int prow = perm[16 * u + cx];
                (void)tailn_unit<L, true, CH_WAVES, true>(a.d, io, u, B, prow, s_zred, s_tile, nullptr, wave, lx);
                CH_LDS_BARRIER();                                      // (the LDS images are reused by the next unit, then by delta1)
            }
            CH_STAMP(5);
            CH_PUBLISH(CH_SEAM_B);
        }
        // ================================ C: G = xhat' delta1 for this slice (all batch rows) ================================
        // lane (feature cx, qx) of step sx holds xhat[row 4 sx + qx][feature cx of chunk gcc] (A operand) and delta1[row 4 sx + qx][16 t + cx]
        // (B operand, from the LDS image of delta1); accumulator element (lane, j) of tile t IS gradient element wbase + t * 256 + j
        {
            const long long wbx = a.off_W0 + ((long long)(c0 + gcc) * nt) * 256 + lx * 4;      // + t * 256: this lane's four elements of tile t
            // the samples do not depend on the tails: requested before the wait (the lines are in L2 since phase A)
            float av[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) {
                    // (rows past the batch: the clamped row index of s_perm; they contribute nothing because the tail zeroes their delta1.
                    // A load under a per-lane condition becomes its own exec-masked block, each with its own wait)
                    av[u][sx] = a.X[(long long)perm[16 * u + 4 * sx + qx] * a.Fp + gchunk + cx];
                }
            CH_STAMP(6);
            CH_WAIT(CH_SEAM_B, (ch_u32)nTail * (ch_u32)(s + 1));
            CH_STAMP(7);
            // delta1 of the batch -> LDS (U x 64 nt float4 items, one batch of loads), the loss partials and this wave's optimizer state
            {
                f32x4 dq[CH_DLOADS];
                const int nitem = U * 64 * nt;
                const int rot = (int)(((long long)p * nitem) / P) & ~7;          // (a multiple of 8 items = one 128-byte line)
#pragma unroll
                for (int i = 0; i < CH_DLOADS; ++i) {
                    // every workgroup reads the same image: each starts elsewhere in it, or all P walk the L2 channels in lockstep
                    const int it = tix + i * CH_THREADS;
                    int src = it + rot; src = src >= nitem ? src - nitem : src;
                    dq[i] = CH_LOAD((const f32x4*)a.delta1 + (it < nitem ? src : 0));       // (unconditional: see av)
                }
                float lp = 0.f;
                if (wave == 0) lp = CH_LOAD(a.part + (size_t)(lx < U ? lx : 0) * PS);
#pragma unroll
                for (int i = 0; i < CH_DLOADS; ++i) {
                    const int it = tix + i * CH_THREADS;
                    int src = it + rot; src = src >= nitem ? src - nitem : src;
                    if (it < nitem) *(f32x4*)(s_big + 4 * src) = dq[i];
                }
                if (wave == 0) {
                    // the batch loss: the units' partials, summed in one fixed order by every workgroup (the same decision everywhere)
                    lp = lx < U ? lp : 0.f;
#pragma unroll
                    for (int o = 1; o < 8; o <<= 1) lp += __shfl_xor(lp, o, 64);     // U <= 8: a fixed tree
                    if (lx == 0) s_lstep = lp;
                }
            }
            f32x4 mW[CH_TPW], vW[CH_TPW];
#pragma unroll
            for (int i = 0; i < CH_TPW; ++i) {
                const int t = gt0 + 2 * i;
                const bool ok = t < nt && gch_ok;
                mW[i] = ok ? CH_LOAD((const f32x4*)(a.m + wbx + (long long)t * 256)) : (f32x4){0.f, 0.f, 0.f, 0.f};
                vW[i] = (ok && a.opt_kind == ISK_OPT_ADAM) ? CH_LOAD((const f32x4*)(a.v + wbx + (long long)t * 256)) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (tix < 128) perm_next2[tix] = pnext;
            CH_LDS_BARRIER();                                          // the delta1 image, s_lstep, the next step's rows
            CH_STAMP(8);
            f32x4 acc[CH_TPW];
            float dsum[CH_TPW];
#pragma unroll
            for (int i = 0; i < CH_TPW; ++i) { acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dsum[i] = 0.f; }
            // (the offset is made opaque once per step: as a loop invariant of the step loop the compiler precomputed all 160 LDS addresses
            // of the unrolled product and spilled them)
            int bo = qx * H1p + cx + 16 * gt0;
            asm volatile("" : "+v"(bo));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u < U) {
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx) {
                        const float* rp = s_big + bo + (16 * u + 4 * sx) * H1p;
#pragma unroll
                        for (int i = 0; i < CH_TPW; ++i) {
                            // NO test of the tile against nt here: a tile past the last one reads the next row's columns (inside the
                            // allocation) into an accumulator nobody uses; a branch per product put an LDS wait before every MFMA
                            const float bq = rp[32 * i];
                            dsum[i] += bq;
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][sx], bq, acc[i], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);                    // (keeps the 20 LDS reads of a unit from being hoisted above the previous unit's)
            }
            CH_STAMP(9);
            // what the optimizer phase reads from other workgroups / from memory is requested BEFORE the next step's rows (loads return in
            // order: behind the row gather they would wait for its HBM latency): the units' small-gradient partials, four consecutive
            // elements per thread, and the optimizer state of the LayerNorm slice
            f32x4 pv[8];
            const int pe4 = 1 + H1 + 4 * tix;
            if (tail_role) {
#pragma unroll
                for (int u = 0; u < 8; ++u)        // (a float4 may end up to 3 floats past a partial vector: the next unit's loss slot, or the slack of the allocation)
                    pv[u] = CH_LOAD((const f32x4u*)(a.part + (size_t)(u < U ? u : 0) * PS + (pe4 < PS ? pe4 : 0)));     // (unconditional: see av)
            }
            float lnm = 0.f, lnv = 0.f;
            const int lnk = c0 * 16 + (tix & 63);
            const long long lngi = ((tix >> 6) ? a.off_lnb : a.off_lns) + lnk;
            const bool ln_mine = ln_on && tix < 128 && lnk < a.nch * 16;
            if (ln_mine) { lnm = CH_LOAD(a.m + lngi); lnv = CH_LOAD(a.v + lngi); }
            // summed at once, in unit order as the launch chain does (eight float4 held to the end of the step spilled)
            f32x4 gs4 = {0.f, 0.f, 0.f, 0.f};
            if (tail_role) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (u < U) gs4 += pv[u];
            }
            // the rows of the NEXT step's forward are requested now (in flight during the optimizer phase and the seam)
            if (s + 1 < a.nb && wave < U) {
                const int r = 16 * wave + cx;
                const float* x = a.X + (long long)perm_next[r] * a.Fp + 4 * qx;
#pragma unroll
                for (int cc = 0; cc < CH_CPS; ++cc) xraw[cc] = *(const f32x4*)(x + 16 * (c0 + cc < a.nch ? c0 + cc : a.nch - 1));
            }
            // D = column sums of delta1 (rows past the batch are 0 in delta1: the tail masks them): the waves of chunk 0 own them
#pragma unroll
            for (int i = 0; i < CH_TPW; ++i) {
                dsum[i] += __shfl_xor(dsum[i], 16, 64);
                dsum[i] += __shfl_xor(dsum[i], 32, 64);
                const int t = gt0 + 2 * i;
                if (gcc == 0 && qx == 0 && t < nt) s_D[16 * t + cx] = dsum[i];
            }
            CH_STAMP(10);
            CH_LDS_BARRIER();                                          // s_D
            const float lstep = s_lstep;
            const bool bad = collapsed || !isfinite(lstep);
            if (bad) collapsed = true; else loss_sum += (double)lstep;
            // LayerNorm: this wave's share of dg[k] = sum_o W1[o,k] G[k,o], dbeta[k] = sum_o W1[o,k] D[o] over its tiles; k = chunk gcc,
            // feature 4 (lane / 16) + j; summed over the 16 lanes cx (= output o within a tile)
            f32x4 wv4[CH_TPW];
            float dgp[4] = {0.f, 0.f, 0.f, 0.f}, dbp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < CH_TPW; ++i) {
                const int t = gt0 + 2 * i;
                wv4[i] = t < nt ? *(const f32x4*)(s_W + (gcc * nt + t) * 256 + lx * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
                const float Do = t < nt ? s_D[16 * t + cx] : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) { dgp[j] += wv4[i][j] * acc[i][j]; dbp[j] += wv4[i][j] * Do; }
            }
            if (ln_on) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) { dgp[j] += __shfl_xor(dgp[j], o, 64); dbp[j] += __shfl_xor(dbp[j], o, 64); }
                }
                if (cx == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { s_lnr[wave * 32 + 4 * qx + j] = dgp[j]; s_lnr[wave * 32 + 16 + 4 * qx + j] = dbp[j]; }
                }
            }
            // the optimizer step on this wave's tiles (LayerNorm unfold: dW1 = g .* G + beta * D'), straight from the accumulators
            if (!bad && gch_ok) {
#pragma unroll
                for (int i = 0; i < CH_TPW; ++i) {
                    const int t = gt0 + 2 * i;
                    if (t < nt) {
                        const float Do = s_D[16 * t + cx];
                        f32x4 th = wv4[i];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int kl = 16 * gcc + 4 * qx + j;
                            const float gW = ln_on ? s_g[kl] * acc[i][j] + s_be[kl] * Do : acc[i][j];
                            float t1 = th[j], m1 = mW[i][j], v1 = vW[i][j];
                            rule(t1, gW, m1, v1);
                            th[j] = t1; mW[i][j] = m1; vW[i][j] = v1;
                        }
                        wv4[i] = th;
                        *(f32x4*)(a.m + wbx + (long long)t * 256) = mW[i];
                        if (a.opt_kind == ISK_OPT_ADAM) *(f32x4*)(a.v + wbx + (long long)t * 256) = vW[i];
                    }
                }
            }
            CH_LDS_BARRIER();                                          // every wave has read the old weights / g / beta; s_lnr is complete
            if (!bad && gch_ok) {
#pragma unroll
                for (int i = 0; i < CH_TPW; ++i) {
                    const int t = gt0 + 2 * i;
                    if (t < nt) *(f32x4*)(s_W + (gcc * nt + t) * 256 + lx * 4) = wv4[i];
                }
            }
            // LayerNorm scale / bias of the slice: thread (which, kl); the two waves of a chunk (t even / odd) in wave order
            if (ln_mine && !bad) {
                const int which = tix >> 6, kl = tix & 63, cc = kl >> 4, f = kl & 15;
                const float gsum = s_lnr[cc * 32 + which * 16 + f] + s_lnr[(cc + 4) * 32 + which * 16 + f];
                float th = which ? s_be[kl] : s_g[kl];
                rule(th, gsum, lnm, lnv);
                if (which) s_be[kl] = th; else s_g[kl] = th;
                a.m[lngi] = lnm; a.v[lngi] = lnv;
            }
            // the small parameters (tail role: every tail workgroup updates ITS copy from the same sums)
            if (tail_role && !bad) {
                // layer-1 bias: its gradient is D
                for (int e = tix; e < H1; e += CH_THREADS) {
                    float th = s_wc[e], mi = s_wm[e], vi = s_wv[e];
                    rule(th, s_D[e], mi, vi);
                    s_wc[e] = th; s_wm[e] = mi; s_wv[e] = vi;
                }
                // tail layers: partial-vector element pe in [1 + H1, PS) = for each tail layer W (Flux order) then b -> its parameter
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pe = pe4 + j;
                    if (pe >= PS) continue;
                    long long gi = -1;
                    int rem = pe - (1 + H1);
#pragma unroll
                    for (int l = 0; l < L; ++l) {
                        const int nw = a.d.lay[l].K * a.d.lay[l].N, nb2 = a.d.lay[l].N;
                        if (gi < 0) {
                            if (rem < nw) gi = a.d.lay[l].gW + rem;
                            else if (rem < nw + nb2) gi = a.d.lay[l].gb + (rem - nw);
                            else rem -= nw + nb2;
                        }
                    }
                    if (gi >= 0) {
                        const float gsum = gs4[j];
                        const int e = (int)(gi - a.off_b0);
                        float th = s_wc[e], mi = s_wm[e], vi = s_wv[e];
                        rule(th, gsum, mi, vi);
                        s_wc[e] = th; s_wm[e] = mi; s_wv[e] = vi;
                    }
                }
            }
            CH_LDS_BARRIER();                                          // the LDS master copies are complete before the next step reads them
        }
        CH_STAMP(11);
        bt1 *= (double)a.beta1; bt2 *= (double)a.beta2;
    }
done:
    // ---- write the slice back (the optimizer state of the layer-1 weights and of g / beta already lives in global memory)
    __syncthreads();
    {
        for (int e = tid; e < CH_CPS * nt * 256; e += CH_THREADS) {
            const int cc = e / (nt * 256);
            if (c0 + cc < a.nch) a.theta[a.off_W0 + (long long)c0 * nt * 256 + e] = s_W[e];
        }
        if (ln_on) {
            for (int e = tid; e < 64; e += CH_THREADS) {
                const int k = c0 * 16 + e;
                if (k < a.nch * 16) { a.theta[a.off_lns + k] = s_g[e]; a.theta[a.off_lnb + k] = s_be[e]; }
            }
        }
        if (p == 0) {
            for (int e = tid; e < nsmall; e += CH_THREADS) { a.theta[a.off_b0 + e] = s_wc[e]; a.m[a.off_b0 + e] = s_wm[e]; a.v[a.off_b0 + e] = s_wv[e]; }
        }
    }
    if (p == 0 && tid == 0) {
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

static size_t chain_lds(int nt, long long nsmall) {
    const int H1p = nt * 16;
    return (size_t)(CH_CPS * nt * 256 + 128 + H1p + CH_WAVES * 32 + 384 + 16 * H1p + CH_BIG(H1p) + 3 * nsmall) * sizeof(float);
}

// plan for a batch of B rows; false: the shape is not covered (the caller keeps the launch chain)
bool isk_train_chain_plan(const TrainShape& s, int Fp, int B, EpochPlan* p) {
    if (!isk_train_tail_narrow_ok(s)) return false;
    if (s.act != ISK_ACT_SIGMOID || s.lastact != ISK_ACT_IDENTITY) return false;        // (the inlined-activation build of the tail)
    const int nch = Fp / 16, nt = isk_pad16(s.widths[1]) / 16;
    if (nt > CH_MAXNT) return false;
    p->cps = CH_CPS; p->KS = (nch + CH_CPS - 1) / CH_CPS; p->nt = nt; p->G = p->KS;
    if (p->G > 32) return false;                                   // one XCD
    p->U = (B + 15) / 16;
    if (p->U < 1 || p->U > 8) return false;                        // one unit per wave in the forward
    if (isk_tailn_desc(s).PS - 1 - s.widths[1] > 4 * CH_THREADS) return false;          // the small-gradient partials: four elements per thread, one round
    p->zrows = p->U * 16;
    p->nS = 0; p->nC = 0;
    p->lds = chain_lds(nt, s.nparams - s.off_b[0]);
    return p->lds <= 158 * 1024;
}

int isk_launch_train_chain(const TrainShape& s, const EpochPlan& p, EpochArgs a, hipStream_t st) {
    a.d = isk_tailn_desc(s);
    a.cps = p.cps; a.KS = p.KS; a.nt = p.nt; a.G = p.G; a.U = p.U; a.zrows = p.zrows; a.nC = 0; a.nS = 0;
    a.nparams = s.nparams;
    const int L = s.nlayers - 1;
    if (hipMemsetAsync(a.cnt, 0, ISK_EP_CNT_WORDS * sizeof(unsigned), st) != hipSuccess) return 1;
#define CHCASE(L_)                                                                                                                       \
    {                                                                                                                                    \
        (void)hipFuncSetAttribute((const void*)train_chain_kernel<L_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);          \
        hipLaunchKernelGGL((train_chain_kernel<L_>), dim3(256), dim3(CH_THREADS), p.lds, st, a);                                         \
    }
    if (L == 1) CHCASE(1) else if (L == 2) CHCASE(2) else CHCASE(3)
#undef CHCASE
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
#endif  // ISK_EXPERIMENTAL
